class StructRef:
    pass


class NoneType:
    pass


class Integer:
    pass


class ArrayCompatible:
    pass


def unliteral(t):
    return t


intc = 'intc'
float64 = 'float64'
