def overload(*args, **kwargs):
    def deco(fn):
        return fn
    return deco


def overload_method(*args, **kwargs):
    def deco(fn):
        return fn
    return deco
