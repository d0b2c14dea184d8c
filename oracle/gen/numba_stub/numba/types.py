from .core.types import *  # noqa: F401,F403
from .core.types import StructRef, NoneType, Integer, ArrayCompatible, unliteral  # noqa: F401
