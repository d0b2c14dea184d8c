from . import structref  # noqa: F401
