from . import public  # noqa: F401
