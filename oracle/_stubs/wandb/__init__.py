from . import apis, errors  # noqa: F401


class Api:
    pass


run = None
