from . import types  # noqa: F401


def rank_zero_only(fn):
    return fn
