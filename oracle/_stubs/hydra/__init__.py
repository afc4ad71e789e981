"""Stub: hydra.main identity decorator + hydra.utils.instantiate (see pytorch_lightning stub header)."""
from . import utils  # noqa: F401


def main(*a, **kw):
    def deco(fn):
        return fn

    return deco
