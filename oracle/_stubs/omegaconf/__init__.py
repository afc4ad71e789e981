"""Stub: attribute-dict DictConfig (see pytorch_lightning stub header)."""
from contextlib import contextmanager

from . import errors  # noqa: F401


class DictConfig(dict):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class ListConfig(list):
    pass


class OmegaConf:
    @staticmethod
    def create(d=None):
        return DictConfig(d or {})

    @staticmethod
    def from_dotlist(lst):
        return DictConfig(dict(s.split("=", 1) for s in lst))

    @staticmethod
    def to_container(c, **kw):
        return dict(c)

    @staticmethod
    def register_new_resolver(*a, **kw):
        pass

    @staticmethod
    def set_struct(*a, **kw):
        pass

    @staticmethod
    def resolve(*a, **kw):
        pass


@contextmanager
def open_dict(cfg):
    yield cfg
