import importlib


def get_class(path):
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(cfg, *args, _recursive_=False, **kwargs):
    params = {k: v for k, v in dict(cfg).items() if k not in ("_target_", "_recursive_")}
    params.update(kwargs)
    return get_class(cfg["_target_"])(*args, **params)
