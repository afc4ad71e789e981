"""Import shim used ONLY by tests/golden/make_golden.py to import the reference offline.

Test infrastructure: minimal stand-ins for the Lightning names the reference touches at
import time and inside predict()/get_loss().  Not part of the product, never shipped to
the GPU path.  Surface derived from SURVEY.md Appendix D.
"""
import inspect
import random

import numpy as np
import torch
from torch import nn

from . import callbacks, loggers, utilities  # noqa: F401


class _HParams(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class LightningModule(nn.Module):
    def __init__(self, *a, **kw):
        super().__init__()
        self._hp = _HParams()
        self._trainer = None

    @property
    def hparams(self):
        return self._hp

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    @property
    def trainer(self):
        return self._trainer

    def save_hyperparameters(self, *args, ignore=None, **kw):
        # Like Lightning: gather the init args of every __init__ frame of THIS object up the stack
        # (child-class kwargs included), child frames first so parents do not clobber them.
        ignore = set(ignore or [])
        frame = inspect.currentframe().f_back
        collected = []
        while frame is not None:
            if frame.f_code.co_name == "__init__" and frame.f_locals.get("self") is self:
                info = inspect.getargvalues(frame)
                d = {}
                for name in info.args:
                    if name != "self":
                        d[name] = info.locals[name]
                if info.keywords and isinstance(info.locals.get(info.keywords), dict):
                    d.update(info.locals[info.keywords])
                collected.append(d)
            frame = frame.f_back
        for d in collected:
            for k, v in d.items():
                if k not in ignore and k not in ("args", "kwargs", "__class__"):
                    self._hp[k] = v

    def log(self, *a, **kw):
        pass

    def log_dict(self, *a, **kw):
        pass


class LightningDataModule:
    def __init__(self, *a, **kw):
        pass


class Trainer:
    def __init__(self, *a, **kw):
        pass


class Callback:
    pass


def seed_everything(seed, workers=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed
