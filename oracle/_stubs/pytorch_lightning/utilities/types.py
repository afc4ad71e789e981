from typing import Any

EVAL_DATALOADERS = Any
TRAIN_DATALOADERS = Any
