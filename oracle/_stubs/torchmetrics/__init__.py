from torch import nn


class MeanSquaredError(nn.Module):
    def __init__(self, *a, **kw):
        super().__init__()


class Metric(nn.Module):
    pass
