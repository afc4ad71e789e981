class WandbLogger:
    def __init__(self, *a, **kw):
        pass
