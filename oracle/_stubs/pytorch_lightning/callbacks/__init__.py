class ModelCheckpoint:
    def __init__(self, *a, **kw):
        pass


class Callback:
    pass
