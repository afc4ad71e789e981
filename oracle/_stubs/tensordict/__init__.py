class TensorDict(dict):
    pass
