class InterpolationResolutionError(Exception):
    pass
