class UsageError(Exception):
    pass


class CommError(Exception):
    pass
