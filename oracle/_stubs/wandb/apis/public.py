class Run:
    pass
