class DataArray:
    pass


class Dataset:
    pass
