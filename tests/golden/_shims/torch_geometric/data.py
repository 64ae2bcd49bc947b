"""Minimal mapping-style containers so the reference nn modules can read/write keyed tensors."""


class Data(dict):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc


class Batch(Data):
    pass


class InMemoryDataset:
    pass
