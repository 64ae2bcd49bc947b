"""Stand-in for the absent `torchtyping` package: annotations only (test infrastructure)."""


class _TT:
    def __getitem__(self, item):
        return self

    def __or__(self, other):
        return self

    def __ror__(self, other):
        return self


TensorType = _TT()
