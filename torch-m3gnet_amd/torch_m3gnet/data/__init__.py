from . import MaterialGraphKey  # noqa: F401
