"""Stand-in for the absent third-party `torch_scatter` package (test infrastructure only).

Used ONLY by tests/golden/generate_golden.py, in the build container, so that the reference's
own nn code can be imported to produce golden vectors. Semantics restated from the published
torch-scatter API: out[..., index[i], ...] += src[..., i, ...] along `dim` (default -1).
"""
import torch


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    dim = dim if dim >= 0 else src.dim() + dim
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    res = torch.zeros(shape, dtype=src.dtype, device=src.device) if out is None else out
    return res.index_add(dim, index, src)
