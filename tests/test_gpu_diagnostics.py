"""GPU: the stamped diagnostic variant of the fused reverse kernel (plan option "stamps" = 3, tools/stamp_report_fused.py) is the
shipped kernel plus s_memtime reads: it must leave energies, forces and stresses bit-identical, and every wave that worked must have
written non-zero phase sums."""
import numpy as np
import pytest
import torch

from helpers import fcc_cu_graph

pytestmark = pytest.mark.gpu


def test_stamped_fused_reverse_is_bit_identical_and_writes_stamps():
    from torch_m3gnet import _lib
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.model.build import build_model

    torch.manual_seed(0)
    model = build_model(5.0, 4.0, 3, 3, 95, 64, 3).cuda()
    model.engine.set_precision("f16x3")
    g = fcc_cu_graph(4, 4, 6).to("cuda")
    out = model(g)
    ref = {k: out[k].clone() for k in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES)}
    eng = model.engine
    eng.set_option("stamps", 3)
    try:
        out = model(g)
        torch.cuda.synchronize()
        for k, v in ref.items():
            assert torch.equal(out[k], v), k
        buf = np.zeros(256 * 16 * 12, dtype=np.uint64)
        _lib.check(eng.lib.m3g_debug_read_stamps(eng.plan, buf.ctypes.data))
        s = buf.reshape(256, 16, 12)
        assert s[:, :8].sum() > 0 and s[:, 8:].sum() == 0   # eight waves per workgroup
        busy = s[:, :8].sum(-1) > 0
        assert busy.any() and (s[:, :8][busy] > 0).all()    # a wave that processed a tile passed every phase mark
    finally:
        eng.set_option("stamps", 0)
    out = model(g)
    for k, v in ref.items():
        assert torch.equal(out[k], v), k
