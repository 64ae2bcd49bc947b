"""GPU: randomised parity -- random model shapes (l_max, n_max, blocks, cutoffs, scales) x random small batches against
the CPU oracle, same tolerances as the fixed cases (tests/checkers/fuzz_parity.py is the longer version of this sweep)."""
import numpy as np
import pytest
import torch

from helpers import random_cell_graph, rel_err
from oracle import m3gnet_oracle as orc
from test_gpu_properties import _oracle_inputs

pytestmark = pytest.mark.gpu


# cases 0-11 run on the MFMA kernels, whose arithmetic mode matters: the default exact-fp32 mode and the opt-in f16x3 mode are both
# swept; 12-19 take the any-size path, which is plain fp32 whatever the option says
@pytest.mark.parametrize("case,precision", [(c, "fp32") for c in range(20)] + [(c, "f16x3") for c in range(12)])
def test_random_model_shape_and_batch_vs_oracle(case, precision):
    """cases 0-11: shapes of the fast (MFMA) kernels; 12-19: any shape the reference accepts (widths up to 160, l_max up to 9,
    n_max up to 10, up to 10 blocks) -- the any-size path.  Energies at the strict 1e-5, forces / stresses at 1e-4."""
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.material_graph import Batch
    from torch_m3gnet.model.build import build_model

    rng = np.random.default_rng(7000 + case)
    if case < 12:
        l_max, n_max, dim, max_blocks = int(rng.integers(1, 5)), int(rng.integers(1, 5)), 64, 5
        if l_max * n_max > 16:
            n_max = 16 // l_max
    else:
        l_max, n_max, dim, max_blocks = int(rng.integers(1, 10)), int(rng.integers(1, 11)), int(rng.integers(8, 161)), 11
    cutoff = float(rng.uniform(3.5, 6.0))
    tb = float(rng.uniform(2.5, cutoff))
    torch.manual_seed(case)
    model = build_model(cutoff=cutoff, threebody_cutoff=tb, l_max=l_max, n_max=n_max, num_types=95, embedding_dim=dim,
                        num_blocks=int(rng.integers(1, max_blocks)), energy_scale=float(rng.uniform(0.5, 3.0)),
                        length_scale=float(rng.uniform(0.8, 1.5)))
    for m in model.model:  # documented chi so the three-body path carries weight
        if type(m).__name__ == "ThreeBodyInteration":
            m.nsb.factors = m.nsb.documented_factors()
    graphs = []
    for s in range(int(rng.integers(1, 5))):
        box = float(rng.uniform(4.5, 9.0))
        n = int(rng.integers(1, max(2, min(40, int(box**3 / 14.0)))))   # keeps the random packing feasible
        graphs.append(random_cell_graph(n, box, seed=1000 * case + s, cutoff=cutoff, tb_cutoff=tb, dmin=1.4))
    model.engine.set_precision(precision)
    g = model(Batch.from_data_list(graphs).to("cuda"))
    p, cfg, c, og = _oracle_inputs(model, g)
    o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
    e_err = float(((g[K.TOTAL_ENERGY].cpu() - o["total_energy"]).abs() / o["total_energy"].abs().clamp_min(1e-6)).max())
    assert e_err < 1e-5
    fmax = float(o["forces"].abs().max())
    assert float((g[K.FORCES].cpu() - o["forces"]).abs().max()) < 1e-4 * fmax + 1e-9
    if float(o["stresses"].abs().max()) > 0:
        assert rel_err(g[K.STRESSES], o["stresses"]) < 1e-4
