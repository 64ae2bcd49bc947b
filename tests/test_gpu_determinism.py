"""GPU: run-to-run determinism of the forces.  The force path uses no atomics (per-centre sums are segmented scans and
CSR gathers), so repeated evaluations of one input must agree bit for bit; a deviation means a data hazard or a race in a
kernel (this test caught exactly that in an experimental build whose parity errors were still inside the tolerances most
of the time)."""
import pytest
import torch

from helpers import CASES, build_engine_model, engine_graph, fcc_cu_graph, load_oracle_case

pytestmark = pytest.mark.gpu
REPS = 25


def _K():
    from torch_m3gnet.data import MaterialGraphKey as K

    return K


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("case,mode", CASES)
def test_forces_bitwise_reproducible_on_golden_cases(case, mode, precision):
    K = _K()
    _, _, _, graph, _ = load_oracle_case(case, mode)
    model, _ = build_engine_model(case, mode)
    model.engine.set_precision(precision)
    g = engine_graph(graph)
    first = model(g)
    ref, ref_e, ref_s = first[K.FORCES].clone(), first[K.TOTAL_ENERGY].clone(), first[K.STRESSES].clone()
    for _ in range(REPS):
        out = model(g)
        assert torch.equal(out[K.FORCES], ref)
        # per-structure sums run without atomics too (sorted batch): energies and stresses are bit-reproducible as well
        assert torch.equal(out[K.TOTAL_ENERGY], ref_e) and torch.equal(out[K.STRESSES], ref_s)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x3"])
def test_forces_bitwise_reproducible_on_a_1500_atom_cell(precision):
    """Enough tiles that every workgroup of the persistent kernels is busy and waves overlap in every phase."""
    from torch_m3gnet.model.build import build_model

    K = _K()
    torch.manual_seed(0)
    model = build_model(5.0, 4.0, 3, 3, 95, 64, 3).cuda()
    model.engine.set_precision(precision)
    g = fcc_cu_graph(5, 5, 15).to("cuda")
    out = model(g)
    ref_f, ref_s = out[K.FORCES].clone(), out[K.STRESSES].clone()
    for _ in range(REPS):
        out = model(g)
        assert torch.equal(out[K.FORCES], ref_f)
        assert torch.equal(out[K.STRESSES], ref_s)   # fixed-order per-structure sums
