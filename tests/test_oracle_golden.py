"""CPU: the oracle (oracle/m3gnet_oracle.py) is pinned against golden vectors produced by running the
reference itself (tests/golden/generate_golden.py), stage by stage; the staged autograd-free
restatement (oracle/staged.py, the kernels' blueprint) is pinned against the oracle."""
import pytest
import torch

from helpers import CASES, GOLDEN, load_oracle_case, preactivation_stats, rel_err
from oracle import m3gnet_oracle as orc, staged

STAGE_KEYS = [
    ("edge_distances", "out_edge_distances"), ("triplet_angles", "out_triplet_angles"), ("edge_weights", "out_edge_weights"),
    ("x0", "mid_x0"), ("edge_attr0", "mid_edge_attr0"), ("mid_edge_features_0", "mid_mid_edge_features_0"),
    ("edge_attr_tb0", "mid_edge_attr_tb0"), ("edge_attr_conv0", "mid_edge_attr_conv0"), ("x_0", "mid_x_0"),
    ("x", "out_x"), ("edge_attr", "out_edge_attr"), ("scaled_atomic_energies", "out_scaled_atomic_energies"),
    ("scaled_total_energy", "out_scaled_total_energy"), ("total_energy", "out_total_energy"),
]


@pytest.mark.parametrize("case,mode", CASES)
def test_oracle_matches_reference_golden(case, mode):
    torch.set_num_threads(1)
    params, cfg, consts, graph, expect = load_oracle_case(case, mode)
    out = orc.energy_forces(params, cfg, consts, graph, legendre_backward="reference")
    for mine, theirs in STAGE_KEYS:
        # forward stages: same fp32 torch ops in the same order -> (near) bit-exact
        tol = 1e-6 if not (case == "alna" and mine.startswith("mid_edge")) else 1e-2
        assert rel_err(out[mine], expect[theirs]) < tol, mine
    assert rel_err(out["forces"], expect["out_forces"]) < 2e-5
    assert rel_err(out["stresses"], expect["out_stresses"]) < 5e-5


def test_constants_reproduce_reference_construction():
    """em/dm/coeff/factors as the reference builds them, and the recomputed Bessel-root table."""
    params, cfg, elemental = orc.load_model_npz(GOLDEN / "model_default_seed0.npz")
    _, expect = orc.load_case_npz(GOLDEN / "case_cu32_ref.npz")
    c = orc.make_constants(cfg, elemental)
    for name in ("em", "dm", "coeff"):
        assert torch.equal(getattr(c, name), expect[f"const_{name}"]), name
    # `factors` is fp32 rounding noise amplified (SURVEY finding 1): platform-dependent, so only its
    # order of magnitude is checked here; tests always inject the fixture's captured value.
    assert c.factors.shape == expect["const_factors"].shape
    assert float(c.factors.min()) > 1e3
    _, expect_doc = orc.load_case_npz(GOLDEN / "case_cu32_doc.npz")
    assert rel_err(orc.documented_factors(cfg), expect_doc["const_factors"]) < 1e-6


def test_bessel_zero_table_known_answer():
    """j_l(z_ln) = 0 for the whole 10x10 table (reference tests/test_basis.py:15-22)."""
    z = torch.tensor(orc.bessel_zeros(), dtype=torch.float32)
    for l in range(10):
        val = orc.spherical_bessel(z[l], l)
        torch.testing.assert_close(val, torch.zeros_like(val))


@pytest.mark.parametrize("order", [0, 1, 2, 3])
def test_special_function_gradients(order):
    """gradcheck of j_l and of the EXACT Legendre derivative in fp64 (reference tests/test_basis.py:25-42)."""
    x = torch.linspace(1e-1, 10, steps=16, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: orc.spherical_bessel(t, order), (x,), eps=1e-4)
    c = torch.linspace(-1, 1, steps=16, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: orc.legendre_cos(t, order, "exact"), (c,), eps=1e-4)


def test_cutoff_function_known_answer():
    """reference tests/test_basis.py:45-49"""
    torch.testing.assert_close(orc.cutoff_function(torch.tensor([0.0, 2.0, 4.0]), 2.0), torch.tensor([1.0, 0.0, 0.0]))


@pytest.mark.parametrize("case,mode", [("cu32", "doc"), ("alna", "ref"), ("mix", "doc")])
def test_staged_pipeline_equals_oracle_fp64(case, mode):
    """The hand-derived reverse pass equals autograd to fp64 rounding."""
    torch.set_num_threads(2)
    params, cfg, consts, graph, _ = load_oracle_case(case, mode, dtype=torch.float64)
    o = orc.energy_forces(params, cfg, consts, graph, legendre_backward="exact")
    s = staged.forward_backward(params, cfg, consts, graph)
    assert rel_err(s["total_energy"], o["total_energy"]) < 1e-12
    assert rel_err(s["forces"], o["forces"]) < 1e-11
    assert rel_err(s["stresses"], o["stresses"]) < 1e-11
    assert rel_err(s["x"], o["x"]) < 1e-12
    assert rel_err(s["edge_attr"], o["edge_attr"]) < 1e-12
    if case != "alna":
        assert rel_err(s["m_0"], o["mid_edge_features_0"]) < 1e-12


def test_forces_match_finite_differences_fp64():
    """Central differences on the fp64 oracle (reference tests/test_model.py:90-120, tighter)."""
    params, cfg, consts, graph, _ = load_oracle_case("alna", "doc", dtype=torch.float64)
    o = orc.energy_forces(params, cfg, consts, graph, legendre_backward="exact")
    delta = 1e-5
    for atom, axis in ((0, 0), (3, 2), (5, 1)):
        e = []
        for sign in (+1, -1):
            g2 = dict(graph)
            pos = graph["pos"].double().clone()
            pos[atom, axis] += sign * delta
            g2["pos"] = pos
            e.append(orc.energy_forces(params, cfg, consts, g2, want_forces=False)["total_energy"].sum())
        fd = -(e[0] - e[1]) / (2 * delta)
        assert abs(float(fd - o["forces"][atom, axis])) < 1e-7 * max(1.0, float(o["forces"].abs().max()))


def test_fitted_weights_work_outside_the_linear_part_of_their_activations():
    """Why the LJ-fitted fixture matters for parity: random-init weights keep every pre-activation below 0.2 (every MLP is
    near-linear, product errors average away); the fitted weights put 40 % of the first edge MLP's layer-2 pre-activations
    beyond |p| > 2 and 8 % beyond 4, with forces of ~2 eV/A."""
    params, cfg, consts, graph, expect = load_oracle_case("cu32fit", "doc")
    stats = preactivation_stats(params, cfg, consts, graph)
    d2, g2 = stats[2], stats[3]          # block 0, edge MLP, layer 2 (dense, gate)
    assert d2[0] == g2[0] == cfg.embedding_dim
    assert min(d2[2], g2[2]) > 0.35 and min(d2[3], g2[3]) > 0.05 and min(d2[1], g2[1]) > 1.5, (d2, g2)
    assert 1.0 < float(expect["out_forces"].abs().max()) < 5.0
    params, cfg, consts, graph, _ = load_oracle_case("cu32", "doc")
    assert max(st[4] for st in preactivation_stats(params, cfg, consts, graph)) < 0.3
