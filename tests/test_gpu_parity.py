"""GPU parity: the HIP engine (through the torch_m3gnet module API and the C ABI) against
(1) golden vectors produced by the reference itself and (2) the CPU oracle on the same inputs.

Tolerances (BASELINE.json north_star / SURVEY.md §8(d)): energies 1e-5 relative; forces 1e-4 relative to
max|F|; three-body aggregate `mid_edge_features` 1e-4 relative in both `factors` modes.
In `doc` mode with l_max = 3 the reference's own autograd forces are off by up to ~1.1e-4 because of its
Legendre backward defect (SURVEY finding 2), so there the force check is against the oracle's exact
derivative, and against the golden forces only at 5e-4.
"""
import pytest
import torch

from helpers import CASES, build_engine_model, engine_graph, load_oracle_case, rel_err

pytestmark = pytest.mark.gpu

E_TOL, F_TOL, M_TOL = 1e-5, 1e-4, 1e-4


# engine option "precision": exact fp32 MFMA products (the default, the reference's arithmetic) / the two opt-in split modes:
# 3 bf16 split products / 3 scaled fp16 split products.  Every test below names the mode it runs in.
PRECISIONS = ["fp32", "bf16x3", "f16x3"]


def _record_margins(case, mode, precision, g, expect):
    """Measured errors against the reference's own numbers, kept with the run (gpurun_out/parity_margins.txt)."""
    import os

    from torch_m3gnet.data import MaterialGraphKey as K

    e_ref = expect["out_total_energy"]
    line = (f"{case}_{mode} {precision}: E {float(((g[K.TOTAL_ENERGY].cpu() - e_ref).abs() / e_ref.abs()).max()):.2e}  "
            f"F {rel_err(g[K.FORCES], expect['out_forces']):.2e} of max|F| = {float(expect['out_forces'].abs().max()):.3e}  "
            f"stress {rel_err(g[K.STRESSES], expect['out_stresses']):.2e}  x {rel_err(g[K.NODE_FEATURES], expect['out_x']):.2e}  "
            f"edge_attr {rel_err(g[K.EDGE_ATTR], expect['out_edge_attr']):.2e}")
    print(line)
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/parity_margins.txt", "a") as fh:
            fh.write(line + "\n")


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case,mode", CASES)
def test_engine_vs_golden_and_oracle(case, mode, precision):
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data import MaterialGraphKey as K

    params, cfg, consts, graph, expect = load_oracle_case(case, mode)
    model, _ = build_engine_model(case, mode)
    model.engine.set_precision(precision)
    g = model(engine_graph(graph))
    torch.cuda.synchronize()
    _record_margins(case, mode, precision, g, expect)

    # ---- against the reference's own numbers
    assert rel_err(g[K.EDGE_DISTANCES], expect["out_edge_distances"]) < 1e-6
    assert float((g[K.TRIPLET_ANGLES].cpu() - expect["out_triplet_angles"]).abs().max()) < 2e-6
    assert rel_err(g[K.EDGE_WEIGHTS], expect["out_edge_weights"]) < 1e-5
    assert rel_err(g[K.NODE_FEATURES], expect["out_x"]) < 1e-5
    assert rel_err(g[K.EDGE_ATTR], expect["out_edge_attr"]) < 1e-5
    assert rel_err(g[K.SCALED_ATOMIC_ENERGIES], expect["out_scaled_atomic_energies"]) < E_TOL
    e_ref = expect["out_total_energy"]
    assert float(((g[K.TOTAL_ENERGY].cpu() - e_ref).abs() / e_ref.abs()).max()) < E_TOL
    if case != "alna":  # alna's neighbours sit on the three-body cutoff: m is ~1e-17, pure rounding
        for b in range(cfg.num_blocks):
            key = f"mid_mid_edge_features_{b}"
            if key in expect:
                assert rel_err(g[K.MID_EDGE_FEATURES][b], expect[key]) < M_TOL, (b, "mid_edge_features")
    # ---- the oracle's exact derivative (fp64 restatement of the same math)
    p64, cfg64, c64, graph64, _ = load_oracle_case(case, mode, dtype=torch.float64)
    o = orc.energy_forces(p64, cfg64, c64, graph64, legendre_backward="exact")
    # forces / stresses against the reference's numbers.  `ref` mode: directly, at north_star's tolerance.  `doc` mode (three-body
    # path visible): the reference's own autograd forces are OFF the true gradient by its Legendre-backward defect (SURVEY
    # finding 2; 1.1e-4 on mix_doc, 1.9e-3 on mixfit_doc whose fitted weights give the three-body path real weight) -- the engine
    # computes the exact derivative, so the gate is that measured defect plus the tolerance (and never more than 5e-3)
    defect_f = rel_err(expect["out_forces"], o["forces"]) if mode == "doc" else 0.0
    defect_s = rel_err(expect["out_stresses"], o["stresses"]) if mode == "doc" else 0.0
    assert defect_f < 5e-3 and defect_s < 5e-3
    assert rel_err(g[K.FORCES], expect["out_forces"]) < F_TOL + defect_f
    assert rel_err(g[K.STRESSES], expect["out_stresses"]) < 5e-4 + defect_s

    # ---- against the oracle's exact derivative
    assert rel_err(g[K.FORCES], o["forces"]) < F_TOL
    assert float(((g[K.TOTAL_ENERGY].cpu().double() - o["total_energy"]).abs() / o["total_energy"].abs()).max()) < E_TOL
    assert rel_err(g[K.STRESSES], o["stresses"]) < F_TOL
    if case != "alna":
        for b in range(cfg.num_blocks):
            assert rel_err(g[K.MID_EDGE_FEATURES][b], o[f"mid_edge_features_{b}"]) < M_TOL
    import os
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/parity_margins.txt", "a") as fh:
            fh.write(f"    vs fp64 oracle (exact derivative): F {rel_err(g[K.FORCES], o['forces']):.2e}  stress {rel_err(g[K.STRESSES], o['stresses']):.2e}  "
                     f"E {float(((g[K.TOTAL_ENERGY].cpu().double() - o['total_energy']).abs() / o['total_energy'].abs()).max()):.2e}"
                     f"   [reference's own forces vs the exact derivative: {defect_f:.2e}]\n")


@pytest.mark.parametrize("precision", PRECISIONS)
def test_energy_only_call_matches(precision):
    from torch_m3gnet.data import MaterialGraphKey as K

    _, _, _, graph, expect = load_oracle_case("cu32", "ref")
    model, _ = build_engine_model("cu32", "ref")
    model.engine.set_precision(precision)
    g = model(engine_graph(graph), forces=False, extras=False)
    assert K.FORCES not in g
    assert rel_err(g[K.TOTAL_ENERGY], expect["out_total_energy"]) < E_TOL


@pytest.mark.parametrize("edge_kernel,precision", [(0, "fp32"), (1, "fp32"), (1, "f16x3"), (1, "bf16x3")])
@pytest.mark.parametrize("case,mode", [("cu32", "doc"), ("alna", "ref"), ("mix", "doc")])
def test_both_edge_kernels_against_oracle(case, mode, edge_kernel, precision):
    """edge_kernel = 1: fused MFMA edge blocks (default) in each arithmetic mode; 0: vector-ALU baseline kernels (plain fp32
    `fmaf` arithmetic whatever the precision option says)."""
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data import MaterialGraphKey as K

    model, _ = build_engine_model(case, mode)
    model.engine.set_precision(precision)
    model.engine.set_option("edge_kernel", edge_kernel)
    _, _, _, graph, _ = load_oracle_case(case, mode)
    g = model(engine_graph(graph))
    p64, cfg64, c64, graph64, _ = load_oracle_case(case, mode, dtype=torch.float64)
    o = orc.energy_forces(p64, cfg64, c64, graph64, legendre_backward="exact")
    assert float(((g[K.TOTAL_ENERGY].cpu().double() - o["total_energy"]).abs() / o["total_energy"].abs()).max()) < E_TOL
    assert rel_err(g[K.FORCES], o["forces"]) < F_TOL
    assert rel_err(g[K.EDGE_ATTR], o["edge_attr"]) < 1e-5
    assert rel_err(g[K.NODE_FEATURES], o["x"]) < 1e-5


@pytest.mark.parametrize("case,mode", [("cu32", "doc"), ("mix", "doc"), ("tri", "doc"), ("mixfit", "doc"), ("cu32fit", "ref")])
def test_threebody_moment_and_list_paths_agree(case, mode):
    """Option "threebody_moments" (default 1): where a centre's partner lists are complete -- every fixture the reference's
    compute_threebody produced is -- the three-body sums run over per-atom moments instead of walking the lists
    (csrc/m3g_threebody.hip).  Both paths against the fp64 oracle at north_star's tolerances, and against each other far below
    them; the two sum in different orders, so bit-identical aggregates would mean the moment path never ran."""
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data import MaterialGraphKey as K

    _, cfg, _, graph, _ = load_oracle_case(case, mode)
    p64, cfg64, c64, graph64, _ = load_oracle_case(case, mode, dtype=torch.float64)
    o = orc.energy_forces(p64, cfg64, c64, graph64, legendre_backward="exact")
    model, _ = build_engine_model(case, mode)
    model.engine.set_precision("fp32")
    out = {}
    eg = engine_graph(graph)
    for moments in (1, 0):
        model.engine.set_option("threebody_moments", moments)
        g = model(eg)
        torch.cuda.synchronize()
        out[moments] = {k: g[k].detach().clone() for k in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES, K.MID_EDGE_FEATURES)}
        assert float(((g[K.TOTAL_ENERGY].cpu().double() - o["total_energy"]).abs() / o["total_energy"].abs()).max()) < E_TOL
        assert rel_err(g[K.FORCES], o["forces"]) < F_TOL
        assert rel_err(g[K.STRESSES], o["stresses"]) < F_TOL
        for b in range(cfg.num_blocks):
            assert rel_err(g[K.MID_EDGE_FEATURES][b], o[f"mid_edge_features_{b}"]) < M_TOL
    a, b = out[1], out[0]
    assert not torch.equal(a[K.MID_EDGE_FEATURES], b[K.MID_EDGE_FEATURES]), "the moment path did not run"
    assert rel_err(a[K.MID_EDGE_FEATURES], b[K.MID_EDGE_FEATURES]) < 5e-6
    assert rel_err(a[K.FORCES], b[K.FORCES]) < 1e-5
    assert rel_err(a[K.STRESSES], b[K.STRESSES]) < 1e-5
    assert float(((a[K.TOTAL_ENERGY] - b[K.TOTAL_ENERGY]).abs() / b[K.TOTAL_ENERGY].abs()).max()) < 2e-6


# (rev_kernel, save_p1, save_p2): every reverse-kernel instantiation the public options select.  fp32 mode:
#   (1,1,1) k_edge_rev_f32<SAVED_P2=true>  on SiLU'(p1) + p2 saved by the forward kernel (SAVE = 2)      -- the default
#   (1,1,0) k_edge_rev_f32<SAVED_P2=false> on raw p1 (SAVE = 1), layer 2 recomputed
#   (0,1,0) k_edge_rev_node_mlp / k_edge_rev_edge_mlp<fp32, SAVED=true> on raw p1 (SAVE = 1)
#   (1,0,0), (0,0,0) the same pair recomputing both layers (SAVE = 0)
# bf16x3 mode (nothing is ever saved): (1,*,*) k_edge_rev_fused, (0,*,*) the kernel pair.
_REV_VARIANTS = {"fp32": [(1, 1, 1), (1, 1, 0), (1, 0, 0), (0, 1, 0), (0, 0, 0)], "bf16x3": [(1, 1, 1), (0, 1, 1)], "f16x3": [(1, 1, 1), (0, 1, 1)]}


@pytest.mark.parametrize("precision", PRECISIONS)
def test_fused_and_split_reverse_kernels_agree(precision):
    """The forward kernel's SAVE mode and the reverse kernel that consumes it are chosen from the same three options; a
    mismatch (raw p1 read as SiLU'(p1), a workspace carved for another mode) would give silently wrong forces.  Every
    combination must give the same energies (1e-6) and forces (2e-6 of max|F|: same math, different summation order),
    and the workspace the engine allocates for a mode must be the size m3g_workspace_bytes reports for it."""
    import ctypes as C

    from torch_m3gnet import _lib
    from torch_m3gnet.nn.modules import _Topology

    case, mode = "tio", "doc"
    params, cfg, consts, graph, expect = load_oracle_case(case, mode)
    model, _ = build_engine_model(case, mode)
    model = model.cuda()
    eng = model.engine
    eng.set_precision(precision)
    outs, sizes = [], []
    try:
        for rk, s1, s2 in _REV_VARIANTS[precision]:
            eng.set_option("rev_kernel", rk)
            eng.set_option("save_p1", s1)
            eng.set_option("save_p2", s2)
            eng._workspace = None          # a fresh carve per mode: the engine must size it itself
            g = engine_graph(graph)
            o = model(g)
            torch.cuda.synchronize()
            topo = _Topology.of(g)
            nbytes = C.c_size_t()
            _lib.check(eng.lib.m3g_workspace_bytes(eng.plan, topo.N, topo.E, topo.T, topo.S, C.byref(nbytes)))
            assert eng._workspace.numel() == nbytes.value
            sizes.append(nbytes.value)
            outs.append((o["total_energy"].clone(), o["forces"].clone(), o["stresses"].clone()))
    finally:
        eng.set_option("rev_kernel", 1)
        eng.set_option("save_p1", 1)
        eng.set_option("save_p2", 1)
    for (e, f, sg), variant in zip(outs[1:], _REV_VARIANTS[precision][1:]):
        assert rel_err(e, outs[0][0]) < 1e-6, variant
        assert rel_err(f, outs[0][1]) < 2e-6, variant
        assert rel_err(sg, outs[0][2]) < 5e-6, variant
    if precision == "fp32":   # saved activations cost workspace: 2 arrays > 1 array > none
        assert sizes[0] > sizes[1] > sizes[2] and sizes[1] == sizes[3] and sizes[2] == sizes[4]
    # and the default of this mode still meets the north_star gates against the reference's numbers
    assert rel_err(outs[0][0], expect["out_total_energy"]) < E_TOL
    assert rel_err(outs[0][1], expect["out_forces"]) < 5e-4


def test_commit_under_another_device_releases_stream_and_events():
    """m3g_plan_commit moves a plan to the device that is current (model.to('cuda:1') after a call on cuda:0).  Streams and
    events are bound to the device they were created under, so the move must release the internal side stream, its fork /
    join events and the profiler's event pool, not only the weight buffers.  Driven on one GPU through the test hook
    `debug_force_move` (the commit takes the move path although the device is unchanged); afterwards every handle is gone
    and the next calls recreate them and return bit-identical results."""
    import ctypes as C

    from torch_m3gnet import _lib

    case, mode = "cu32", "ref"
    _, _, _, graph, _ = load_oracle_case(case, mode)
    model, _ = build_engine_model(case, mode)
    model = model.cuda()
    eng = model.engine
    g = engine_graph(graph)

    def handles():
        n = C.c_int32()
        _lib.check(eng.lib.m3g_debug_live_handles(eng.plan, C.byref(n)))
        return n.value

    eng.set_option("overlap", 1)      # side stream + fork / join events (not used while the profiler is on)
    ref = model(g)
    e0, f0 = ref["total_energy"].clone(), ref["forces"].clone()
    assert handles() == 3
    eng.profile(True)                 # event pool
    model(g)
    eng.profile_read()
    assert handles() > 3
    eng.set_option("debug_force_move", 1)
    eng._sig = None                   # forces Engine.run to commit again
    out = model(g)                    # commit (move path) + call: handles were released, then recreated lazily
    assert torch.equal(out["total_energy"], e0) and torch.equal(out["forces"], f0)
    eng.profile(False)
    eng.set_option("overlap", 0)
    eng.set_option("debug_force_move", 1)
    eng.commit()
    assert handles() == 0             # nothing survives a move; nothing is recreated until it is needed
    eng._sig = None
    out = model(g)
    assert torch.equal(out["total_energy"], e0)


def test_side_stream_overlap_option_gives_identical_results():
    """Option "overlap" = 1 runs each block's three-body reverse on an internal side stream beside the node reverse's
    gather (fork/join by events): same kernels, same order of every sum -> bitwise identical energies and forces."""
    case, mode = "mix", "doc"
    params, cfg, consts, graph, expect = load_oracle_case(case, mode)
    model, _ = build_engine_model(case, mode)
    model = model.cuda()
    outs = []
    for ov in (0, 1, 1):
        model.engine.set_option("overlap", ov)
        o = model(engine_graph(graph))
        outs.append((o["total_energy"].clone(), o["forces"].clone()))
    model.engine.set_option("overlap", 0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[1][0], outs[2][0])
    assert rel_err(outs[1][1], outs[0][1]) < 1e-6 and torch.equal(outs[1][1], outs[2][1])


def test_graph_replay_gives_identical_results_and_tracks_inputs():
    """Option "graph_replay": the launch sequence is captured into a hipGraph on the first call and replayed afterwards.
    Same kernels, same order -> bitwise identical outputs; positions updated IN PLACE are picked up by the replay."""
    case, mode = "cu32", "doc"
    params, cfg, consts, graph, expect = load_oracle_case(case, mode)
    model, _ = build_engine_model(case, mode)
    model = model.cuda()
    g = engine_graph(graph)
    ref = model(g)
    e0, f0 = ref["total_energy"].clone(), ref["forces"].clone()
    model.engine.set_option("graph_replay", 1)
    try:
        for _ in range(3):   # capture, then two replays
            out = model(g)
            assert torch.equal(out["total_energy"], e0) and torch.equal(out["forces"], f0)
        g["pos"].add_(0.01 * torch.randn_like(g["pos"]))   # same storage, new values
        moved = model(g)
        e1, f1 = moved["total_energy"].clone(), moved["forces"].clone()
        assert not torch.equal(e1, e0)
    finally:
        model.engine.set_option("graph_replay", 0)
    plain = model(g)
    assert torch.equal(plain["total_energy"], e1) and torch.equal(plain["forces"], f1)


def _scaled_like_trained(model, w_scale=4.0, b_scale=2.0):
    """Random-init weights keep every MLP near-linear.  Scale all Linear weights x4 and biases x2 so that SiLU and sigmoid
    saturate the way they do in a trained potential (pre-activations of several units)."""
    with torch.no_grad():
        for name, p in model.model.named_parameters():
            p.mul_(b_scale if name.endswith("bias") else w_scale)
    return model


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", ["cu32", "mix"])
def test_saturated_activations_stress_case(case, precision):
    """Parity with trained-like weight magnitudes (x4 weights, x2 biases, energy_scale 10) against the fp64 oracle.
    fp32 mode (default): north_star's tolerances, energies 1e-5 and forces 1e-4 of max|F|.  bf16x3 mode: its ~2^-16 product
    error no longer hides behind near-linear MLPs -- measured 1.7e-4 on the energy of the Cu cell, 9e-5 on the forces -- so it
    is gated at 1e-3 / 5e-4 here and documented as the fast reduced-accuracy mode (precision table)."""
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.model.build import build_model

    params, cfg, consts, graph, expect = load_oracle_case(case, "doc", dtype=torch.float64)
    torch.manual_seed(5)
    model = build_model(cfg.cutoff, cfg.threebody_cutoff, cfg.l_max, cfg.n_max, cfg.num_types, cfg.embedding_dim, cfg.num_blocks,
                        elemental_energies=consts.elemental_energies.float(), energy_scale=10.0, length_scale=cfg.length_scale)
    model = _scaled_like_trained(model)
    for m in model.model:
        if type(m).__name__ == "ThreeBodyInteration":
            m.nsb.factors = expect["const_factors"].float().clone()
    model.engine.set_precision(precision)
    g = model(engine_graph(graph))
    cfg.energy_scale = 10.0
    p64 = {f"model.{k}": v.detach().cpu().double() for k, v in model.model.state_dict().items()}
    o = orc.energy_forces(p64, cfg, consts, graph, legendre_backward="exact")
    # what "trained-like" means here, measured and asserted (random-init: every |p| < 0.2): in the LAST block the second-layer
    # pre-activations have mean |p| ~ 1 or more, more than a tenth of them lie beyond |p| > 2 and the largest beyond 3.5 -- SiLU and
    # sigmoid work well outside their linear part.  (The LJ-fitted fixture cu32fit goes further for physical reasons: 45 % of
    # the first edge MLP's layer-2 pre-activations beyond 2, 10 % beyond 4 -- tests/test_oracle_golden.py.)
    from helpers import preactivation_stats

    stats = preactivation_stats(p64, cfg, consts, graph)
    last_block_l2 = [st for st in stats if st[0] == cfg.embedding_dim][-8:-4]   # the 4 layer-2 Linears of the last conv block
    shares = [st[2] for st in last_block_l2]
    assert min(st[1] for st in last_block_l2) > 0.9 and min(shares) > 0.1 and max(st[4] for st in last_block_l2) > 3.5, last_block_l2
    e_err = float(((g[K.TOTAL_ENERGY].cpu().double() - o["total_energy"]).abs() / o["total_energy"].abs()).max())
    f_err = rel_err(g[K.FORCES], o["forces"])
    s_err = rel_err(g[K.STRESSES], o["stresses"])
    line = (f"stress case {case} {precision}: E rel err {e_err:.2e}, F err {f_err:.2e} of max|F| = {float(o['forces'].abs().max()):.3e}, "
            f"stress err {s_err:.2e}; last block, layer-2 pre-activations beyond |p| > 2: {min(shares):.2f}-{max(shares):.2f}")
    print(line)
    import os
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/stress_case_margins.txt", "a") as fh:
            fh.write(line + "\n")
    e_tol, f_tol = (E_TOL, F_TOL) if precision != "bf16x3" else (1e-3, 5e-4)   # fp32 and f16x3: north_star's tolerances
    assert e_err < e_tol and f_err < f_tol
    assert s_err < (5e-4 if precision != "bf16x3" else 2e-3)


def test_single_pair_triplet_list_against_the_reference_entry_by_entry():
    """A triplet list of ONE pair leaves nothing to average over: `mid_edge_features` of block 0 is a single fp32
    Bessel x Legendre x envelope x sigmoid product per channel.  Compared entry by entry with what the REFERENCE computes for the
    same input (fixture case_cu32pair_doc, generated by its own nn code) at 1e-5 of the aggregate's scale (measured 2-3e-7); the fp64
    oracle is the second witness (the reference's own fp32 numbers sit 8e-7 from it)."""
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data import MaterialGraphKey as K

    params, cfg, consts, graph, expect = load_oracle_case("cu32pair", "doc")
    assert graph["triplet_edge_index"].shape == (2, 1)
    model, _ = build_engine_model("cu32pair", "doc")
    for precision in PRECISIONS:
        model.engine.set_precision(precision)
        g = model(engine_graph(graph))
        mid = g[K.MID_EDGE_FEATURES][0].cpu()
        ref = expect["mid_mid_edge_features_0"]
        e1 = int(graph["triplet_edge_index"][0, 0])
        others = torch.ones(ref.size(0), dtype=torch.bool)
        others[e1] = False
        assert float(ref[e1].abs().max()) > 0 and float(ref[others].abs().max()) == 0.0   # one live row
        scale = float(ref.abs().max())
        worst = float((mid - ref).abs().max()) / scale
        p64, cfg64, c64, graph64, _ = load_oracle_case("cu32pair", "doc", dtype=torch.float64)
        o = orc.energy_forces(p64, cfg64, c64, graph64, legendre_backward="exact")
        ref_vs_64 = float((ref.double() - o["mid_edge_features_0"]).abs().max()) / scale
        mine_vs_64 = float((mid.double() - o["mid_edge_features_0"]).abs().max()) / scale
        print(f"single pair {precision}: engine vs reference {worst:.2e}, reference vs fp64 {ref_vs_64:.2e}, engine vs fp64 {mine_vs_64:.2e}")
        assert worst < 1e-5, (precision, worst, ref_vs_64)   # measured 2-3e-7; the reference itself is 8e-7 from fp64
        assert mine_vs_64 < 1e-5
        assert rel_err(g[K.FORCES], expect["out_forces"]) < F_TOL
        assert rel_err(g[K.TOTAL_ENERGY], expect["out_total_energy"]) < E_TOL


@pytest.mark.parametrize("energy_scale,w_scale,outlier", [(1e-6, 1.0, 0.0), (1e6, 1.0, 0.0), (1.0, 4.0, 0.0), (1.0, 1.0, 60.0), (3e-4, 4.0, 25.0)])
def test_f16x3_range_handling_tracks_the_exact_mode(energy_scale, w_scale, outlier):
    """The f16x3 mode represents every operand as two fp16 parts of a power-of-two-scaled value; fp16 has 5 exponent bits.  What
    keeps that safe is the per-edge scale taken from the data and the per-model weight scale -- exercised here where a fixed
    scale would overflow or flush: gradients of 1e-9 and of 1e+6 (energy_scale 1e-6 / 1e6 multiplies the whole reverse pass),
    saturated activations (weights x4; at x5.5 this random model is chaotic -- max|F| = 3e4 eV/A, node features of the two modes
    7e-6 apart -- and at x8 its forces underflow to exactly zero in the exact mode as well), and single weights 25-60 x larger than the rest of their
    matrix (the model's weight scale is set by the largest, the others must keep their precision).  The exact-fp32 mode of the
    same engine is the witness: energies to 1e-5, forces to 5e-5 of max|F|, stresses to 1e-4 (measured: 2e-7 .. 2e-6 on the well-conditioned
    cases); a range failure would show as inf / NaN or errors of order one."""
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.model.build import build_model

    params, cfg, consts, graph, expect = load_oracle_case("mix", "doc")
    torch.manual_seed(11)
    model = build_model(cfg.cutoff, cfg.threebody_cutoff, cfg.l_max, cfg.n_max, cfg.num_types, cfg.embedding_dim, cfg.num_blocks,
                        elemental_energies=consts.elemental_energies.float(), energy_scale=energy_scale, length_scale=cfg.length_scale)
    with torch.no_grad():
        for name, p in model.model.named_parameters():
            if not name.endswith("bias"):
                p.mul_(w_scale)
        if outlier:
            for name, p in model.model.named_parameters():
                if name.endswith("dense.2.weight") or name.endswith("gate.0.weight"):
                    p.view(-1)[7] = outlier * float(p.abs().mean())
                    p.view(-1)[-3] = -outlier * float(p.abs().mean())
    for m in model.model:
        if type(m).__name__ == "ThreeBodyInteration":
            m.nsb.factors = expect["const_factors"].clone()
    outs = {}
    for precision in ("fp32", "f16x3"):
        model.engine.set_precision(precision)
        g = model(engine_graph(graph))
        outs[precision] = {k: g[k].clone() for k in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES, K.NODE_FEATURES, K.EDGE_ATTR)}
    a, b = outs["fp32"], outs["f16x3"]
    assert torch.isfinite(b[K.FORCES]).all() and torch.isfinite(b[K.TOTAL_ENERGY]).all()
    assert float(a[K.FORCES].abs().max()) > 0
    e_err = float(((a[K.TOTAL_ENERGY] - b[K.TOTAL_ENERGY]).abs() / a[K.TOTAL_ENERGY].abs()).max())
    f_err, s_err = rel_err(b[K.FORCES], a[K.FORCES]), rel_err(b[K.STRESSES], a[K.STRESSES])
    print(f"f16x3 vs fp32 mode (energy_scale {energy_scale:g}, weights x{w_scale:g}, outlier {outlier:g}): E {e_err:.1e} F {f_err:.1e} of max|F| = "
          f"{float(a[K.FORCES].abs().max()):.2e}, stress {s_err:.1e}, x {rel_err(b[K.NODE_FEATURES], a[K.NODE_FEATURES]):.1e}")
    assert e_err < 1e-5 and f_err < 5e-5 and s_err < 1e-4
    assert rel_err(b[K.NODE_FEATURES], a[K.NODE_FEATURES]) < 1e-5 and rel_err(b[K.EDGE_ATTR], a[K.EDGE_ATTR]) < 1e-5


# ---- option "legendre_backward" = 1: the reference's OWN forces (SURVEY finding 2) ------------------------------------------------
REF_LEGENDRE_F_TOL = 1e-5   # of max|F|: no defect allowance any more (measured: gpurun_out/parity_margins.txt)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case,mode", CASES)
def test_reference_legendre_backward_reproduces_the_references_forces(case, mode, precision):
    """The reference's LegendreCosPolynomial.backward (nn/interaction.py:373-382) is not the derivative for l >= 2: its forces are
    off the gradient of its own energy by 1.1e-4 (mix_doc) .. 1.9e-3 (mixfit_doc) of max|F|.  With the option set the engine
    returns what the reference returns: forces and stresses against the golden vectors WITHOUT the defect allowance of
    test_engine_vs_golden_and_oracle, and against the fp64 oracle running the same defective backward."""
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data import MaterialGraphKey as K

    _, _, _, graph, expect = load_oracle_case(case, mode)
    model, _ = build_engine_model(case, mode)
    model.engine.set_precision(precision)
    model.engine.set_option("threebody_moments", 0)   # the option runs the list kernels: the same forward kernels for both runs
    exact = model(engine_graph(graph))
    f_exact, e_exact = exact[K.FORCES].clone(), exact[K.TOTAL_ENERGY].clone()
    model.engine.set_option("legendre_backward", 1)
    g = model(engine_graph(graph))
    # cu32 is the perfect fcc crystal: its forces are the rounding residue of terms that cancel (max|F| ~ 8e-5 eV/A)
    tol = (3e-5 if case == "cu32" else REF_LEGENDRE_F_TOL) if precision == "fp32" else F_TOL
    assert torch.equal(g[K.TOTAL_ENERGY], e_exact)   # the forward pass is untouched
    assert rel_err(g[K.FORCES], expect["out_forces"]) < tol
    assert rel_err(g[K.STRESSES], expect["out_stresses"]) < 10 * tol
    p64, cfg64, c64, graph64, _ = load_oracle_case(case, mode, dtype=torch.float64)
    o = orc.energy_forces(p64, cfg64, c64, graph64, legendre_backward="reference")
    assert rel_err(g[K.FORCES], o["forces"]) < tol
    assert rel_err(g[K.STRESSES], o["stresses"]) < 10 * tol
    import os
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/parity_margins.txt", "a") as fh:
            fh.write(f"{case}_{mode} {precision} legendre_backward=1: F vs the reference {rel_err(g[K.FORCES], expect['out_forces']):.2e}, vs the fp64 "
                     f"oracle with the reference's backward {rel_err(g[K.FORCES], o['forces']):.2e}; stress {rel_err(g[K.STRESSES], expect['out_stresses']):.2e}"
                     f"   [exact-derivative forces vs the reference: {rel_err(f_exact, expect['out_forces']):.2e}]\n")
    # and back: the option is a switch, not a state
    model.engine.set_option("legendre_backward", 0)
    again = model(engine_graph(graph))
    assert torch.equal(again[K.FORCES], f_exact)


@pytest.mark.parametrize("kern", [0, 1, 2])
def test_reference_legendre_backward_l_max_4(kern):
    """l_max = 4: the reference's recurrence nests grad_output twice (order 3) -- every engine (MFMA kernels, vector-ALU baseline,
    any-size path) against the fp64 oracle running the reference's backward."""
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.material_graph import Batch
    from torch_m3gnet.model.build import build_model
    from helpers import random_cell_graph
    from test_gpu_properties import _oracle_inputs

    torch.manual_seed(5)
    model = build_model(5.0, 4.0, 4, 3, 60, 32, 2, elemental_energies=torch.linspace(-1, 1, 60), energy_scale=1.3)
    for m in model.model:
        if type(m).__name__ == "ThreeBodyInteration":
            m.nsb.factors = m.nsb.documented_factors()
    # weights large enough that grad_output of P_l is not negligible beside 1 (the defect is go-dependent)
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.5)
    cells = [random_cell_graph(12 + 4 * s, 5.5 + 0.4 * s, 30 + s, cutoff=5.0, tb_cutoff=4.0, zmax=59) for s in range(2)]
    model.engine.set_option("edge_kernel", kern)
    model.engine.set_option("legendre_backward", 1)
    out = model(Batch.from_data_list([c.clone() for c in cells]).to("cuda"))
    p, cfg, c, og = _oracle_inputs(model, out)
    p = {k: v.double() for k, v in p.items()}
    c = orc.make_constants(cfg, model.model[1].elemental_energies.cpu(), dtype=torch.float64)
    c.factors = model.model[6].nsb.factors.double()
    ref = orc.energy_forces(p, cfg, c, og, legendre_backward="reference")
    exact = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
    defect = rel_err(ref["forces"], exact["forces"])
    err = rel_err(out[K.FORCES], ref["forces"])
    assert err < 1e-5, (kern, err, defect)
    assert rel_err(out[K.STRESSES], ref["stresses"]) < 1e-4
    assert defect > 10 * err, (kern, err, defect)   # the test distinguishes the two backwards
