"""GPU: the device-resident trajectory graph (torch_m3gnet.data.md.VerletGraph over the C ABI's m3g_verlet_*).

The contract: for ANY positions, `VerletGraph.update(pos)` is the graph a fresh build at those positions returns -- same
edges, order, shifts and triplets -- and the engine's results on it are BIT-identical to the fresh build's, whichever of its
three paths (reuse / refill from the skin list / new search) produced it.  Reference behaviour preserved: the edge set and
the triplet set of data/material_graph.py:168-254 (everything within the cutoffs, rebuilt from the positions alone)."""
import numpy as np
import pytest
import torch

from helpers import random_cell_arrays

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _K():
    from torch_m3gnet.data import MaterialGraphKey as K

    return K


def _model():
    from torch_m3gnet.model.build import build_model

    torch.manual_seed(0)
    model = build_model(cutoff=5.0, threebody_cutoff=4.0, l_max=3, n_max=3, num_types=95, embedding_dim=64, num_blocks=3)
    for m in model.model:
        if type(m).__name__ == "ThreeBodyInteration":
            m.nsb.factors = m.nsb.documented_factors()
    return model


def _same_graph(a, b):
    K = _K()
    for key in (K.EDGE_INDEX, K.EDGE_CELL_SHIFT, K.TRIPLET_EDGE_INDEX, K.NUM_TRIPLET_I, K.NUM_TRIPLET_IJ, K.BATCH, K.ATOM_TYPES):
        assert a[key].shape == b[key].shape and torch.equal(a[key], b[key]), key
    assert torch.equal(a[K.POS], b[K.POS]) and torch.equal(a[K.LATTICE], b[K.LATTICE])


def test_edge_order_does_not_depend_on_which_side_of_a_cell_face_an_atom_sits():
    """The canonical order sorts a centre's edges by the shift relative to the GIVEN coordinates: an atom that drifts across a
    cell face (fractional coordinate -0.002 -> +0.002) changes neither the shifts nor the order of any edge -- the property the
    skin list rests on.  Host builder (torch_m3gnet/data/neighbors.py) and GPU builder alike, element by element."""
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.data.material_graph import MaterialGraph

    K = _K()
    a = 3.61
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    gi = np.stack(np.meshgrid(np.arange(2), np.arange(2), np.arange(2), indexing="ij"), -1)
    pos = (gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a + np.random.default_rng(3).uniform(0.03, 0.06, (32, 3))
    lat = np.eye(3) * 2 * a
    z = np.full(32, 29)
    graphs = []
    for dx in (-0.002, +0.002):
        p = pos.copy()
        p[0] = [dx * 2 * a, 0.04, -0.001 if dx < 0 else 0.001]   # the atom at the origin, just outside / just inside two faces
        p[5, 1] += 2 * a                                          # and one atom a whole cell away from home in both builds
        graphs.append((batch_from_arrays([lat], [p], [z], 5.0, 4.0, device=DEV), MaterialGraph.from_arrays(lat, p, z, 5.0, 4.0)))
    (g0, h0), (g1, h1) = graphs
    for key in (K.EDGE_INDEX, K.EDGE_CELL_SHIFT, K.TRIPLET_EDGE_INDEX):
        assert torch.equal(g0[key], g1[key]), key                        # GPU builder: same lists on both sides of the face
        assert torch.equal(g0[key].cpu(), h0[key]) and torch.equal(g1[key].cpu(), h1[key]), key   # and equal to the host builder's


def test_verlet_graph_is_the_fresh_build_along_a_trajectory():
    """A random walk of three batched cells (random species; one of them smaller than the cutoff, i.e. with self-images): every
    step compared with a fresh build at the same positions -- index tensors equal, energies / forces / stresses bit-identical --
    and all three paths taken."""
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.data.md import VerletGraph

    K = _K()
    model = _model()
    cells = [random_cell_arrays(40, 8.0, 11), random_cell_arrays(25, 7.0, 12), random_cell_arrays(6, 4.2, 13, dmin=1.8)]
    lats, pos0, zs = zip(*cells)
    sizes = [len(p) for p in pos0]
    vg = VerletGraph(lats, zs, 5.0, 4.0, skin=0.4, device=DEV)
    vg.speculate_after = 0   # evaluate() always queues the step ahead of the verdict here: this test is about that path
    rng = np.random.default_rng(5)
    pos = np.concatenate(pos0)
    for step in range(24):
        # small thermal-like moves; every 8th step one atom jumps by a lattice vector (a wrapped coordinate) -> search path
        # (every 4th step moves nothing to speak of: in a disordered cell some pair crosses a cutoff on any real move, so this
        # is where the reuse path runs)
        pos = pos + rng.normal(0.0, 1e-9 if step % 4 == 2 else 0.02, pos.shape)
        if step % 8 == 7:
            pos[3] += lats[0][1]
        # odd steps: the skin test's verdict is read AFTER the evaluation was queued (evaluate = begin / model / confirm, re-run
        # when the lists had to be rebuilt); even steps wait for it first
        if step % 2:
            out = vg.evaluate(model, torch.tensor(pos, device=DEV), extras=False)
            g = vg.graph
            assert out is g
        else:
            g = vg.update(torch.tensor(pos, device=DEV))
            out = model(g, extras=False)
        e, f, s = out[K.TOTAL_ENERGY].clone(), out[K.FORCES].clone(), out[K.STRESSES].clone()
        split = np.split(pos, np.cumsum(sizes)[:-1])
        fresh = batch_from_arrays(lats, split, zs, 5.0, 4.0, device=DEV)
        _same_graph(g, fresh)
        ref = model(fresh, extras=False)
        assert torch.equal(e, ref[K.TOTAL_ENERGY]) and torch.equal(f, ref[K.FORCES]) and torch.equal(s, ref[K.STRESSES]), step
    assert vg.stats["reuse"] > 0 and vg.stats["refill"] > 1 and vg.stats["search"] >= 3, vg.stats
    # a change of cell (variable-cell relaxation): everything is rebuilt in the new cells, again equal to a fresh build
    lats2 = [L * s_ for L, s_ in zip(lats, (1.02, 0.99, 1.01))]
    pos2 = np.concatenate([p_ * s_ for p_, s_ in zip(np.split(pos, np.cumsum(sizes)[:-1]), (1.02, 0.99, 1.01))])
    vg.set_lattice(lats2)
    g = vg.update(torch.tensor(pos2, device=DEV))
    _same_graph(g, batch_from_arrays(lats2, np.split(pos2, np.cumsum(sizes)[:-1]), zs, 5.0, 4.0, device=DEV))


def test_verlet_graph_reuses_everything_on_the_headline_cell():
    """BASELINE config 3's cell under the bench's MD-style jitter (+-0.025 A around the lattice sites: no shell crosses 5 A or
    4 A): after the first step every update takes the reuse path -- same graph object, same tensors, the engine's cached topology
    and certificate -- and the results are bit-identical to a fresh build's."""
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.data.md import VerletGraph
    from torch_m3gnet.nn.modules import _Topology

    K = _K()
    model = _model()
    a = 3.61
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    gi = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(25), indexing="ij"), -1)
    pos0 = (gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a
    lat = np.diag([10 * a, 10 * a, 25 * a]).astype(float)
    z = np.full(len(pos0), 29)
    vg = VerletGraph([lat], [z], 5.0, 4.0, skin=0.5, device=DEV)
    vg.speculate_after = 0   # (the step queued ahead of the verdict on the last iteration, whatever the history)
    rng = np.random.default_rng(0)
    first, topo = None, None
    for step in range(4):
        pos = pos0 + rng.uniform(-0.025, 0.025, pos0.shape)
        if step == 3:   # the last step without waiting for the verdict first
            out = vg.evaluate(model, torch.tensor(pos, device=DEV), extras=False)
            g = vg.graph
        else:
            g = vg.update(torch.tensor(pos, device=DEV))
            out = model(g, extras=False)
        e, f = out[K.TOTAL_ENERGY].clone(), out[K.FORCES].clone()
        if step == 0:
            first, topo = g, _Topology.of(g)
            assert g[K.NUM_EDGES] == 420_000 and g[K.NUM_TRIPLETS] == 3_060_000
        else:
            assert g is first and _Topology.of(g) is topo      # nothing was rebuilt
        fresh = batch_from_arrays([lat], [pos], [z], 5.0, 4.0, device=DEV)
        _same_graph(g, fresh)
        ref = model(fresh, extras=False)
        assert torch.equal(e, ref[K.TOTAL_ENERGY]) and torch.equal(f, ref[K.FORCES]), step
    assert vg.stats == {"reuse": 3, "refill": 1, "search": 1}, vg.stats


@pytest.mark.parametrize("driver", ["evaluate", "step"])
def test_nve_trajectory_conserves_energy_across_list_updates(driver):
    """(`driver`: VerletGraph.evaluate -- the interpreter sequences skin test, refill, topology, engine -- or VerletGraph.step -- one
    m3g_md_step call per step.)  Physics-level check of the whole trajectory path: velocity-Verlet NVE dynamics of a 108-atom Cu cell on the LJ-FITTED weights
    (trained by the reference's own code, tests/golden/model_fitted_lj.npz), positions resident on the device, lists maintained by
    VerletGraph.evaluate (no wait in front of the step).  Forces that were not the exact gradient of the energy, a stale or
    mis-ordered list after a refill, or a pair missed by the skin list would show as a drift of the total energy that does not
    shrink with the time step; here the error is the integrator's own: it falls by ~4 when dt is halved.

    The REFERENCE MODEL's energy is discontinuous where a pair crosses the two-body cutoff (its radial basis does not vanish
    there: tests/checkers/nve_probe.py, 5.5 meV per pair on these weights -- reproduced faithfully, oracle and engine alike), so the test
    keeps the cutoff in a gap of the fcc shells (4.42 A < 4.76 A < 5.10 A) and the amplitude small: no pair crosses it, while
    pairs cross the three-body cutoff (4.38 A, next to the 4.42 A shell) on most steps -- in the `ref` mode of `factors` the
    three-body term is ~1e-9 of the energy, so those crossings rebuild the triplet lists and the topology without a jump."""
    from helpers import CASE_MODEL, GOLDEN
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data.md import VerletGraph
    from torch_m3gnet.model.build import build_model

    K = _K()
    rc, r3 = 4.76, 4.38
    params, cfg, elemental = orc.load_model_npz(GOLDEN / f"{CASE_MODEL['cu32fit']}.npz")
    model = build_model(rc, r3, cfg.l_max, cfg.n_max, cfg.num_types, cfg.embedding_dim, cfg.num_blocks, elemental_energies=elemental,
                        energy_scale=cfg.energy_scale, length_scale=cfg.length_scale)
    model.load_state_dict({k: v for k, v in params.items()})
    a = 3.61
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    gi = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing="ij"), -1)
    lat = np.eye(3) * 3 * a
    mass, kB, acc_unit = 63.546, 8.617333e-5, 9.64853e-3       # amu, eV/K, (eV/A/amu) -> A/fs^2

    def run(dt, n_steps):
        rng = np.random.default_rng(9)
        pos = torch.tensor((gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a + rng.normal(0, 0.01, (108, 3)), device=DEV)
        vel = torch.tensor(rng.normal(0, np.sqrt(kB * 50.0 / mass * acc_unit), (108, 3)), device=DEV)   # ~50 K, A/fs
        vel -= vel.mean(0, keepdim=True)
        vg = VerletGraph([lat], [np.full(108, 29)], rc, r3, skin=0.3, device=DEV)

        def energy_forces(p):
            out = vg.evaluate(model, p, extras=False) if driver == "evaluate" else vg.step(model, p)
            return out[K.TOTAL_ENERGY].double().sum(), out[K.FORCES].double().clone()

        e_pot, f = energy_forces(pos)
        totals, pots, sizes = [], [], set()
        for _ in range(n_steps):
            vel = vel + 0.5 * dt * acc_unit / mass * f
            pos = pos + dt * vel
            e_pot, f = energy_forces(pos)
            vel = vel + 0.5 * dt * acc_unit / mass * f
            totals.append(float(e_pot + 0.5 * mass / acc_unit * (vel * vel).sum()))
            pots.append(float(e_pot))
            sizes.add((int(vg.graph[K.NUM_EDGES]), int(vg.graph[K.NUM_TRIPLETS])) if driver == "evaluate" else vg._step_sizes)
        totals, pots = np.array(totals), np.array(pots)
        return np.abs(totals - totals[0]).max(), pots.max() - pots.min(), sizes, vg.stats

    drift1, swing, sizes, stats = run(1.0, 300)
    drift2, _, _, _ = run(2.0, 150)
    print(f"NVE 300 fs: potential energy swings {swing:.3f} eV; total energy within {drift1:.1e} eV at dt = 1 fs, {drift2:.1e} eV at 2 fs; paths {stats}")
    assert len({e for e, _ in sizes}) == 1 and len({t for _, t in sizes}) > 50     # no pair crossed the cutoff, triplets changed all the time
    assert stats["refill"] > 100 and stats["search"] >= 2
    assert swing > 0.2 and drift1 < 3e-4 and drift1 < 1e-3 * swing, (drift1, swing)
    assert 2.0 < drift2 / drift1 < 8.0, (drift1, drift2)                            # second-order integrator error, nothing else


def test_verlet_graph_fuzz_random_lattices():
    """30 random batches (tests/checkers/fuzz_graph_build.py's generator: cubic to sheared and left-handed lattices of 2-25 A, 1-120 atoms
    placed up to half a cell outside the home cell, cutoffs 2.5-9 A, 1-4 structures): a four-step random walk each, the skin-list
    graph against a fresh build, index tensors identical."""
    import importlib.util
    from pathlib import Path

    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.data.md import VerletGraph

    spec = importlib.util.spec_from_file_location("fuzz_graph_build", Path(__file__).resolve().parent / "checkers" / "fuzz_graph_build.py")
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(123)
    done = 0
    paths = {"reuse": 0, "refill": 0, "search": 0}
    while done < 30:
        cutoff = float(rng.uniform(2.5, 9.0))
        tb = float(rng.uniform(0.5, 1.0) * cutoff)
        cells = [fz.random_cell(rng) for _ in range(int(rng.integers(1, 5)))]
        if min(abs(np.linalg.det(l)) for l, _ in cells) < 8.0:
            continue
        lats, poss = [l for l, _ in cells], [p for _, p in cells]
        sizes = [len(p) for p in poss]
        zs = [rng.integers(1, 90, n) for n in sizes]
        vg = VerletGraph(lats, zs, cutoff, tb, skin=0.35, device=DEV)
        pos = np.concatenate(poss)
        try:
            for step in range(4):
                pos = pos + rng.normal(0.0, (0.0, 0.25, 0.02, 1e-9)[step], pos.shape)   # the first search, a second one (moves beyond skin / 2), a refill, a reuse
                g = vg.update(torch.tensor(pos, device=DEV))
                if g[_K().NUM_TRIPLETS] > 4_000_000:
                    break
                _same_graph(g, batch_from_arrays(lats, np.split(pos, np.cumsum(sizes)[:-1]), zs, cutoff, tb, device=DEV))
        except AssertionError:
            print(f"case {done}: cutoff {cutoff:.3f} tb {tb:.3f} sizes {sizes}")
            raise
        for k in paths:
            paths[k] += vg.stats[k]
        done += 1
    assert paths["reuse"] >= 15 and paths["refill"] >= 60 and paths["search"] >= 50, paths


def test_two_launch_refill_writes_the_lists_of_the_general_calls():
    """m3g_verlet_fill_lists (edges + triplets + counts of a refill in two launches) against m3g_verlet_fill + m3g_threebody_build on
    the same candidates and positions: every index tensor equal, along a random walk of a batch that includes a cell smaller than
    the cutoff (self-images, rows of ~190 candidates) and a dense one."""
    from torch_m3gnet.data.md import VerletGraph

    cells = [random_cell_arrays(40, 8.0, 21), random_cell_arrays(6, 4.2, 22, dmin=1.8), random_cell_arrays(30, 6.0, 23, dmin=1.4)]
    lats, pos0, zs = zip(*cells)
    a, b = (VerletGraph(lats, zs, 5.0, 4.0, skin=0.4, device=DEV) for _ in range(2))
    b.split_fill = True
    rng = np.random.default_rng(9)
    pos = np.concatenate(pos0)
    for step in range(12):
        pos = pos + rng.normal(0, 0.04, pos.shape)
        p = torch.tensor(pos, device=DEV)
        ga, gb = a.update(p, force="refill"), b.update(p, force="refill")
        _same_graph(ga, gb)
    assert a.stats["refill"] >= 12 and b.stats["refill"] >= 12


def test_canonical_topology_build_equals_the_checked_build():
    """A graph marked as written by the library's own builders takes m3g_topology_build_canonical (no mirror / completeness checks);
    the same tensors without the mark take the checked build: same certificate word, bit-identical results.  An in-place change of
    an index tensor voids the mark (the checked build then raises for the broken list)."""
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.nn.modules import _Topology

    K = _K()
    model = _model()
    lat, pos, z = random_cell_arrays(48, 8.5, 31)
    g = batch_from_arrays([lat], [pos], [z], 5.0, 4.0, device=DEV)
    assert g.get("_m3g_canonical_lists") == _Topology.signature(g)
    h = g.clone() if hasattr(g, "clone") else g
    plain = type(g).__new__(type(g))
    dict.__init__(plain)
    for k, v in g.items():
        if not str(k).startswith("_m3g_"):
            plain[k] = v
    t_can, t_chk = _Topology(g), _Topology(plain)
    assert t_can.query_hints() == t_chk.query_hints() and (t_can.query_hints() & 1)
    out_c, out_p = model(g), model(plain)
    for key in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES):
        assert torch.equal(out_c[key], out_p[key]), key
    # an in-place edit voids the mark
    g[K.TRIPLET_EDGE_INDEX][1, 0] = g[K.TRIPLET_EDGE_INDEX][1, 1]
    assert g.get("_m3g_canonical_lists") != _Topology.signature(g)


def test_queued_topology_build_gives_the_same_results_and_survives_unevaluated_graphs():
    """The trajectory graph queues its topology build behind the fill (m3g_topology_build_canonical_begin) and the engine waits for
    it when it needs the buffer.  Same energies / forces, bit for bit, as with the build at the engine's call; and graphs that
    are refilled again before anyone evaluated them (their builds share one pinned verdict block) do not disturb later ones."""
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.md import VerletGraph
    from torch_m3gnet.nn.modules import _Topology

    model = _model()
    a = 3.61
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    gi = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing="ij"), -1)
    pos = torch.tensor((gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a, device="cuda")
    lat, z = np.eye(3) * 3 * a, np.full(108, 29)
    outs = {}
    for eager in (True, False):
        vg = VerletGraph([lat], [z], 5.0, 4.0, skin=0.4, device="cuda")
        vg.eager_topology = eager
        gen = torch.Generator(device="cuda").manual_seed(3)
        res = []
        for it in range(6):
            p = pos + (torch.rand(pos.shape, generator=gen, device="cuda", dtype=torch.float64) - 0.5) * 0.08
            g = vg.update(p, force="refill")
            topo = g["_m3g_topology"][1] if eager else None
            assert (topo is not None and topo._pending is not None) == eager
            if it % 2 == 1:
                continue   # never evaluated: the next refill finishes its build before queueing its own
            out = model(g, forces=True, extras=False)
            res.append((out[K.TOTAL_ENERGY].clone(), out[K.FORCES].clone()))
            assert _Topology.of(g)._pending is None
        outs[eager] = res
    for (e1, f1), (e0, f0) in zip(outs[True], outs[False]):
        assert torch.equal(e1, e0) and torch.equal(f1, f0)


def test_evaluate_guesses_only_after_a_streak_of_unchanged_lists():
    """VerletGraph.evaluate queues the step ahead of the skin test's verdict only after `speculate_after` consecutive "unchanged"
    verdicts: a trajectory whose lists change on every step pays for exactly one evaluation per step (a wrong guess would cost
    two), a frozen one runs without the wait from the fifth step on.  Same results either way (bit-identical to update + model)."""
    from torch_m3gnet.data.md import VerletGraph

    K = _K()
    model = _model()
    calls = [0]

    def counted(g, **kw):
        calls[0] += 1
        return model(g, **kw)

    a = 3.61
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    gi = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing="ij"), -1)
    pos0 = torch.tensor((gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a, device=DEV)
    lat, z = np.eye(3) * 3 * a, np.full(108, 29)
    gen = torch.Generator(device=DEV).manual_seed(1)
    # hot: +-0.15 A jitter of a perfect crystal whose 5.1 A shell sits 0.1 A outside the cutoff -- pairs cross it on every step
    vg = VerletGraph([lat], [z], 5.0, 4.0, skin=1.0, device=DEV)
    ref = VerletGraph([lat], [z], 5.0, 4.0, skin=1.0, device=DEV)
    for step in range(10):
        p = pos0 + (torch.rand(pos0.shape, generator=gen, device=DEV, dtype=torch.float64) - 0.5) * 0.3
        out = vg.evaluate(counted, p, extras=False)
        e, f = out[K.TOTAL_ENERGY].clone(), out[K.FORCES].clone()
        want = model(ref.update(p), extras=False)
        assert torch.equal(e, want[K.TOTAL_ENERGY]) and torch.equal(f, want[K.FORCES])
    assert vg.stats["reuse"] == 0 and calls[0] == 10, (vg.stats, calls)
    # frozen: moves of 1e-9 A change nothing; the guess starts after four confirmations and is right every time
    calls[0] = 0
    for step in range(10):
        p = pos0 + (torch.rand(pos0.shape, generator=gen, device=DEV, dtype=torch.float64) - 0.5) * 1e-9
        vg.evaluate(counted, p, extras=False)
        assert (vg._reuse_streak >= vg.speculate_after) == (step >= 4), (step, vg._reuse_streak)
    assert calls[0] == 10
    # a change after a long streak: one wasted evaluation, then back to waiting first
    p = pos0 + (torch.rand(pos0.shape, generator=gen, device=DEV, dtype=torch.float64) - 0.5) * 0.3
    calls[0] = 0
    out = vg.evaluate(counted, p, extras=False)
    want = model(ref.update(p), extras=False)
    assert calls[0] == 2 and vg._reuse_streak == 0
    assert torch.equal(out[K.TOTAL_ENERGY], want[K.TOTAL_ENERGY]) and torch.equal(out[K.FORCES], want[K.FORCES])


def test_step_through_the_c_side_trajectory_object_is_the_update_path_bit_for_bit():
    """VerletGraph.step (m3g_md_step: skin test, refill, topology and engine sequenced by the library on capacity buffers) against
    model(update(pos)) of a second VerletGraph along the same random walk of three batched cells: energies, forces, stresses and the
    lists bit-identical on every step, all three paths taken, a lattice-vector jump (search) included; then the two ways mixed on
    ONE object."""
    from torch_m3gnet.data.md import VerletGraph

    K = _K()
    model = _model()
    lats, pos0, zs = [], [], []
    for s, (n, box) in enumerate(((20, 6.5), (9, 4.4), (31, 7.3))):   # the second cell is smaller than the cutoff: self-images
        lat, p, z = random_cell_arrays(n, box, seed=50 + s)
        lats.append(lat); pos0.append(p); zs.append(z)
    a, b = (VerletGraph(lats, zs, 5.0, 4.0, skin=0.4, device=DEV) for _ in range(2))
    rng = np.random.default_rng(8)
    pos = np.concatenate(pos0)
    for step in range(20):
        pos = pos + rng.normal(0.0, 1e-9 if step % 4 == 2 else 0.02, pos.shape)
        if step % 8 == 7:
            pos[3] += lats[0][1]
        p = torch.tensor(pos, device=DEV)
        got = a.step(model, p)
        g = b.update(p)
        want = model(g, extras=False)
        for key in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES):
            assert torch.equal(got[key], want[key]), (step, key)
        lists = a.step_lists()
        for key in (K.EDGE_INDEX, K.EDGE_CELL_SHIFT, K.TRIPLET_EDGE_INDEX, K.NUM_TRIPLET_I, K.NUM_TRIPLET_IJ):
            assert torch.equal(lists[key], g[key]), (step, key)
    assert a.stats == b.stats and all(v > 0 for v in a.stats.values()), (a.stats, b.stats)
    # energies only
    p = torch.tensor(pos + rng.normal(0.0, 0.02, pos.shape), device=DEV)
    e_only = a.step(model, p, forces=False)
    assert K.FORCES not in e_only and torch.equal(e_only[K.TOTAL_ENERGY], model(b.update(p), extras=False)[K.TOTAL_ENERGY])
    # the two ways on ONE object, alternating: each re-derives the lists the other wrote last
    for step in range(8):
        pos = pos + rng.normal(0.0, 0.02, pos.shape)
        p = torch.tensor(pos, device=DEV)
        want = model(b.update(p), extras=False)
        got = a.step(model, p) if step % 2 else model(a.update(p), extras=False)
        for key in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES):
            assert torch.equal(got[key], want[key]), (step, key)
    # buffers kept from an earlier search that turn out too small for the triplets of a new one: the library refuses the refill
    # (nothing evaluated), `step` makes buffers of the exact capacity and calls again -- same results
    a._md_buffers["cap_t"] = 8
    a._md_buffers["tei"] = torch.empty(16, dtype=torch.int64, device=DEV)
    pos = pos + rng.normal(0.0, 0.02, pos.shape)
    p = torch.tensor(pos, device=DEV)
    got = a.step(model, p, force="search")
    want = model(b.update(p, force="search"), extras=False)
    assert a._md_buffers["cap_t"] > 8
    for key in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES):
        assert torch.equal(got[key], want[key]), key
    # a new cell (variable-cell relaxation): both objects search again in it
    scale = 1.02
    lats2 = [l * scale for l in lats]
    a.set_lattice(lats2)
    b.set_lattice(lats2)
    p = torch.tensor(pos * scale, device=DEV)
    got, want = a.step(model, p), model(b.update(p), extras=False)
    for key in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES):
        assert torch.equal(got[key], want[key]), key
    # a species outside the model's table fails as in the reference
    bad = VerletGraph([lats[0]], [np.full(3, 120)], 5.0, 4.0, skin=0.4, device=DEV)
    with pytest.raises(IndexError):
        bad.step(model, torch.tensor(pos0[0][:3], device=DEV))


@pytest.mark.parametrize("name,pos,z", [("isolated atom", [[1.0, 2.0, 3.0]], [29]),
                                        ("pair beyond the three-body cutoff", [[1.0, 2.0, 3.0], [5.5, 2.0, 3.0]], [29, 8])])
def test_step_on_cells_without_edges_or_without_triplets(name, pos, z):
    """m3g_md_step on a cell with no edge at all, and on one with edges but no triplet (capacity 0 for the triplet buffers): the same
    energies / forces / stresses as update + model, through the refill and the reuse path."""
    from torch_m3gnet.data.md import VerletGraph

    K = _K()
    model = _model()
    lat, pos, z = np.eye(3) * 20.0, np.asarray(pos), np.asarray(z)
    a, b = (VerletGraph([lat], [z], 5.0, 4.0, skin=0.4, device=DEV) for _ in range(2))
    for it in range(3):
        p = torch.tensor(pos + 0.01 * it, device=DEV)
        got = a.step(model, p)
        want = model(b.update(p), extras=False)
        for key in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES):
            assert torch.equal(got[key], want[key]), (name, it, key)
    assert a.stats == b.stats and a.stats["reuse"] == 2


def test_evaluate_after_step_after_speculative_evaluates_on_one_object():
    """evaluate() past `speculate_after` (so the next evaluate takes begin()'s queued-ahead path), then step() (which takes the
    lists over to the C side and drops `self.graph`), then evaluate() again on the SAME object: the speculative path must fall back
    to update() instead of writing into a graph that is gone (advisor finding, round 5).  Results equal a second object driven
    through update() alone, bit for bit."""
    from torch_m3gnet.data.md import VerletGraph

    K = _K()
    model = _model()
    lat, p0, z = random_cell_arrays(24, 7.0, seed=77)
    a, b = (VerletGraph([lat], [z], 5.0, 4.0, skin=0.6, device=DEV) for _ in range(2))
    pos = p0.copy()
    rng = np.random.default_rng(5)

    def both(kind):
        p = torch.tensor(pos, device=DEV)
        want = model(b.update(p), extras=False)
        got = a.step(model, p) if kind == "step" else a.evaluate(model, p, extras=False)
        for key in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES):
            assert torch.equal(got[key], want[key]), (kind, key)

    for _ in range(a.speculate_after + 2):        # frozen cell: every verdict is "unchanged"
        pos = pos + rng.normal(0.0, 1e-10, pos.shape)
        both("evaluate")
    assert a._reuse_streak >= a.speculate_after
    both("step")
    pos = pos + rng.normal(0.0, 1e-10, pos.shape)
    both("evaluate")                              # crashed with TypeError before the fix
    assert a._reuse_streak < a.speculate_after    # the streak counts verdicts about the current lists only
    for _ in range(a.speculate_after + 2):
        pos = pos + rng.normal(0.0, 1e-10, pos.shape)
        both("evaluate")
    both("step")
    pos = pos + rng.normal(0.0, 0.05, pos.shape)  # and one with the lists changed
    both("evaluate")


def test_trajectory_step_reports_sticky_error_bits_of_the_previous_step():
    """m3g_md_step reads the sticky error word of its topology buffer behind the wait for the skin test's verdict (the word rides to
    pinned host memory with the fp32 copy of the positions): a step whose in-launch wait ran out (forced: option
    debug_node_tb_polls < 0 -> forces NaN + M3G_TOPO_ERR_SYNC) makes the NEXT step fail with M3G_ERR_STATE instead of handing out more
    numbers from that buffer; the step after that re-derives lists and topology (which clears the word) and is clean again."""
    from torch_m3gnet.data.md import VerletGraph

    K = _K()
    model = _model()
    lat, p0, z = random_cell_arrays(40, 8.0, seed=11)   # <= 128 atoms: the node + three-body reverse of a block share a launch
    vg = VerletGraph([lat], [z], 5.0, 4.0, skin=0.6, device=DEV)
    ref = VerletGraph([lat], [z], 5.0, 4.0, skin=0.6, device=DEV)
    pos = torch.tensor(p0, device=DEV)
    good = vg.step(model, pos)
    assert torch.isfinite(good[K.FORCES]).all()
    model.engine.set_option("debug_node_tb_polls", -64)
    try:
        bad = vg.step(model, pos)                      # standing lists: the same topology buffer
        assert torch.isnan(bad[K.FORCES]).any()        # loud already: the term that was waited for is NaN
    finally:
        model.engine.set_option("debug_node_tb_polls", 0)
    with pytest.raises(RuntimeError, match="error bits"):
        vg.step(model, pos)
    again = vg.step(model, pos)                        # lists and topology re-derived: clean
    want = model(ref.update(pos), extras=False)
    for key in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES):
        assert torch.equal(again[key], want[key]), key
