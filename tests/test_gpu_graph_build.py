"""GPU graph construction against the host builder, element by element.

Both builders emit the canonical order (centre, cell shift relative to the given coordinates lexicographic, neighbour), so index tensors must
be IDENTICAL (integer work: bit-exact); distances are fp64 on both sides and compared to 1e-12 relative.
Counts pinned by the reference's own fixtures: Al/Na 132 edges at r_c = 5... are in tests/golden (case_alna)."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, build_engine_model, load_oracle_case

pytestmark = pytest.mark.gpu


def _host(lat, pos, cutoff, tb):
    from torch_m3gnet.data.neighbors import neighbor_list, threebody_index

    ei, sh, d = neighbor_list(lat, pos, cutoff)
    tei, nti, ntij = threebody_index(len(pos), ei, d.astype(np.float32), tb)
    return ei, sh, d, tei, nti, ntij


def _gpu(lats, poss, cutoff, tb):
    from torch_m3gnet.data.graph_gpu import neighbor_list_gpu, threebody_index_gpu

    lat = torch.tensor(np.stack(lats), dtype=torch.float64, device="cuda")
    pos = torch.tensor(np.concatenate(poss), dtype=torch.float64, device="cuda")
    batch = torch.tensor(np.repeat(np.arange(len(poss)), [len(p) for p in poss]), device="cuda")
    ei, sh, d = neighbor_list_gpu(lat, pos, batch, cutoff)
    tei, nti, ntij = threebody_index_gpu(pos.size(0), ei, d, tb)
    # the one-wait path (counts edges AND triplets in the neighbour count pass) must produce the very same tensors
    from torch_m3gnet.data.graph_gpu import graph_indices_gpu

    for a, b in zip(graph_indices_gpu(lat, pos, batch, cutoff, tb), (ei, sh, d, tei, nti, ntij)):
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)
    return [t.cpu().numpy() for t in (ei, sh, d, tei, nti, ntij)]


def _assert_same(host, gpu):
    names = ["edge_index", "edge_cell_shift", "distances", "triplet_edge_index", "num_triplet_i", "num_triplet_ij"]
    for n, h, g in zip(names, host, gpu):
        assert h.shape == g.shape, (n, h.shape, g.shape)
        if n == "distances":
            np.testing.assert_allclose(g, h, rtol=1e-12, atol=1e-12)
        else:
            assert h.dtype == g.dtype, (n, h.dtype, g.dtype)
            np.testing.assert_array_equal(g, h, err_msg=n)


def _cells():
    rng = np.random.default_rng(5)
    out = {}
    a = 3.61
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    out["cu_primitive_like"] = (np.eye(3) * a, base * a)                                  # many self images
    grid = np.stack(np.meshgrid(*[np.arange(2)] * 3, indexing="ij"), -1).reshape(-1, 1, 3)
    out["cu_2x2x2"] = (np.eye(3) * 2 * a, ((grid + base[None]).reshape(-1, 3) * a) + rng.uniform(-.02, .02, (32, 3)))
    tri = np.array([[4.1, 0.3, -0.2], [1.7, 5.2, 0.4], [-0.9, 1.1, 6.3]])
    out["triclinic_unwrapped"] = (tri, rng.uniform(-1.5, 2.5, (11, 3)) @ tri)                # atoms outside the home cell
    out["single_atom_small_cell"] = (np.diag([2.1, 2.6, 3.3]), np.zeros((1, 3)))           # only self images
    out["skewed_left_handed"] = (np.array([[0, 5.0, 0.5], [6.0, 0, 0], [2.5, 2.5, 7.0]]), rng.uniform(0, 1, (17, 3)) @
                                 np.array([[0, 5.0, 0.5], [6.0, 0, 0], [2.5, 2.5, 7.0]]))    # det < 0
    out["two_bins_per_axis"] = (np.eye(3) * 10.7, rng.uniform(0, 10.7, (60, 3)))            # linked cells with nb = 2: +-1 reach the same bin
    out["three_bins_sheared"] = (np.array([[16.0, 0, 0], [4.0, 15.5, 0], [-3.0, 2.0, 17.0]]), rng.uniform(0, 1, (150, 3)) @
                                 np.array([[16.0, 0, 0], [4.0, 15.5, 0], [-3.0, 2.0, 17.0]]))
    slab = rng.uniform(0, 1, (40, 3)) * np.array([8.0, 8.0, 6.0])
    out["slab_in_vacuum"] = (np.diag([8.0, 8.0, 400.0]), slab)                                  # bins coarsened over empty space
    return out


@pytest.mark.parametrize("name", list(_cells()))
@pytest.mark.parametrize("cutoff,tb", [(5.0, 4.0), (3.3, 3.3)])
def test_single_structure_matches_host_builder(name, cutoff, tb):
    lat, pos = _cells()[name]
    _assert_same(_host(lat, pos, cutoff, tb), _gpu([lat], [pos], cutoff, tb))


def test_reference_fixture_counts():
    """The reference's own test structures (tests/conftest.py:89-115: fcc Al + bcc Na, nearest-neighbour distance 3.0,
    cutoff 3.0001).  The golden case_alna holds the graph the reference's compute_threebody produced for the ideal
    lattices (48 + 16 edges, 640 triplets); the GPU builder must reproduce that edge set and the triplet counts."""
    _, cfg, _, graph, _ = load_oracle_case("alna", "ref")
    r = 3.0
    lat_al, lat_na = r * np.sqrt(2) * np.eye(3), r / np.sqrt(3) * 2 * np.eye(3)
    lats = [lat_al, lat_na]
    poss = [np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]]) @ lat_al, np.array([[0, 0, 0], [0.5, 0.5, 0.5]]) @ lat_na]
    ei, sh, d, tei, nti, ntij = _gpu(lats, poss, cfg.cutoff, cfg.threebody_cutoff)
    assert ei.shape[1] == graph["edge_index"].shape[1] == 64
    assert tei.shape[1] == graph["triplet_edge_index"].shape[1] == 640
    np.testing.assert_allclose(d, 3.0, atol=1e-9)

    # the fixture carries the canonical order (centre, image lexicographic, neighbour): element by element
    np.testing.assert_array_equal(ei, graph["edge_index"].numpy())
    np.testing.assert_array_equal(sh, graph["edge_cell_shift"].numpy())
    np.testing.assert_array_equal(nti, np.bincount(graph["edge_index"][0].numpy()[graph["triplet_edge_index"][0].numpy()], minlength=6))


def test_batched_structures_match_concatenated_host_graphs():
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph

    cells = _cells()
    names = ["cu_2x2x2", "triclinic_unwrapped", "single_atom_small_cell", "skewed_left_handed"]
    rng = np.random.default_rng(0)
    zs = [rng.integers(1, 95, len(cells[n][1])) for n in names]
    host = Batch.from_data_list([MaterialGraph.from_arrays(cells[n][0], cells[n][1], z, 5.0, 4.0) for n, z in zip(names, zs)])
    dev = batch_from_arrays([cells[n][0] for n in names], [cells[n][1] for n in names], zs, 5.0, 4.0)
    for key in ("edge_index", "edge_cell_shift", "triplet_edge_index", "num_triplet_i", "num_triplet_ij", "batch", "atom_types"):
        assert torch.equal(dev[key].cpu(), host[key]), key
    torch.testing.assert_close(dev["pos"].cpu(), host["pos"])
    torch.testing.assert_close(dev["lattice"].cpu(), host["lattice"])
    assert dev["num_nodes"] == host["num_nodes"] and dev["num_edges"] == host["num_edges"] and dev["num_triplets"] == host["num_triplets"]


def test_gpu_built_graph_gives_the_same_energy_and_forces():
    """End to end: structure arrays -> GPU graph -> engine equals host graph -> engine bit for bit."""
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph

    model, cfg = build_engine_model("cu32", "ref")
    model = model.cuda()
    lat, pos = _cells()["cu_2x2x2"]
    z = np.full(len(pos), 29)
    host = Batch.from_data_list([MaterialGraph.from_arrays(lat, pos, z, cfg.cutoff, cfg.threebody_cutoff)]).to("cuda")
    dev = batch_from_arrays([lat], [pos], [z], cfg.cutoff, cfg.threebody_cutoff)
    a, b = model(host), model(dev)
    assert torch.equal(a["total_energy"], b["total_energy"])
    assert torch.equal(a["forces"], b["forces"])


def test_large_supercell_counts_and_properties():
    """BASELINE config 3 (10,000 Cu atoms): 420,000 edges / 3,060,000 triplets, built on the GPU."""
    from torch_m3gnet.data.graph_gpu import batch_from_arrays

    a = 3.61
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    g = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(25), indexing="ij"), -1)
    pos = (g.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a
    pos = pos + np.random.default_rng(0).uniform(-0.025, 0.025, pos.shape)
    lat = np.diag([10 * a, 10 * a, 25 * a])
    gr = batch_from_arrays([lat], [pos], [np.full(len(pos), 29)], 5.0, 4.0)
    assert gr["num_edges"] == 420_000 and gr["num_triplets"] == 3_060_000
    ei, sh = gr["edge_index"], gr["edge_cell_shift"]
    assert bool((ei[0, 1:] >= ei[0, :-1]).all())                         # centre sorted
    # every edge has its reverse: (j, i, -shift)
    fwd = torch.cat([ei.t(), sh.long()], 1)
    rev = torch.cat([ei.flip(0).t(), -sh.long()], 1)
    def key(t):
        t = t + torch.tensor([0, 0, 8, 8, 8], device=t.device)
        return (((t[:, 0] * 10_000 + t[:, 1]) * 17 + t[:, 2]) * 17 + t[:, 3]) * 17 + t[:, 4]
    assert torch.equal(key(fwd).sort().values, key(rev).sort().values)
    # distances recomputed from the emitted shifts are inside the cutoff
    p = torch.tensor(pos, device="cuda")
    r = p[ei[1]] + sh.double() @ torch.tensor(lat, device="cuda") - p[ei[0]]
    d = r.norm(dim=1)
    assert float(d.max()) <= 5.0 + 1e-8 and float(d.min()) > 1.0
    tei = gr["triplet_edge_index"]
    assert bool((ei[0][tei[0]] == ei[0][tei[1]]).all()) and bool((tei[0] != tei[1]).all())
    assert int(gr["num_triplet_ij"].sum()) == gr["num_triplets"] == int(gr["num_triplet_i"].sum())


def test_long_segments_take_the_fallback_paths():
    """Both wave-per-atom kernels stage one segment in LDS and fall back to a single lane when it does not fit: more than 512
    neighbours of an atom inside ONE periodic image (dense cell, 12 A cutoff, two bins per axis), and more than 256 edges of a
    centre inside the three-body cutoff.  Same element-by-element comparison with the host builder."""
    a = 3.61
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    rng = np.random.default_rng(9)
    grid = np.stack(np.meshgrid(*[np.arange(7)] * 3, indexing="ij"), -1).reshape(-1, 1, 3)
    pos = (grid + base[None]).reshape(-1, 3) * a + rng.uniform(-0.03, 0.03, (4 * 343, 3))
    lat = np.eye(3) * 7 * a
    host = _host(lat, pos, 12.0, 3.0)
    per_image = np.unique(np.concatenate([host[0][:1].T, host[1]], axis=1), axis=0, return_counts=True)[1]
    assert per_image.max() > 512, per_image.max()     # the neighbour kernel's oversized-segment path is exercised
    _assert_same(host, _gpu([lat], [pos], 12.0, 3.0))
    grid = np.stack(np.meshgrid(*[np.arange(3)] * 3, indexing="ij"), -1).reshape(-1, 1, 3)
    pos = (grid + base[None]).reshape(-1, 3) * a + rng.uniform(-0.03, 0.03, (108, 3))
    lat = np.eye(3) * 3 * a
    host = _host(lat, pos, 10.0, 10.0)
    assert host[4].max() > 256 * 255, host[4].max()   # a centre with more than 256 valid edges: the triplet kernel's long-row path
    _assert_same(host, _gpu([lat], [pos], 10.0, 10.0))


def test_random_cells_match_host_builder():
    """60 random batches (tests/checkers/fuzz_graph_build.py: cubic to sheared and left-handed lattices of 2-25 A, 1-120 atoms, cutoffs
    2.5-9 A, 1-6 structures per call) -- the sub-wave image groups of small cells, multi-bin cells and mixed batches."""
    import importlib.util
    import sys
    from pathlib import Path

    path = Path(__file__).resolve().parent / "checkers" / "fuzz_graph_build.py"
    spec = importlib.util.spec_from_file_location("fuzz_graph_build", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    sys.argv = [str(path), "60", "7"]
    try:
        mod.main()
    finally:
        sys.argv = argv


def test_graphs_without_edges_or_triplets_run_end_to_end():
    """Isolated atoms (no edge at all), a dimer (edges, no triplet) and a batch mixing an isolated atom with a dense cell, built on
    the GPU and evaluated: finite energies, zero force on isolated atoms, Newton's third law on the dimer, triplet counts."""
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.model.build import build_model

    torch.manual_seed(0)
    model = build_model(5.0, 4.0, 3, 3, 95, 64, 3)
    big = np.eye(3) * 30.0
    g = batch_from_arrays([big], [np.array([[1.0, 1.0, 1.0], [15.0, 15.0, 15.0]])], [np.array([29, 8])], 5.0, 4.0)
    assert g[K.NUM_EDGES] == 0 and g[K.NUM_TRIPLETS] == 0 and tuple(g[K.TRIPLET_EDGE_INDEX].shape) == (2, 0)
    out = model(g)
    assert torch.isfinite(out[K.TOTAL_ENERGY]).all() and float(out[K.FORCES].abs().max()) == 0.0
    g = batch_from_arrays([big], [np.array([[1.0, 1.0, 1.0], [3.0, 1.0, 1.0]])], [np.array([29, 29])], 5.0, 4.0)
    assert g[K.NUM_EDGES] == 2 and g[K.NUM_TRIPLETS] == 0
    out = model(g)
    f = out[K.FORCES]
    assert torch.isfinite(out[K.TOTAL_ENERGY]).all() and float(f.abs().max()) > 0 and float(f.sum(0).abs().max()) < 1e-6 * float(f.abs().max())
    a = 3.61
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]]) * a
    g = batch_from_arrays([big, np.eye(3) * a], [np.array([[1.0, 1.0, 1.0]]), base], [np.array([3]), np.full(4, 29)], 5.0, 4.0)
    assert g[K.NUM_EDGES] == 4 * 42 and g[K.NUM_TRIPLET_I].tolist() == [0] + [18 * 17] * 4
    out = model(g)
    assert torch.isfinite(out[K.TOTAL_ENERGY]).all() and float(out[K.FORCES][0].abs().max()) == 0.0


def test_long_rows_with_atoms_outside_the_home_cell_are_ordered_canonically():
    """The canonical order sorts a centre's edges by the shift relative to the GIVEN coordinates.  The search emits them by image
    of the wrapped cell, so rows that hold a neighbour outside the home cell are re-ranked (k_rows_canonical): from an LDS stage
    for rows of up to 512 edges, by a block-wise odd-even merge through the same stage beyond -- a 7-atom cell of 3.3 A under a 9 A cutoff has ~590 edges per
    centre, and half of the atoms are moved out of the home cell by lattice vectors.  Against the host builder, element by
    element."""
    rng = np.random.default_rng(17)
    lat = np.array([[3.3, 0.2, 0.0], [-0.3, 3.4, 0.1], [0.2, -0.1, 3.2]])
    frac = rng.uniform(0, 1, (7, 3))
    frac[::2] += rng.integers(-2, 3, (4, 3))           # every other atom outside the home cell
    pos = frac @ lat
    host = _host(lat, pos, 9.0, 3.0)
    deg = np.bincount(host[0][0], minlength=7)
    assert deg.max() > 512 and (np.abs(np.floor(frac)).sum() > 0)
    _assert_same(host, _gpu([lat], [pos], 9.0, 3.0))
    # rows of ~1,700 edges: seven blocks of 256 through the block-wise odd-even merge of k_rows_canonical
    host = _host(lat, pos, 12.8, 3.0)
    assert np.bincount(host[0][0], minlength=7).max() > 1536
    _assert_same(host, _gpu([lat], [pos], 12.8, 3.0))
    # and a moderate case through the LDS stage (rows of ~150 edges)
    host = _host(lat, pos, 5.5, 4.0)
    assert 64 < np.bincount(host[0][0], minlength=7).max() <= 512
    _assert_same(host, _gpu([lat], [pos], 5.5, 4.0))


def test_batch_from_structures_and_pinned_staging():
    """The two host-side ways into the engine agree with each other: structure objects -> GPU builder (`batch_from_structures`),
    and host-built graphs staged through pinned memory with asynchronous copies (`Batch.pin_memory().to(device,
    non_blocking=True)`)."""
    from types import SimpleNamespace

    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.graph_gpu import batch_from_structures
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph

    cells = _cells()
    names = ["cu_2x2x2", "triclinic_unwrapped"]
    structs = [SimpleNamespace(lattice=SimpleNamespace(matrix=cells[n][0]), cart_coords=cells[n][1],
                               atomic_numbers=np.full(len(cells[n][1]), 29 if n.startswith("cu") else 8)) for n in names]
    on_gpu = batch_from_structures(structs, 5.0, 4.0)
    host = Batch.from_data_list([MaterialGraph.from_structure(s, 5.0, 4.0) for s in structs])
    pinned = host.pin_memory()
    assert all(t.is_pinned() for t in pinned.values() if torch.is_tensor(t))
    staged = pinned.to("cuda", non_blocking=True)
    torch.cuda.synchronize()
    for key in (K.POS, K.ATOM_TYPES, K.EDGE_INDEX, K.EDGE_CELL_SHIFT, K.TRIPLET_EDGE_INDEX, K.LATTICE, K.BATCH):
        assert torch.equal(staged[key], on_gpu[key]), key


# ---- the six-launch topology build for canonical lists ---------------------------------------------------------------------------------
def _topology_buffers(g):
    """The same graph through the general build (m3g_topology_build_hints) and the canonical one, into zero-filled buffers:
    (data bytes of both, hints of both, path the canonical call took)."""
    import ctypes as C

    from torch_m3gnet import _lib
    from torch_m3gnet.data import MaterialGraphKey as K

    lib = _lib.load_library()
    ei, tei, batch = g[K.EDGE_INDEX].contiguous().long(), g[K.TRIPLET_EDGE_INDEX].contiguous().long(), g[K.BATCH].contiguous().long()
    N, E, T, S = int(batch.numel()), int(ei.size(1)), int(tei.size(1)), int(g[K.LATTICE].size(0))
    nbytes, dbytes = C.c_size_t(), C.c_size_t()
    _lib.check(lib.m3g_topology_bytes(N, E, T, S, C.byref(nbytes)))
    _lib.check(lib.m3g_topology_data_bytes(N, E, T, S, C.byref(dbytes)))
    assert 0 < dbytes.value <= nbytes.value
    out = []
    for build in (lib.m3g_topology_build_hints, lib.m3g_topology_build_canonical):
        buf = torch.zeros(nbytes.value, dtype=torch.uint8, device=ei.device)
        flags, hints = (C.c_int32 * 1)(0), C.c_int32(0)
        _lib.check(build(N, E, T, S, C.c_void_p(ei.data_ptr()), C.c_void_p(tei.data_ptr()), C.c_void_p(batch.data_ptr()),
                         C.c_void_p(buf.data_ptr()), nbytes.value, flags, C.byref(hints), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        out.append((buf[: dbytes.value].cpu(), int(hints.value), int(flags[0])))
    path = C.c_int32(-1)
    _lib.check(lib.m3g_topology_debug_last_path(C.byref(path)))
    return out[0], out[1], int(path.value)


def _canonical_graphs():
    from helpers import random_cell_arrays
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.data.synthetic import fcc_cu_graph

    yield "cu 2x2x2", fcc_cu_graph(2, 2, 2).to("cuda")
    yield "cu 6x6x6", fcc_cu_graph(6, 6, 6).to("cuda")
    yield "cu 10x10x25", fcc_cu_graph(10, 10, 25).to("cuda")
    yield "one atom", batch_from_arrays(*zip(random_cell_arrays(1, 3.0, 5)), 5.0, 4.0, device="cuda")
    cells = [random_cell_arrays(8 + 7 * s, 6.0 + 0.5 * s, 40 + s) for s in range(9)]
    yield "9 random cells", batch_from_arrays(*zip(*cells), 5.0, 4.0, device="cuda")
    # three-body cutoff == cutoff at ~60 neighbours: windows of more than kTbCap rows / the certificate says "not every window"
    yield "dense, r3 = rc = 6", batch_from_arrays(*zip(random_cell_arrays(300, 16.5, 3)), 6.0, 6.0, device="cuda")
    # rows of ~150 edges (several 64-lane passes per row), atoms outside the home cell
    rng = np.random.default_rng(17)
    lat = np.array([[3.3, 0.2, 0.0], [-0.3, 3.4, 0.1], [0.2, -0.1, 3.2]])
    frac = rng.uniform(0, 1, (7, 3))
    frac[::2] += rng.integers(-2, 3, (4, 3))
    yield "7-atom cell, rows of ~150 edges", batch_from_arrays([lat], [frac @ lat], [np.full(7, 28)], 5.5, 4.0, device="cuda")
    # atoms without any active edge, and structures without triplets beside structures with them
    sparse = [random_cell_arrays(3, 9.0, 70 + s) for s in range(4)] + [random_cell_arrays(30, 7.5, 80)]
    yield "sparse + dense structures", batch_from_arrays(*zip(*sparse), 5.0, 3.0, device="cuda")


def test_canonical_topology_build_writes_the_general_builds_buffer():
    """m3g_topology_build_canonical (six launches; csrc/m3g_topology.hip, k_canon_*) against m3g_topology_build_hints on the lists
    the library's builders write: every array of the topology buffer bit for bit, the certificate's word, and the fast path taken."""
    from torch_m3gnet.data import MaterialGraphKey as K

    for name, g in _canonical_graphs():
        if int(g[K.NUM_TRIPLETS]) == 0:
            continue
        (a, ha, fa), (b, hb, fb), path = _topology_buffers(g)
        assert path == 1, name
        assert fa == 0 and fb == 0, name
        assert ha == hb, (name, hex(ha), hex(hb))
        if not torch.equal(a, b):
            bad = torch.nonzero(a != b).flatten()
            raise AssertionError(f"{name}: {bad.numel()} bytes differ, first at {int(bad[0])} of {a.numel()}")


def test_canonical_topology_build_falls_back_on_lists_that_are_not_canonical():
    """A shuffled triplet list / an unsorted edge list handed to the canonical entry point: its checks fail, the general build runs
    (path 0) and returns what it returns for that list."""
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.synthetic import fcc_cu_graph

    g = fcc_cu_graph(3, 3, 3).to("cuda")
    tei = g[K.TRIPLET_EDGE_INDEX]
    perm = torch.randperm(tei.size(1), device=tei.device, generator=torch.Generator(device=tei.device).manual_seed(0))
    g[K.TRIPLET_EDGE_INDEX] = tei[:, perm].contiguous()
    (a, ha, fa), (b, hb, fb), path = _topology_buffers(g)
    assert path == 0 and fa == 0 and fb == 0 and ha == hb
    # (the general build under the canonical flag skips the triplet mirror check only: same lists)
    g = fcc_cu_graph(3, 3, 3).to("cuda")
    ei = g[K.EDGE_INDEX].clone()
    ei[:, [0, -1]] = ei[:, [-1, 0]]   # first and last edge swapped: row 0 no longer sorted
    g[K.EDGE_INDEX] = ei
    (a, ha, fa), (b, hb, fb), path = _topology_buffers(g)
    assert path == 0 and (fa & 1) and (fb & 1)
    # rows of ~590 edges: beyond the in-edge kernel's LDS stage, the general build sorts instead (same buffer either way)
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    rng = np.random.default_rng(17)
    lat = np.array([[3.3, 0.2, 0.0], [-0.3, 3.4, 0.1], [0.2, -0.1, 3.2]])
    frac = rng.uniform(0, 1, (7, 3))
    g = batch_from_arrays([lat], [frac @ lat], [np.full(7, 28)], 9.0, 3.0, device="cuda")
    (a, ha, fa), (b, hb, fb), path = _topology_buffers(g)
    assert path == 0 and fa == 0 and fb == 0 and ha == hb and torch.equal(a, b)


# ------------------------------------------------------------------ the library's own scan and sort (csrc/m3g_prims.h)
@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 2047, 2048, 2049, 4096, 100_003, 2048 * 2048 + 5])
@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
def test_device_exclusive_scan_against_cumsum(n, dtype):
    """The exclusive scan behind the neighbour search and the list builders (tiles of 2,048, recursive tile totals): every length
    around the tile and recursion boundaries, both integer widths, in place and out of place -- integer results, bit-exact."""
    import ctypes as C

    from torch_m3gnet import _lib

    lib = _lib.load_library()
    gen = torch.Generator().manual_seed(n + (7 if dtype == torch.int64 else 0))
    x = torch.randint(0, 5 if dtype == torch.int32 else 1 << 33, (n,), dtype=dtype, generator=gen).cuda()
    want = torch.cumsum(x, 0) - x if n else x.clone()
    out = torch.full_like(x, -1)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.m3g_debug_exclusive_scan(x.element_size(), n, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), s))
    assert torch.equal(out, want)
    y = x.clone()
    _lib.check(lib.m3g_debug_exclusive_scan(y.element_size(), n, C.c_void_p(y.data_ptr()), C.c_void_p(y.data_ptr()), s))
    assert torch.equal(y, want)


@pytest.mark.parametrize("n,bits", [(1, 8), (255, 5), (256, 8), (257, 13), (4096, 16), (4097, 17), (70_001, 24), (1_000_003, 32)])
def test_device_radix_sort_is_a_stable_sort(n, bits):
    """The LSD radix sort of the general topology build (incoming-edge lists of asymmetric graphs, unsorted triplet lists): 32-bit
    keys with values -- stability checked through the values (a stable sort of (key, original position) pairs is unique) -- and
    64-bit keys, key bits [0, bits) and a window that does not start at bit 0."""
    import ctypes as C

    from torch_m3gnet import _lib

    lib = _lib.load_library()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator().manual_seed(n)
    keys = torch.randint(0, 1 << min(bits, 31), (n,), dtype=torch.int64, generator=gen)
    if bits < 16:
        keys = keys % 7        # many duplicates
    k32 = keys.to(torch.int32).cuda()
    vals = torch.arange(n, dtype=torch.int32).cuda()
    _lib.check(lib.m3g_debug_radix_sort(4, n, C.c_void_p(k32.data_ptr()), C.c_void_p(vals.data_ptr()), 0, min(bits, 31), s))
    want_k, want_i = torch.sort(keys, stable=True)
    assert torch.equal(k32.cpu().long(), want_k) and torch.equal(vals.cpu().long(), want_i)
    # 64-bit keys (triplet keys: (first edge << 32) | second edge), keys only
    hi = torch.randint(0, 1 << 20, (n,), dtype=torch.int64, generator=gen)
    k64 = ((hi << 32) | keys).cuda()
    want64 = torch.sort(k64.cpu())[0]
    _lib.check(lib.m3g_debug_radix_sort(8, n, C.c_void_p(k64.data_ptr()), None, 0, 52, s))
    assert torch.equal(k64.cpu(), want64)
    # a window of key bits that starts above bit 0: ordered by those bits only, ties in input order
    k = ((hi << 32) | keys).cuda()
    v = torch.arange(n, dtype=torch.int32).cuda()
    _lib.check(lib.m3g_debug_radix_sort(8, n, C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), 32, 52, s))
    _, want_i = torch.sort(hi, stable=True)
    assert torch.equal(v.cpu().long(), want_i)
