"""CPU-only checks of the host side: C-ABI export surface, drop-in API/state_dict parity, graph
construction, error behaviour, and the no-fallback rule.  No compute calls (there is no GPU here)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import GOLDEN

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from torch_m3gnet import _lib

    header = (ROOT / "include" / "m3gnet_hip.h").read_text()
    declared = set(re.findall(r"\b(m3g_[a-z_]+)\s*\(", header))
    declared -= {"m3g_plan"}  # type name
    lib = _lib.load_library()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/m3gnet_hip.h but not exported"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes prototype"
    info = _lib.M3GInfo()
    assert lib.m3g_get_info(C.byref(info)) == 0
    assert info.abi_version == _lib.ABI_VERSION


def test_struct_layouts_match_header_sizes():
    from torch_m3gnet import _lib

    assert C.sizeof(_lib.M3GConfig) == 4 * 8 + 6 * 4
    assert C.sizeof(_lib.M3GIO) == 4 * 8 + 17 * 8 + 8   # + topo_hints, reserved (int32 each)
    assert C.sizeof(_lib.M3GMdLists) == 5 * 8 + 3 * 8 + 21 * 8   # m3g_md_lists: 5 counts, 3 doubles, 17 pointers + 4 sizes
    assert C.sizeof(_lib.M3GMdResult) == 2 * 4 + 2 * 8 + 8


def test_seeded_build_model_reproduces_reference_weights_and_keys():
    """Same construction order as the reference => same RNG stream => identical initial weights."""
    from torch_m3gnet.model.build import build_model

    torch.manual_seed(0)
    model = build_model(5.0, 4.0, 3, 3, 95, 64, 3)
    z = np.load(GOLDEN / "model_default_seed0.npz")
    keys = [k for k in z.files if not k.startswith("__")]
    sd = model.state_dict()
    assert set(sd) == set(keys) and len(sd) == 80
    for k in keys:
        assert np.array_equal(sd[k].numpy(), z[k]), k
    assert sum(p.numel() for p in model.parameters()) == 227_549  # reference docs/architecture.md:50
    g = np.load(GOLDEN / "case_cu32_ref.npz")
    assert np.array_equal(model.model[4].em.numpy(), g["const_em"])
    assert np.array_equal(model.model[4].dm.numpy(), g["const_dm"])
    assert np.array_equal(model.model[4].coeff.numpy(), g["const_coeff"])
    gd = np.load(GOLDEN / "case_cu32_doc.npz")
    np.testing.assert_allclose(model.model[6].nsb.documented_factors().numpy(), gd["const_factors"], rtol=1e-6)


def test_small_model_keys_match_reference():
    from torch_m3gnet.model.build import build_model

    z = np.load(GOLDEN / "model_small_seed0.npz")
    torch.manual_seed(0)
    model = build_model(float(z["__cfg_cutoff"]), float(z["__cfg_threebody_cutoff"]), 2, 3, 93, 17, 2)
    for k in (k for k in z.files if not k.startswith("__")):
        assert np.array_equal(model.state_dict()[k].numpy(), z[k]), k


def test_reference_import_paths_exist():
    from torch_m3gnet.data import MaterialGraphKey  # noqa: F401
    from torch_m3gnet.data.material_graph import BatchMaterialGraph, MaterialGraph  # noqa: F401
    from torch_m3gnet.nn.atom_ref import AtomRef  # noqa: F401
    from torch_m3gnet.nn.conv import M3GNetConv  # noqa: F401
    from torch_m3gnet.nn.core import GatedMLP  # noqa: F401
    from torch_m3gnet.nn.featurizer import AtomFeaturizer, EdgeAdjustor, EdgeFeaturizer  # noqa: F401
    from torch_m3gnet.nn.gradient import Gradient  # noqa: F401
    from torch_m3gnet.nn.interaction import (SPHERICAL_BESSEL_ZEROS, ThreeBodyInteration, cutoff_function,  # noqa: F401
                                             legendre_cos, spherical_bessel)
    from torch_m3gnet.nn.invariant import DistanceAndAngle  # noqa: F401
    from torch_m3gnet.nn.readout import AtomWiseReadout  # noqa: F401
    from torch_m3gnet.nn.scale import ScaleLength  # noqa: F401

    assert len(SPHERICAL_BESSEL_ZEROS) == 10 and len(SPHERICAL_BESSEL_ZEROS[0]) == 10


def test_basis_functions_known_answers():
    """reference tests/test_basis.py: zeros, gradcheck, cutoff values -- on the package's host functions."""
    from torch_m3gnet.nn.interaction import SPHERICAL_BESSEL_ZEROS, cutoff_function, legendre_cos, spherical_bessel

    for order in range(10):
        for root in SPHERICAL_BESSEL_ZEROS[order]:
            torch.testing.assert_close(spherical_bessel(torch.tensor([root]), order), torch.tensor([0.0]))
    for order in range(4):
        x = torch.linspace(1e-1, 10, steps=16, dtype=torch.float64, requires_grad=True)
        assert torch.autograd.gradcheck(spherical_bessel, (x, order), eps=1e-4)
        c = torch.linspace(-1, 1, steps=16, dtype=torch.float64, requires_grad=True)
        assert torch.autograd.gradcheck(legendre_cos, (c, order), eps=1e-4)
    torch.testing.assert_close(cutoff_function(torch.tensor([0.0, 2.0, 4.0]), 2.0), torch.tensor([1.0, 0.0, 0.0]))


def _al_na():
    from torch_m3gnet.data.material_graph import MaterialGraph

    r = 3.0
    la, ln = r * np.sqrt(2) * np.eye(3), r / np.sqrt(3) * 2 * np.eye(3)
    al = MaterialGraph.from_arrays(la, np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0]]) @ la, [13] * 4, r + 1e-4, r + 1e-4)
    na = MaterialGraph.from_arrays(ln, np.array([[0, 0, 0], [.5, .5, .5]]) @ ln, [11] * 2, r + 1e-4, r + 1e-4)
    return al, na


def test_batch_collation_known_answers():
    """reference tests/test_data.py:10-23: batch vector, fcc 12*11 and bcc 8*7 triplets per atom."""
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.material_graph import Batch

    g = Batch.from_data_list(list(_al_na()))
    torch.testing.assert_close(g.batch, torch.tensor([0, 0, 0, 0, 1, 1]))
    assert g.num_nodes == 6 and g.pos.shape == (6, 3) and g.lattice.shape == (2, 3, 3)
    torch.testing.assert_close(g[K.NUM_TRIPLET_I], torch.tensor([132, 132, 132, 132, 56, 56]))
    # offsets of __inc__: second graph's edges point at atoms 4,5 and its triplets at edges >= 48
    assert int(g.edge_index[:, 48:].min()) == 4 and int(g.triplet_edge_index[:, 4 * 132:].min()) == 48


def test_threebody_index_matches_loop_restatement():
    """Vectorised enumeration == the plain triple loop of the reference's compute_threebody
    (data/material_graph.py:229-248), on a cell where some edges exceed the three-body cutoff."""
    from torch_m3gnet.data.neighbors import neighbor_list, threebody_index

    rng = np.random.default_rng(5)
    lat = np.eye(3) * 6.0
    pos = rng.uniform(0, 6.0, (12, 3))
    ei, shift, dist = neighbor_list(lat, pos, 4.5)
    tei, nti, ntij = threebody_index(12, ei, dist, 3.2)
    valid = np.nonzero(dist <= 3.2)[0]
    deg = np.bincount(ei[0][valid], minlength=12)
    loop = []
    off = 0
    for i in range(12):
        for j in range(deg[i]):
            for k in range(deg[i]):
                if j != k:
                    loop.append((valid[off + j], valid[off + k]))
        off += deg[i]
    assert np.array_equal(tei.T, np.array(loop).reshape(-1, 2))
    assert np.array_equal(nti, deg * (deg - 1))
    # geometry of the list itself
    r = pos[ei[1]] + shift @ lat - pos[ei[0]]
    np.testing.assert_allclose(np.linalg.norm(r, axis=1), dist, atol=1e-12)
    assert np.all(np.diff(ei[0]) >= 0) and dist.max() <= 4.5 + 1e-8


def test_neighbor_list_rotation_invariance_and_self_images():
    """Sorted distances are invariant under rotation of a strained cell (reference
    tests/test_invariance.py:41-66); tiny cells produce self-image edges i -> i."""
    from torch_m3gnet.data.neighbors import neighbor_list

    rng = np.random.default_rng(2)
    lat = np.eye(3) * 8.01 + 0.1 * rng.uniform(-1, 1, (3, 3))
    pos = rng.uniform(0, 8, (20, 3))
    rot = np.dot(np.array([[.5, np.sqrt(3) / 2, 0], [-np.sqrt(3) / 2, .5, 0], [0, 0, 1]]),
                 np.array([[0, 0, 1], [1 / np.sqrt(2), -1 / np.sqrt(2), 0], [1 / np.sqrt(2), 1 / np.sqrt(2), 0]]))
    _, _, d1 = neighbor_list(lat, pos, 5.0)
    _, _, d2 = neighbor_list(lat @ rot.T, pos @ rot.T, 5.0)
    np.testing.assert_allclose(np.sort(d1), np.sort(d2), atol=1e-9)
    ei, shift, _ = neighbor_list(np.eye(3) * 2.5, np.zeros((1, 3)), 3.6)
    assert ei.shape[1] == 18 and np.all(ei[0] == 0) and np.all(ei[1] == 0) and np.all(np.abs(shift).sum(1) > 0)


def test_error_behaviour_matches_reference():
    from torch_m3gnet.data.material_graph import MaterialGraph
    from torch_m3gnet.nn.interaction import NormalizedSphericalBessel

    with pytest.raises(ValueError, match="Too large l_max"):
        NormalizedSphericalBessel(5.0, 10, 3)  # nn/interaction.py:250-251
    with pytest.raises(ValueError, match="Too large n_max"):
        NormalizedSphericalBessel(5.0, 3, 11)  # nn/interaction.py:252-253
    with pytest.raises(ValueError, match="Three body cutoff"):
        MaterialGraph.from_arrays(np.eye(3) * 4, np.zeros((1, 3)), [1], 3.0, 3.5)  # material_graph.py:149-150


def test_c_abi_plan_error_codes():
    from torch_m3gnet import _lib

    lib = _lib.load_library()
    cfg = _lib.M3GConfig(5.0, 4.0, 1.0, 1.0, 3, 3, 95, 64, 3, 0)
    plan = C.c_void_p()
    assert lib.m3g_plan_create(C.byref(cfg), C.byref(plan)) == _lib.M3G_OK
    buf = np.zeros(64 * 95, dtype=np.float32)
    assert lib.m3g_plan_set_param(plan, b"model.3.linear.weight", buf.ctypes.data, buf.size) == _lib.M3G_OK
    assert lib.m3g_plan_set_param(plan, b"model.3.linear.weight", buf.ctypes.data, 7) == _lib.M3G_ERR_VALUE
    assert lib.m3g_plan_set_param(plan, b"model.99.nope", buf.ctypes.data, 7) == _lib.M3G_ERR_VALUE
    assert b"unknown parameter" in lib.m3g_last_error()
    assert lib.m3g_plan_commit(plan) == _lib.M3G_ERR_STATE  # parameters missing
    lib.m3g_plan_destroy(plan)
    bad = _lib.M3GConfig(5.0, 4.0, 1.0, 1.0, 10, 3, 95, 64, 3, 0)
    assert lib.m3g_plan_create(C.byref(bad), C.byref(plan)) == _lib.M3G_ERR_VALUE
    with pytest.raises(ValueError, match="Too large l_max"):
        _lib.check(_lib.M3G_ERR_VALUE)
    # sizes beyond the MFMA kernels' tiles are accepted (any-size path); only absurd ones are refused
    big = _lib.M3GConfig(5.0, 4.0, 1.0, 1.0, 9, 10, 95, 256, 12, 0)
    assert lib.m3g_plan_create(C.byref(big), C.byref(plan)) == _lib.M3G_OK
    assert lib.m3g_plan_set_option(plan, b"edge_kernel", 1) == _lib.M3G_ERR_UNSUPPORTED   # such a model cannot run on the MFMA kernels
    lib.m3g_plan_destroy(plan)
    huge = _lib.M3GConfig(5.0, 4.0, 1.0, 1.0, 3, 3, 95, 64, 33, 0)
    assert lib.m3g_plan_create(C.byref(huge), C.byref(plan)) == _lib.M3G_ERR_UNSUPPORTED
    too_many_n = _lib.M3GConfig(5.0, 4.0, 1.0, 1.0, 3, 11, 95, 64, 3, 0)
    assert lib.m3g_plan_create(C.byref(too_many_n), C.byref(plan)) == _lib.M3G_ERR_VALUE
    assert b"Too large n_max" in lib.m3g_last_error()
    tb = _lib.M3GConfig(4.0, 5.0, 1.0, 1.0, 3, 3, 95, 64, 3, 0)
    assert lib.m3g_plan_create(C.byref(tb), C.byref(plan)) == _lib.M3G_ERR_VALUE


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors / without a GPU -- never compute on the host."""
    from torch_m3gnet.data.material_graph import Batch
    from torch_m3gnet.model.build import build_model
    from torch_m3gnet.nn.featurizer import AtomFeaturizer
    from torch_m3gnet.nn.invariant import DistanceAndAngle
    from torch_m3gnet.nn.scale import ScaleLength

    g = Batch.from_data_list(list(_al_na()))
    model = build_model(3.0001, 3.0001, 2, 3, 93, 17, 2)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        model(g)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        torch.nn.Sequential(ScaleLength(1.0), DistanceAndAngle())(g)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        AtomFeaturizer(15, 8)(g)
    g["x"] = torch.zeros(6, 17)
    g["edge_attr"] = torch.zeros(int(g["edge_index"].size(1)), 17)
    with pytest.raises(RuntimeError, match="GPU tensor"):   # the stand-alone block modules have no host path either
        model.model[7](g)
    from torch_m3gnet.nn.core import GatedMLP

    with pytest.raises(RuntimeError, match="GPU tensor"):
        GatedMLP(4, [3])(torch.zeros(2, 4))


def test_unfused_layout_is_rejected():
    from torch_m3gnet.nn import modules as nn

    with pytest.raises(RuntimeError, match="module order of build_model"):
        nn.Gradient(torch.nn.Sequential(nn.ScaleLength(1.0), nn.DistanceAndAngle())).engine


def test_torch_free_c_abi_host_is_built_and_links():
    """`make` also builds tests/c_abi/m3g_c_abi_check.cpp, a host of the C ABI without torch (exercised on the GPU by
    tests/test_gpu_c_abi.py).  Here: the binary exists, resolves libm3gnet_hip.so through its rpath and prints its usage."""
    import subprocess
    from pathlib import Path

    for name in ("m3g_c_abi_check", "m3g_md_check"):   # (the second: the trajectory loop over m3g_md_*, tests/c_abi/m3g_md_check.cpp)
        binary = Path(__file__).resolve().parent.parent / "torch-m3gnet_amd" / "lib" / name
        assert binary.exists(), "run `make -C torch-m3gnet_amd` (or __graft_entry__.build())"
        proc = subprocess.run([str(binary)], capture_output=True, text=True, timeout=60)
        assert proc.returncode == 1 and "usage:" in proc.stderr, name


class _Lattice:
    def __init__(self, m):
        self.matrix = np.asarray(m, dtype=float)


class _Specie:
    def __init__(self, z):
        self.Z = int(z)


class _Site:
    def __init__(self, z):
        self.specie = _Specie(z)


class _Structure:
    """Duck-typed stand-in for pymatgen's Structure: the three attributes the reference reads (data/material_graph.py:139-147)."""

    def __init__(self, lattice, cart_coords, zs, with_numbers=True):
        self.lattice = _Lattice(lattice)
        self.cart_coords = np.asarray(cart_coords, dtype=float)
        self._sites = [_Site(z) for z in zs]
        if with_numbers:
            self.atomic_numbers = tuple(int(z) for z in zs)

    def __iter__(self):
        return iter(self._sites)

    def __len__(self):
        return len(self._sites)


def test_reference_data_layer_names_and_tuple_shapes():
    """Drop-in names of the reference's data layer (data/material_graph.py:132-254): `MaterialGraph.from_structure`,
    module-level `get_all_neighbors_with_cell_shifts` and `compute_threebody`, with the reference's tuple shapes and dtypes,
    on a duck-typed structure (no pymatgen in the image)."""
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.material_graph import MaterialGraph, compute_threebody, get_all_neighbors_with_cell_shifts

    rng = np.random.default_rng(11)
    lat = np.eye(3) * 6.2 + 0.3 * rng.uniform(-1, 1, (3, 3))
    pos = rng.uniform(0, 6, (10, 3))
    zs = rng.integers(1, 90, 10)
    for with_numbers in (True, False):   # Structure.atomic_numbers, or site.specie.Z as the reference iterates
        st = _Structure(lat, pos, zs, with_numbers=with_numbers)
        g = MaterialGraph.from_structure(st, 4.5, 3.5)
        ref = MaterialGraph.from_arrays(lat, pos, zs, 4.5, 3.5)
        for key in (K.POS, K.ATOM_TYPES, K.EDGE_INDEX, K.EDGE_CELL_SHIFT, K.TRIPLET_EDGE_INDEX, K.NUM_TRIPLET_I, K.NUM_TRIPLET_IJ, K.LATTICE):
            assert torch.equal(g[key], ref[key]), key
        assert g[K.ATOM_TYPES].dtype == torch.long and torch.equal(g[K.ATOM_TYPES], torch.tensor(zs - 1))
    ei, shift, dist = get_all_neighbors_with_cell_shifts(st, 4.5)
    assert ei.dtype == torch.long and ei.shape[0] == 2 and shift.dtype == torch.int and shift.shape == (ei.shape[1], 3)
    assert dist.dtype == torch.float and dist.shape == (ei.shape[1],)
    assert torch.equal(ei, g[K.EDGE_INDEX]) and torch.equal(shift, g[K.EDGE_CELL_SHIFT])
    tei, nti, ntij = compute_threebody(len(st), ei, dist, 3.5)
    assert tei.dtype == torch.long and ntij.dtype == torch.int
    assert torch.equal(tei, g[K.TRIPLET_EDGE_INDEX]) and torch.equal(nti, g[K.NUM_TRIPLET_I]) and torch.equal(ntij, g[K.NUM_TRIPLET_IJ])
    with pytest.raises(ValueError, match="Three body cutoff"):
        MaterialGraph.from_structure(st, 3.0, 3.5)


def test_model_config_builds_the_default_model():
    """`torch_m3gnet.config.ModelConfig`: the model half of the reference's RunConfig (config.py:10-17), used by bench.py."""
    from torch_m3gnet.config import ModelConfig
    from torch_m3gnet.model.build import build_model

    cfg = ModelConfig()
    assert (cfg.cutoff, cfg.threebody_cutoff, cfg.l_max, cfg.n_max, cfg.num_types, cfg.embedding_dim, cfg.num_blocks) == (5.0, 4.0, 3, 3, 95, 64, 3)
    torch.manual_seed(0)
    a = cfg.build()
    torch.manual_seed(0)
    b = build_model(5.0, 4.0, 3, 3, 95, 64, 3)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


def test_readout_depth_other_than_three_is_refused_loudly():
    """The reference's AtomWiseReadout takes any num_layers (nn/readout.py:31-37); build_model only ever passes 3
    (model/build.py:72) and that is what m3g_readout / the engine implement -- anything else must raise, not index the
    wrong tensors."""
    from torch_m3gnet.nn import modules as nn

    ro = nn.AtomWiseReadout(8, 2, 1.0)
    assert [k for k in ro.state_dict()] == ["gated.dense.0.weight", "gated.dense.0.bias", "gated.dense.2.weight", "gated.dense.2.bias",
                                            "gated.gate.0.weight", "gated.gate.0.bias", "gated.gate.2.weight", "gated.gate.2.bias"]

    class _OnGpu(torch.Tensor):   # the depth check comes right after the device check: a tensor that claims to live on the GPU
        @property
        def is_cuda(self):
            return True

    g = {"x": torch.zeros(2, 8).as_subclass(_OnGpu)}
    with pytest.raises(ValueError, match="num_layers = 3"):
        ro(g)
    head = [nn.ScaleLength(1.0), nn.AtomRef(torch.zeros(5)), nn.DistanceAndAngle(), nn.AtomFeaturizer(5, 8), nn.EdgeFeaturizer(3, 5.0),
            nn.EdgeAdjustor(3, 8)]
    with pytest.raises(ValueError, match="num_layers = 3"):
        nn.Gradient(torch.nn.Sequential(*head, nn.AtomWiseReadout(8, 4, 1.0))).engine


def test_host_neighbor_order_does_not_depend_on_the_wrap_state():
    """Canonical edge order (round 4): centre, cell shift relative to the GIVEN coordinates, neighbour.  An atom drifting across a
    cell face changes neither the shifts nor the order of any edge; moving an atom by a whole lattice vector relabels the shifts of
    its edges (they refer to the coordinates given) and nothing else: same pair vectors, same distances, same triplet count."""
    from torch_m3gnet.data.neighbors import neighbor_list, threebody_index

    rng = np.random.default_rng(1)
    lat = np.array([[6.1, 0.2, 0.0], [-0.4, 5.8, 0.3], [0.1, -0.2, 6.4]])
    pos = rng.uniform(0.05, 0.95, (14, 3)) @ lat
    a, b = pos.copy(), pos.copy()
    a[0] = np.array([-0.001, 0.3, 0.999]) @ lat      # just outside two faces
    b[0] = np.array([+0.001, 0.3, 1.001]) @ lat      # just inside / outside the other way
    ea, sa, da = neighbor_list(lat, a, 5.0)
    eb, sb, db = neighbor_list(lat, b, 5.0)
    keep_a = np.abs(da - 5.0) > 0.05                  # (pairs within 0.05 A of the cutoff may enter / leave under the 0.02 A move)
    keep_b = np.abs(db - 5.0) > 0.05
    assert np.array_equal(ea[:, keep_a], eb[:, keep_b]) and np.array_equal(sa[keep_a], sb[keep_b])
    c = pos.copy()
    c[3] += 2 * lat[0] - lat[2]                       # a whole-lattice-vector move
    e0, s0, d0 = neighbor_list(lat, pos, 5.0)
    ec, sc, dc = neighbor_list(lat, c, 5.0)
    vec = lambda p, e, s: p[e[1]] + s @ lat - p[e[0]]   # noqa: E731
    k0 = sorted(map(tuple, np.round(np.c_[e0.T, vec(pos, e0, s0)], 9)))
    kc = sorted(map(tuple, np.round(np.c_[ec.T, vec(c, ec, sc)], 9)))
    assert k0 == kc
    t0 = threebody_index(14, e0, d0.astype(np.float32), 4.0)[0].shape[1]
    tc = threebody_index(14, ec, dc.astype(np.float32), 4.0)[0].shape[1]
    assert t0 == tc


def test_engine_parameter_signature_sees_every_kind_of_change():
    """Engine._signature decides on every call whether the plan's weights must be committed again (engine.py).  It is built from
    cached dicts with C-level maps (the host's critical path of a caller that waits for the device each step); it must still see:
    an in-place update (optimiser step, load_state_dict), a replaced Parameter, a replaced submodule, a Parameter registered later
    on a module that had none, new captured constants -- and nothing when nothing changed."""
    import torch

    from torch_m3gnet.model.build import build_model

    torch.manual_seed(0)
    model = build_model(5.0, 4.0, 3, 3, 95, 64, 3)
    eng = model.engine

    class Dev:
        index = 0

    s0 = eng._signature(Dev)
    assert eng._signature(Dev) == s0
    with torch.no_grad():
        next(model.parameters()).add_(1.0)
    s1 = eng._signature(Dev)
    assert s1 != s0
    model.load_state_dict(model.state_dict())            # copy_ into every parameter: versions move
    s2 = eng._signature(Dev)
    assert s2 != s1
    lin = model.model[3].linear
    lin.weight = torch.nn.Parameter(torch.zeros_like(lin.weight))
    s3 = eng._signature(Dev)
    assert s3 != s2
    model.model[3].linear = torch.nn.Linear(95, 64, bias=False)
    s4 = eng._signature(Dev)
    assert s4 != s3
    model.model[3].linear.register_parameter("bias", torch.nn.Parameter(torch.zeros(64)))
    s5 = eng._signature(Dev)
    assert s5 != s4 and len(s5[1]) == len(s4[1]) + 1
    tb = [m for m in model.model if type(m).__name__ == "ThreeBodyInteration"][0]
    tb.nsb.factors = tb.nsb.factors.clone()
    s6 = eng._signature(Dev)
    assert s6 != s5

    class Dev1:
        index = 1

    assert eng._signature(Dev1) != s6 and eng._signature(Dev) == s6
