"""GPU: the reference's own property tests re-expressed against the HIP engine (tests/test_model.py,
tests/test_invariance.py, tests/test_nn.py of the reference), edge cases, and BASELINE.json's full-size
configurations checked through size-independent properties and against the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import build_engine_model, engine_graph, fcc_cu_graph, load_oracle_case, random_cell_graph, rel_err
from oracle import m3gnet_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _K():
    from torch_m3gnet.data import MaterialGraphKey as K

    return K


def _small_model(seed=0):
    from torch_m3gnet.model.build import build_model

    torch.manual_seed(seed)
    return build_model(3.0001, 3.0001, 2, 3, 93, 17, 2)  # reference tests/conftest.py:150-178


def _default_model(seed=0, **kw):
    from torch_m3gnet.model.build import build_model

    torch.manual_seed(seed)
    args = dict(cutoff=5.0, threebody_cutoff=4.0, l_max=3, n_max=3, num_types=95, embedding_dim=64, num_blocks=3)
    args.update(kw)
    model = build_model(**args)
    for m in model.model:  # documented chi so the three-body path carries weight
        if type(m).__name__ == "ThreeBodyInteration":
            m.nsb.factors = m.nsb.documented_factors()
    return model


def _al_na(perturb=True):
    from torch_m3gnet.data.material_graph import MaterialGraph

    r = 3.0
    la, ln = r * np.sqrt(2) * np.eye(3), r / np.sqrt(3) * 2 * np.eye(3)
    al = MaterialGraph.from_arrays(la, np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0]]) @ la, [13] * 4, r + 1e-4, r + 1e-4)
    na = MaterialGraph.from_arrays(ln, np.array([[0, 0, 0], [.5, .5, .5]]) @ ln, [11] * 2, r + 1e-4, r + 1e-4)
    if perturb:
        torch.manual_seed(0)
        for g in (al, na):
            g["pos"] = g["pos"] + 1e-1 * (torch.rand(g["pos"].shape) - 0.5)
    return [al, na]


def _oracle_inputs(model, graph):
    """(params, cfg, consts, graph dict) of the CPU oracle for a torch_m3gnet model (checker only)."""
    from oracle import m3gnet_oracle as orc

    seq = model.model
    n_blocks = (len(seq) - 7) // 2
    tb = seq[6]
    ls = float(seq[0].length_scale)
    cfg = orc.OracleConfig(cutoff=float(seq[4].cutoff) * ls, threebody_cutoff=float(tb.threebody_cutoff) * ls, l_max=tb.l_max,
                           n_max=tb.n_max, num_types=seq[3].num_types, embedding_dim=seq[3].linear.out_features,
                           num_blocks=n_blocks, energy_scale=float(seq[-1].scale), length_scale=ls)
    consts = orc.make_constants(cfg, seq[1].elemental_energies.cpu())
    consts.factors = tb.nsb.factors.clone().cpu()
    params = {f"model.{k}": v.detach().cpu().clone() for k, v in seq.state_dict().items()}
    g = {k: graph[k].cpu() for k in ("pos", "atom_types", "edge_index", "edge_cell_shift", "triplet_edge_index", "lattice", "batch")}
    return params, cfg, consts, g


# ------------------------------------------------------------------ reference tests/test_model.py
def test_model_outputs_finite():
    from torch_m3gnet.data.material_graph import Batch

    K = _K()
    g = _small_model()(Batch.from_data_list(_al_na(False)).to(DEV))
    for key in (K.NODE_FEATURES, K.EDGE_ATTR, K.SCALED_ATOMIC_ENERGIES, K.FORCES, K.STRESSES):
        assert torch.isfinite(g[key]).all(), key


def test_three_body_interaction_invariant_to_triplet_order():
    """reference tests/test_model.py:21-38.  The topology build canonicalises the triplet order, so the
    three-body aggregate itself is bitwise identical and so is everything on the force path (no atomics there); only the
    per-structure energy sums and the virial use float atomics, hence assert_close (as the reference does) over all keys."""
    from torch_m3gnet.data.material_graph import Batch

    K = _K()
    model = _default_model()
    g = Batch.from_data_list([random_cell_graph(20, 6.5, 3)]).to(DEV)
    first = model(g)
    out1 = {k: first[k].clone() for k in (K.EDGE_ATTR, K.FORCES, K.TOTAL_ENERGY, K.MID_EDGE_FEATURES)}
    perm = torch.randperm(g[K.TRIPLET_EDGE_INDEX].size(1), device=DEV)
    g[K.TRIPLET_EDGE_INDEX][0] = g[K.TRIPLET_EDGE_INDEX][0][perm]
    g[K.TRIPLET_EDGE_INDEX][1] = g[K.TRIPLET_EDGE_INDEX][1][perm]
    out2 = model(g)
    assert torch.equal(out1[K.MID_EDGE_FEATURES][0], out2[K.MID_EDGE_FEATURES][0])
    for k, v in out1.items():
        torch.testing.assert_close(v, out2[k], rtol=1e-5, atol=1e-7)
    assert not torch.isnan(out1[K.EDGE_ATTR]).any()


def test_rotation_invariance_of_node_features():
    """reference tests/test_model.py:41-56 (Ti8O24 cell, fixed rotation)."""
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph

    K = _K()
    _, _, _, graph, _ = load_oracle_case("tio", "ref")
    lat = graph["lattice"][0].double().numpy()
    pos = graph["pos"].double().numpy()
    z = graph["atom_types"].numpy() + 1
    rot = np.dot(np.array([[.5, np.sqrt(3) / 2, 0], [-np.sqrt(3) / 2, .5, 0], [0, 0, 1]]),
                 np.array([[0, 0, 1], [1 / np.sqrt(2), -1 / np.sqrt(2), 0], [1 / np.sqrt(2), 1 / np.sqrt(2), 0]]))
    model = _default_model()
    g1 = model(Batch.from_data_list([MaterialGraph.from_arrays(lat, pos, z, 5.0, 4.0)]).to(DEV))
    g2 = model(Batch.from_data_list([MaterialGraph.from_arrays(lat @ rot.T, pos @ rot.T, z, 5.0, 4.0)]).to(DEV))
    torch.testing.assert_close(g1[K.NODE_FEATURES], g2[K.NODE_FEATURES], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(g1[K.TOTAL_ENERGY], g2[K.TOTAL_ENERGY], rtol=1e-5, atol=1e-6)
    f1 = g1[K.FORCES].double().cpu().numpy() @ rot.T
    assert np.abs(f1 - g2[K.FORCES].double().cpu().numpy()).max() < 1e-4 * np.abs(f1).max() + 1e-7


def test_batched_equals_per_graph():
    """reference tests/test_model.py:59-78 and tests/test_invariance.py:15-38."""
    from torch_m3gnet.data.material_graph import Batch

    K = _K()
    model = _small_model()
    graphs = _al_na()
    gb = model(Batch.from_data_list([g.clone() for g in graphs]).to(DEV))
    singles = [model(Batch.from_data_list([g.clone()]).to(DEV)) for g in graphs]
    torch.testing.assert_close(gb[K.TOTAL_ENERGY], torch.cat([s[K.TOTAL_ENERGY] for s in singles]))
    torch.testing.assert_close(gb[K.FORCES], torch.cat([s[K.FORCES] for s in singles]))
    torch.testing.assert_close(gb[K.EDGE_DISTANCES], torch.cat([s[K.EDGE_DISTANCES] for s in singles]))
    torch.testing.assert_close(gb[K.TRIPLET_ANGLES], torch.cat([s[K.TRIPLET_ANGLES] for s in singles]))


def test_forces_against_finite_differences_of_engine_energy():
    """reference tests/test_model.py:90-120: central differences, delta 1e-2, atol 1e-3 / rtol 1e-2."""
    from torch_m3gnet.data.material_graph import Batch

    K = _K()
    model = _small_model()
    g0 = Batch.from_data_list(_al_na()).to(DEV)
    forces = model(g0.clone())[K.FORCES].cpu()
    delta = 1e-2
    for node in range(6):
        s = int(g0[K.BATCH][node])
        for axis in range(3):
            e = []
            for sign in (+1, -1):
                g = g0.clone()
                g[K.POS][node, axis] += sign * delta
                e.append(float(model(g, forces=False, extras=False)[K.TOTAL_ENERGY][s]))
            torch.testing.assert_close(forces[node, axis], torch.tensor(-(e[0] - e[1]) / (2 * delta)), atol=1e-3, rtol=1e-2)


# ------------------------------------------------------------------ reference tests/test_invariance.py / test_nn.py
def test_standalone_distance_angle_and_featurizers():
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph
    from torch_m3gnet.nn.atom_ref import AtomRef
    from torch_m3gnet.nn.featurizer import AtomFeaturizer, EdgeFeaturizer
    from torch_m3gnet.nn.invariant import DistanceAndAngle
    from torch_m3gnet.nn.scale import ScaleLength

    K = _K()
    g = Batch.from_data_list(_al_na(False)).to(DEV)
    seq = torch.nn.Sequential(ScaleLength(1.0), DistanceAndAngle(), EdgeFeaturizer(degree=3, cutoff=5.0))
    g = seq(g)
    ang = g[K.TRIPLET_ANGLES]
    assert (ang <= 1).all() and (ang >= -1).all()
    theta_sum = float(torch.arccos(ang.double()).sum()) / np.pi
    # reference tests/test_invariance.py:69-82, same bound (atol 1e-4 in units of pi): the reported cosines of collinear
    # triplets are snapped to exactly +-1 (k_triplet_angles), as the reference's clamp does for the half of them that land
    # outside [-1, 1]
    assert abs(theta_sum - round(theta_sum)) < 1e-4
    torch.testing.assert_close(g[K.EDGE_DISTANCES], torch.full_like(g[K.EDGE_DISTANCES], 3.0), rtol=1e-5, atol=1e-5)
    assert g[K.EDGE_WEIGHTS].shape[1] == 3 and torch.isfinite(g[K.EDGE_WEIGHTS]).all()
    # oracle cross-check of the standalone stage outputs
    from oracle import m3gnet_oracle as orc

    cfg = orc.OracleConfig(cutoff=5.0, n_max=3)
    ew = orc.radial_basis(g[K.EDGE_DISTANCES].cpu(), cfg, orc.make_constants(cfg))
    assert rel_err(g[K.EDGE_WEIGHTS], ew) < 1e-5
    g = AtomFeaturizer(num_types=15, embedding_dim=64)(g)
    assert g[K.NODE_FEATURES].shape == (6, 64) and torch.isfinite(g[K.NODE_FEATURES]).all()
    # AtomRef known answer (reference tests/test_nn.py:16-30): Li Li H H with energies [2,0,3] -> 10
    lih = MaterialGraph.from_arrays(np.eye(3), np.array([[0, 0, 0], [.5, .5, .5], [.5, 0, 0], [0, .5, .5]]), [3, 3, 1, 1], 1, 1)
    out = AtomRef(torch.tensor([2, 0, 3]))(Batch.from_data_list([lih]).to(DEV))
    assert abs(float(out[K.ELEMENTAL_ENERGIES].sum()) - 10.0) < 1e-6


# ------------------------------------------------------------------ edge cases
def test_malformed_graphs_raise_value_error():
    from torch_m3gnet.data.material_graph import Batch

    K = _K()
    model = _small_model()
    g = Batch.from_data_list(_al_na()).to(DEV)
    bad = g.clone()
    bad[K.EDGE_INDEX] = bad[K.EDGE_INDEX].flip(1)
    with pytest.raises(ValueError, match="sorted by centre"):
        model(bad)
    bad = g.clone()
    bad[K.TRIPLET_EDGE_INDEX][1, 0] = 10_000
    with pytest.raises(ValueError, match="out of range"):
        model(bad)


def test_empty_and_ragged_inputs():
    """No triplets at all, no edges at all, and a batch mixing an isolated atom with a dense cell."""
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph

    K = _K()
    model = _default_model()
    lone = MaterialGraph.from_arrays(np.eye(3) * 20.0, np.array([[1.0, 2.0, 3.0]]), [8], 5.0, 4.0)  # no edges
    pair = MaterialGraph.from_arrays(np.eye(3) * 20.0, np.array([[0, 0, 0], [0, 0, 4.5]]), [8, 1], 5.0, 4.0)  # edges, no triplets
    dense = random_cell_graph(12, 5.2, 9)
    assert lone[K.NUM_EDGES] == 0 and pair[K.NUM_EDGES] == 2 and pair[K.NUM_TRIPLETS] == 0
    for graphs in ([lone], [pair], [lone, dense, pair]):
        g = model(Batch.from_data_list([x.clone() for x in graphs]).to(DEV))
        assert torch.isfinite(g[K.TOTAL_ENERGY]).all() and torch.isfinite(g[K.FORCES]).all()
        p, cfg, c, og = _oracle_inputs(model, g)
        from oracle import m3gnet_oracle as orc

        o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
        assert rel_err(g[K.TOTAL_ENERGY], o["total_energy"]) < 1e-5
        assert float((g[K.FORCES].cpu() - o["forces"]).abs().max()) < 1e-4 * float(o["forces"].abs().max()) + 1e-9
    assert float(model(Batch.from_data_list([lone]).to(DEV))[K.FORCES].abs().max()) == 0.0


def test_dense_three_body_rows_beyond_the_staged_window():
    """~100 three-body partners per centre: the rows a three-body workgroup stages in LDS (kTbCap) and its byte-sized
    partner lists (32 per row) both overflow, so partners are fetched through the global-memory path of
    csrc/m3g_threebody.hip.  Same tolerances as every other parity case."""
    from torch_m3gnet.data.material_graph import Batch
    from oracle import m3gnet_oracle as orc

    K = _K()
    model = _default_model(seed=2, threebody_cutoff=5.0)
    dense = random_cell_graph(24, 5.0, 5, cutoff=5.0, tb_cutoff=5.0, dmin=1.2)
    per_centre = dense[K.NUM_TRIPLETS] / 24
    assert per_centre > 80 * 79, per_centre
    for graphs in ([dense], [random_cell_graph(6, 5.5, 1), dense]):
        g = model(Batch.from_data_list([x.clone() for x in graphs]).to(DEV))
        p, cfg, c, og = _oracle_inputs(model, g)
        o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
        assert rel_err(g[K.TOTAL_ENERGY], o["total_energy"]) < 1e-5
        assert rel_err(g[K.FORCES], o["forces"]) < 1e-4
        assert rel_err(g[K.STRESSES], o["stresses"]) < 1e-4


def test_scales_and_elemental_energies():
    from torch_m3gnet.data.material_graph import Batch
    from oracle import m3gnet_oracle as orc

    K = _K()
    model = _default_model(seed=4, energy_scale=3.5, length_scale=1.3, elemental_energies=torch.linspace(-2, 1, 95))
    g = model(Batch.from_data_list([random_cell_graph(16, 6.0, s) for s in (1, 2)]).to(DEV))
    p, cfg, c, og = _oracle_inputs(model, g)
    o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
    assert rel_err(g[K.TOTAL_ENERGY], o["total_energy"]) < 1e-5
    assert rel_err(g[K.FORCES], o["forces"]) < 1e-4
    assert rel_err(g[K.STRESSES], o["stresses"]) < 1e-4


def test_out_of_range_species_raise_and_leading_module_keys_are_published():
    """The reference raises IndexError from `elemental_energies[atom_types]` (nn/atom_ref.py:27) for a species index outside
    the model's table; the kernels would otherwise clamp it silently.  With `extras` the fused call also leaves the keys of
    the leading modules on the graph (scaled_pos, scaled_lattice: nn/scale.py:24-29; elemental_energies: nn/atom_ref.py)."""
    from torch_m3gnet.data.material_graph import Batch

    K = _K()
    model = _default_model(num_types=30, length_scale=1.7, elemental_energies=torch.linspace(-1, 1, 30))
    g = Batch.from_data_list([random_cell_graph(12, 6.0, 3, zmax=30)]).to(DEV)
    out = model(g)
    torch.testing.assert_close(out[K.SCALED_POS], g[K.POS] / 1.7)
    torch.testing.assert_close(out[K.SCALED_LATTICE], g[K.LATTICE] / 1.7)
    torch.testing.assert_close(out[K.ELEMENTAL_ENERGIES], torch.linspace(-1, 1, 30, device=DEV)[g[K.ATOM_TYPES]])
    bad = g.clone()
    bad[K.ATOM_TYPES] = bad[K.ATOM_TYPES].clone()
    bad[K.ATOM_TYPES][3] = 30
    with pytest.raises(IndexError):
        model(bad)
    bad[K.ATOM_TYPES][3] = -1
    with pytest.raises(IndexError):
        model(bad)


def test_unsorted_batch_vector_gives_the_same_energies_and_forces():
    """The reference's scatter_sum accepts any `batch` vector (nn/readout.py:49-53, nn/gradient.py:41); its own batching always
    emits a sorted one, which is what the atomics-free per-structure sums rely on.  With the atoms of two structures interleaved
    the float-atomic fallback kernels take over (topology flag): same energies, same forces up to the relabelling."""
    from torch_m3gnet.data.material_graph import Batch

    K = _K()
    model = _default_model(seed=1, energy_scale=1.5)
    g = Batch.from_data_list([random_cell_graph(13, 6.0, 11), random_cell_graph(17, 6.4, 12)])
    ref = model(g.clone().to(DEV))
    n = int(g[K.NUM_NODES])
    new_index = torch.randperm(n, generator=torch.Generator().manual_seed(3))     # new label of old atom a
    src, dst = new_index[g[K.EDGE_INDEX][0]], new_index[g[K.EDGE_INDEX][1]]
    order = torch.argsort(src, stable=True)                                        # edges stay sorted by (new) centre
    inv_order = torch.empty_like(order)
    inv_order[order] = torch.arange(order.numel())
    h = g.clone()
    for key in (K.POS, K.ATOM_TYPES, K.BATCH, K.NUM_TRIPLET_I):
        out = torch.empty_like(g[key])
        out[new_index] = g[key]
        h[key] = out
    h[K.EDGE_INDEX] = torch.stack([src[order], dst[order]])
    h[K.EDGE_CELL_SHIFT] = g[K.EDGE_CELL_SHIFT][order]
    h[K.NUM_TRIPLET_IJ] = g[K.NUM_TRIPLET_IJ][order]
    h[K.TRIPLET_EDGE_INDEX] = inv_order[g[K.TRIPLET_EDGE_INDEX]]
    assert not bool((h[K.BATCH][1:] >= h[K.BATCH][:-1]).all())                      # really unsorted
    out = model(h.to(DEV))
    torch.testing.assert_close(out[K.TOTAL_ENERGY], ref[K.TOTAL_ENERGY], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out[K.FORCES], ref[K.FORCES][torch.argsort(new_index).to(DEV)], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(out[K.STRESSES], ref[K.STRESSES], rtol=1e-4, atol=1e-8)


def test_one_sided_and_filtered_triplet_lists():
    """The reference's gather + scatter_sum (nn/interaction.py:188-217) accepts ANY list of (e1, e2) edge pairs sharing a
    centre, not only the symmetric list `compute_threebody` emits.  One-sided (e1 < e2 only) and randomly thinned lists
    contain edges that only ever appear as the partner e2: the engine must keep their rows (they were dropped before the
    round-2 topology fix, silently corrupting energies and forces)."""
    from torch_m3gnet.data.material_graph import Batch

    K = _K()
    model = _default_model(seed=2, energy_scale=2.0)
    base = Batch.from_data_list([random_cell_graph(20, 6.5, s) for s in (7, 8)])
    tei = base[K.TRIPLET_EDGE_INDEX]
    gen = torch.Generator().manual_seed(0)
    variants = {
        "one_sided": tei[:, tei[0] < tei[1]],
        "thinned": tei[:, torch.rand(tei.size(1), generator=gen) < 0.3],
        "single_pair": tei[:, :1],
    }
    for name, sub in variants.items():
        g = base.clone()
        g[K.TRIPLET_EDGE_INDEX] = sub.contiguous()
        g[K.NUM_TRIPLETS] = int(sub.size(1))
        only_e2 = set(sub[1].tolist()) - set(sub[0].tolist())
        assert name == "single_pair" or len(only_e2) > 0     # the case the fix is about really occurs
        out = model(g.to(DEV))
        p, cfg, c, og = _oracle_inputs(model, out)
        p = {k: v.double() for k, v in p.items()}
        c = orc.make_constants(cfg, model.model[1].elemental_energies.cpu(), dtype=torch.float64)
        c.factors = model.model[6].nsb.factors.double()
        o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
        assert float(((out[K.TOTAL_ENERGY].cpu().double() - o["total_energy"]).abs() / o["total_energy"].abs()).max()) < 1e-5, name
        assert rel_err(out[K.FORCES], o["forces"]) < 1e-4, name
        # (a single triplet leaves nothing to average over: its one fp32 Bessel x Legendre x envelope product is compared
        # entry by entry with the fp64 oracle, measured 1.3e-4 on the worst entry)
        m_tol = 1e-3 if name == "single_pair" else 1e-4
        for b in range(3):
            assert rel_err(out[K.MID_EDGE_FEATURES][b], o[f"mid_edge_features_{b}"]) < m_tol, (name, b)


def test_asymmetric_edge_lists_take_the_sorted_incoming_lists():
    """The incoming-edge lists of a SYMMETRIC edge list come from the mirrors of each atom's own row (k_in_edges_symmetric, round 4);
    the reference's modules accept any centre-sorted edge list (`edge_index` is only ever gathered from and scattered by,
    nn/conv.py:63-97), so a list with edges removed on one side -- i -> j kept, j -> i dropped -- must fall back to the stable
    sort and still match the oracle.  Also: the same graph evaluated with a symmetric list afterwards (mirror path) agrees with
    the oracle too, and a multigraph cell (several images of the same neighbour) runs the mirror path on duplicate (i, j) pairs."""
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data.material_graph import Batch
    from torch_m3gnet.data.neighbors import threebody_index

    K = _K()
    model = _default_model(seed=4, energy_scale=1.5)
    cases = {"thinned": random_cell_graph(24, 7.0, 21), "multigraph": random_cell_graph(5, 3.9, 22, dmin=1.8)}
    for name, cell in cases.items():
        base = Batch.from_data_list([cell])
        ei, shift = base[K.EDGE_INDEX], base[K.EDGE_CELL_SHIFT]
        variants = {"symmetric": torch.ones(ei.size(1), dtype=torch.bool)}
        if name == "thinned":
            keep = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(1)) < 0.85   # drops one direction of many pairs
            variants["asymmetric"] = keep
        for vname, keep in variants.items():
            g = base.clone()
            g[K.EDGE_INDEX] = ei[:, keep].contiguous()
            g[K.EDGE_CELL_SHIFT] = shift[keep].contiguous()
            # triplets of the remaining edges (distances from the positions, as the builders form them)
            pos, lat = base[K.POS].double(), base[K.LATTICE][0].double()
            src, dst = g[K.EDGE_INDEX]
            d = (pos[dst] + g[K.EDGE_CELL_SHIFT].double() @ lat - pos[src]).norm(dim=1)
            tei, nti, ntij = threebody_index(int(pos.size(0)), g[K.EDGE_INDEX].numpy(), d.float().numpy(), 4.0)
            g[K.TRIPLET_EDGE_INDEX] = torch.from_numpy(tei)
            g[K.NUM_TRIPLET_I], g[K.NUM_TRIPLET_IJ] = torch.from_numpy(nti), torch.from_numpy(ntij)
            g[K.NUM_EDGES], g[K.NUM_TRIPLETS] = int(g[K.EDGE_INDEX].size(1)), int(tei.shape[1])
            out = model(g.to(DEV))
            p, cfg, c, og = _oracle_inputs(model, out)
            p = {k: v.double() for k, v in p.items()}
            c = orc.make_constants(cfg, model.model[1].elemental_energies.cpu(), dtype=torch.float64)
            c.factors = model.model[6].nsb.factors.double()
            o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
            assert float(((out[K.TOTAL_ENERGY].cpu().double() - o["total_energy"]).abs() / o["total_energy"].abs()).max()) < 1e-5, (name, vname)
            assert rel_err(out[K.FORCES], o["forces"]) < 1e-4, (name, vname)
            assert rel_err(out[K.NODE_FEATURES], o["x"]) < 1e-5, (name, vname)


def test_topology_hints_tell_complete_lists_from_the_rest():
    """m3g_topology_hints (include/m3gnet_hip.h): bit 0 is set exactly when every centre's triplet list holds each ordered
    pair of its active edges once -- what compute_threebody emits (data/material_graph.py:196-254) -- and then the word also
    carries the largest three-body window (rows, atoms) the moment kernels size their LDS by.  Any other list (one-sided,
    thinned, a duplicated pair, a permuted-but-complete list is still complete) must say 0 or the moment kernels would be
    wrong for it."""
    from torch_m3gnet.data.material_graph import Batch
    from torch_m3gnet.nn.modules import _Topology

    K = _K()
    base = Batch.from_data_list([random_cell_graph(20, 6.5, s) for s in (7, 8)]).to(DEV)
    tei = base[K.TRIPLET_EDGE_INDEX]

    def hints_of(sub):
        g = base.clone()
        g[K.TRIPLET_EDGE_INDEX] = sub.contiguous()
        g[K.NUM_TRIPLETS] = int(sub.size(1))
        return _Topology(g).query_hints()

    h = hints_of(tei)
    rows, atoms = (h >> 8) & 0xFF, (h >> 16) & 0xFF
    assert h & 1 and 0 < rows <= 255 and 0 < atoms <= 64
    perm = torch.randperm(tei.size(1), generator=torch.Generator().manual_seed(1)).to(tei.device)
    assert hints_of(tei[:, perm]) == h                       # order of the list does not matter
    assert hints_of(tei[:, tei[0] < tei[1]]) == 0            # one-sided
    assert hints_of(tei[:, 1:]) == 0                         # one pair missing
    assert hints_of(torch.cat([tei, tei[:, :1]], dim=1)) == 0   # one pair twice
    g0 = base.clone()
    g0[K.TRIPLET_EDGE_INDEX] = tei[:, :0].contiguous()
    g0[K.NUM_TRIPLETS] = 0
    assert _Topology(g0).query_hints() == 0                  # no triplets at all


def test_hints_word_of_another_topology_is_refused_not_trusted():
    """The three-body moment kernels size their LDS from m3g_io.topo_hints.  A word that was not certified for the topology buffer
    of the call (never asked for, stale after a rebuild, copied from another buffer) must not be trusted: the kernels compare it
    with the word m3g_topology_hints left ON the buffer, touch nothing when they differ and flag M3G_TOPO_ERR_HINTS
    (m3g_topology_status) -- no out-of-bounds access, no silent use of uninitialised window records."""
    from torch_m3gnet.data.material_graph import Batch
    from torch_m3gnet.nn.modules import _Topology

    K = _K()
    model = _default_model()
    g = Batch.from_data_list([random_cell_graph(20, 6.5, s) for s in (7, 8)]).to(DEV)
    topo = _Topology.of(g)
    assert topo.status() == 0
    good = model(g)
    e_good = good[K.TOTAL_ENERGY].clone()
    assert topo.query_hints() & 1 and topo.status() == 0
    # a fresh topology of the same graph whose certificate was never formed, called with the (plausible) word of the other one
    g2 = g.clone()
    g2[K.EDGE_INDEX] = g2[K.EDGE_INDEX].clone()
    _Topology.WITH_HINTS = False          # a plain m3g_topology_build, as a C-ABI caller that never asks for the certificate makes it
    try:
        topo2 = _Topology.of(g2)
    finally:
        _Topology.WITH_HINTS = True
    assert topo2 is not topo and topo2._hints is None
    topo2._hints = topo.query_hints()
    model(g2)
    torch.cuda.synchronize()
    assert topo2.status() == 1          # M3G_TOPO_ERR_HINTS: results of that call are invalid, and say so
    assert topo.status() == 0
    # certified properly, the same buffer runs the moment kernels and agrees bit for bit with the first graph
    topo3 = _Topology(g2)
    dict.__setitem__(g2, "_m3g_topology", (_Topology.signature(g2), topo3))
    out = model(g2)
    assert topo3.status() == 0 and torch.equal(out[K.TOTAL_ENERGY], e_good)


# ------------------------------------------------------------------ BASELINE.json configurations
@pytest.fixture(scope="module")
def config3_oracle():
    """One CPU-oracle evaluation of the 10,000-atom cell (fp32 torch on the host, ~10 s), shared by the three engine modes."""
    from oracle import m3gnet_oracle as orc

    model = _default_model()
    g = fcc_cu_graph(10, 10, 25).to(DEV)
    out = model(g, forces=False)   # only to hand the oracle the same inputs and constants
    torch.set_num_threads(8)
    p, cfg, c, og = _oracle_inputs(model, out)
    o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
    return {k: o[k] for k in ("scaled_atomic_energies", "forces", "mid_edge_features_0")}, cfg.energy_scale


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "bf16x3"])
def test_config3_10k_atom_cu_supercell_vs_oracle(config3_oracle, precision):
    """10,000-atom Cu supercell (BASELINE config 3, the size the headline is timed at): full-size parity of every arithmetic
    mode -- the default exact-fp32 one first -- against the fp32 CPU oracle, plus size-independent properties (net force ~ 0,
    translation invariance).  Random-init weights keep every MLP near-linear, where bf16x3 also meets north_star's tolerances
    (tests/test_gpu_parity.py has the saturated case)."""
    K = _K()
    o, energy_scale = config3_oracle
    model = _default_model()
    model.engine.set_precision(precision)
    g = fcc_cu_graph(10, 10, 25).to(DEV)
    assert g[K.NUM_NODES] == 10_000 and g[K.NUM_EDGES] == 420_000 and g[K.NUM_TRIPLETS] == 3_060_000
    out = model(g)
    # Per-atom energies are compared with the reference arithmetic (fp32 torch on the CPU).  The TOTAL is
    # compared with the fp64 sum of those: the reference path adds 10,000 near-identical fp32 numbers
    # sequentially (scatter_sum), which is itself 3.8e-5 away from the exact sum (fp64 oracle: -317.028015,
    # fp32 CPU path: -317.0400, engine: -317.02808).
    assert rel_err(out[K.SCALED_ATOMIC_ENERGIES], o["scaled_atomic_energies"]) < 1e-5
    e_exact = float(o["scaled_atomic_energies"].double().sum()) * energy_scale
    assert abs(float(out[K.TOTAL_ENERGY][0]) - e_exact) < 1e-5 * abs(e_exact)
    assert rel_err(out[K.FORCES], o["forces"]) < 1e-4
    assert rel_err(out[K.MID_EDGE_FEATURES][0], o["mid_edge_features_0"]) < 1e-4
    f = out[K.FORCES].double()
    assert float(f.sum(0).abs().max()) < 1e-3 * float(f.abs().max())
    e0, f0 = out[K.TOTAL_ENERGY].clone(), out[K.FORCES].clone()
    shifted = g.clone()
    shifted[K.POS] = shifted[K.POS] + torch.tensor([0.37, -1.1, 2.3], device=DEV)
    out2 = model(shifted)
    assert rel_err(out2[K.TOTAL_ENERGY], e0) < 1e-5
    assert rel_err(out2[K.FORCES], f0) < 1e-3


@pytest.mark.parametrize("mode", ["ref", "doc"])
def test_config3_10k_atom_cu_supercell_vs_the_reference_itself(mode, request):
    """BASELINE config 3 pinned to the REFERENCE's own outputs at the size the headline is timed at: tests/golden/big_cu10k_{ref,doc}.npz
    hold what the reference's nn code (tests/golden/generate_golden.py --cases cu10k, run in the build container) returns for
    this cell -- per-atom energies, total, forces, and 64 sampled rows of block 0's three-body aggregate and of the final edge
    features with their (centre, neighbour, shift), which tie the regenerated graph to the one the reference saw.
    Per-atom energies 1e-5, forces 1e-4 of max|F| (`doc`: plus the reference's own Legendre-backward defect, measured against
    the exact-derivative oracle run above, as in tests/test_gpu_parity.py).  The TOTAL: the reference adds 10,000 fp32 numbers
    sequentially (scatter_sum, nn/readout.py:49-53) and lands 3.8e-5 from the exact sum of its own per-atom energies; the engine's
    tree sum is compared with that exact (fp64) sum at 1e-5 and the distance to the reference's fp32 total is recorded."""
    import numpy as np
    from helpers import GOLDEN

    K = _K()
    z = np.load(GOLDEN / f"big_cu10k_{mode}.npz")
    model = _default_model()
    for m in model.model:
        if type(m).__name__ == "ThreeBodyInteration":
            m.nsb.factors = torch.tensor(z["const_factors"])
    g = fcc_cu_graph(10, 10, 25).to(DEV)
    out = model(g)
    idx = torch.tensor(z["sample_edges"], device=DEV)
    assert torch.equal(out[K.EDGE_INDEX][0][idx].cpu(), torch.tensor(z["sample_src"]))
    assert torch.equal(out[K.EDGE_INDEX][1][idx].cpu(), torch.tensor(z["sample_dst"]))
    assert torch.equal(out[K.EDGE_CELL_SHIFT][idx].cpu().to(torch.int64), torch.tensor(z["sample_shift"]).to(torch.int64))
    ea_ref = torch.tensor(z["out_scaled_atomic_energies"])
    assert rel_err(out[K.SCALED_ATOMIC_ENERGIES], ea_ref) < 1e-5
    assert rel_err(out[K.EDGE_ATTR][idx], torch.tensor(z["sample_edge_attr"])) < 1e-5
    assert rel_err(out[K.MID_EDGE_FEATURES][0][idx], torch.tensor(z["sample_mid_edge_features_0"])) < 1e-4
    f_ref = torch.tensor(z["out_forces"])
    defect = 0.0
    if mode == "doc":   # the reference's autograd forces carry its Legendre-backward defect here: measured against the exact derivative
        o, _ = request.getfixturevalue("config3_oracle")
        defect = rel_err(f_ref, o["forces"])
        assert defect < 5e-3
    f_err = rel_err(out[K.FORCES], f_ref)
    assert f_err < 1e-4 + defect
    # the reference's own backward of P_l (engine option "legendre_backward"): its forces without the defect allowance
    f_exact = out[K.FORCES].clone()
    model.engine.set_option("legendre_backward", 1)
    f_ref_mode = model(g, extras=False)[K.FORCES].clone()
    model.engine.set_option("legendre_backward", 0)
    f_err_ref_mode = rel_err(f_ref_mode, f_ref)
    assert f_err_ref_mode < 2e-5
    assert mode == "ref" or f_err_ref_mode < 0.5 * f_err
    assert torch.equal(model(g, extras=False)[K.FORCES], f_exact)
    e_sum64, e_ref32 = float(z["sum64_scaled_atomic_energies"]), float(z["out_total_energy"][0])
    e_eng = float(out[K.TOTAL_ENERGY][0])
    assert abs(e_eng - e_sum64) < 1e-5 * abs(e_sum64)
    assert abs(e_ref32 - e_sum64) < 1e-4 * abs(e_sum64)   # the reference's own sequential fp32 sum: ~3.8e-5 from its exact sum
    import os
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/config3_vs_reference.txt", "a") as fh:
            fh.write(f"cu10k_{mode}: per-atom E {rel_err(out[K.SCALED_ATOMIC_ENERGIES], ea_ref):.2e}  F {f_err:.2e} of max|F| = {float(f_ref.abs().max()):.3e} "
                     f"(reference's own defect vs the exact derivative: {defect:.2e}; with legendre_backward=1: F {f_err_ref_mode:.2e})  total: engine {e_eng:.6f}, fp64 sum of the reference's "
                     f"per-atom energies {e_sum64:.6f}, reference's fp32 scatter_sum {e_ref32:.6f} ({abs(e_ref32 - e_sum64) / abs(e_sum64):.2e} off its own exact sum)\n")


def test_config2_batched_random_species_cells():
    """Batched 64-atom random-species cells (BASELINE config 2): all 256 cells in one batch (graph built on the GPU);
    the first 32 against the CPU oracle, then the 256 batched == the same cells evaluated in two halves, and no net
    force on any cell."""
    from oracle import m3gnet_oracle as orc
    from helpers import random_cell_arrays
    from torch_m3gnet.data.graph_gpu import batch_from_arrays

    K = _K()
    model = _default_model()
    cells = [random_cell_arrays(64, 9.1, s) for s in range(256)]
    out = model(batch_from_arrays(*zip(*cells), 5.0, 4.0, device=DEV), extras=False)
    assert out[K.NUM_NODES] == 16_384 and out[K.TOTAL_ENERGY].shape == (256,)
    e_all, f_all = out[K.TOTAL_ENERGY].clone(), out[K.FORCES].clone()
    sub = model(batch_from_arrays(*zip(*cells[:32]), 5.0, 4.0, device=DEV))
    torch.set_num_threads(8)
    p, cfg, c, og = _oracle_inputs(model, sub)
    o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
    assert float(((e_all[:32].cpu() - o["total_energy"]).abs() / o["total_energy"].abs()).max()) < 1e-5
    assert rel_err(f_all[: 32 * 64], o["forces"]) < 1e-4
    halves = []
    for a, b in ((0, 128), (128, 256)):
        h = model(batch_from_arrays(*zip(*cells[a:b]), 5.0, 4.0, device=DEV), extras=False)
        halves.append((h[K.TOTAL_ENERGY].clone(), h[K.FORCES].clone()))
    torch.testing.assert_close(e_all, torch.cat([h[0] for h in halves]), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(f_all, torch.cat([h[1] for h in halves]), rtol=1e-5, atol=1e-7)
    f = f_all.double().view(256, 64, 3)
    assert float(f.sum(1).abs().max()) < 1e-3 * float(f.abs().max())


def test_config5_high_triplet_density():
    """r_cut = 6 A with 3-body cutoff 6 A (~60 neighbours, thousands of triplets per atom), reduced to 250
    atoms for the oracle comparison."""
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data.material_graph import Batch

    K = _K()
    model = _default_model(cutoff=6.0, threebody_cutoff=6.0)
    cell = random_cell_graph(250, 15.55, 0, cutoff=6.0, tb_cutoff=6.0)
    out = model(Batch.from_data_list([cell]).to(DEV))
    assert out[K.NUM_TRIPLETS] / out[K.NUM_NODES] > 2000
    torch.set_num_threads(8)
    p, cfg, c, og = _oracle_inputs(model, out)
    o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
    assert rel_err(out[K.TOTAL_ENERGY], o["total_energy"]) < 1e-5
    assert rel_err(out[K.FORCES], o["forces"]) < 1e-4
    assert rel_err(out[K.MID_EDGE_FEATURES][2], o["mid_edge_features_2"]) < 1e-4


def test_config5_full_size_through_properties():
    """BASELINE config 5 at full size: 2,000 atoms, r_cut = r_3 = 6 A (~60 neighbours, ~3,500 triplets per atom, 7.0 M
    triplets: the long-partner-list instantiation of the three-body kernels and their global-memory fallback are active).
    Too large for the CPU oracle in test time, so checked through size-independent properties: no net force, translation
    invariance, and the cell batched with a second structure == the cell on its own."""
    from helpers import random_cell_arrays
    from torch_m3gnet.data.graph_gpu import batch_from_arrays

    K = _K()
    model = _default_model(cutoff=6.0, threebody_cutoff=6.0)
    big = random_cell_arrays(2000, 31.1, seed=0)
    small = random_cell_arrays(40, 8.5, seed=1)
    g = batch_from_arrays(*zip(big), 6.0, 6.0, device=DEV)
    assert g[K.NUM_NODES] == 2000 and g[K.NUM_TRIPLETS] > 3000 * 2000
    out = model(g, extras=False)
    e1, f1 = out[K.TOTAL_ENERGY].clone(), out[K.FORCES].clone()
    assert torch.isfinite(e1).all() and torch.isfinite(f1).all()
    assert float(f1.double().sum(0).abs().max()) < 1e-3 * float(f1.abs().max())
    moved = (big[0], big[1] + np.array([0.7, -2.3, 1.9]), big[2])
    out2 = model(batch_from_arrays(*zip(moved), 6.0, 6.0, device=DEV), extras=False)
    assert rel_err(out2[K.TOTAL_ENERGY], e1) < 1e-5
    assert rel_err(out2[K.FORCES], f1) < 1e-3
    both = model(batch_from_arrays(*zip(big, small), 6.0, 6.0, device=DEV), extras=False)
    torch.testing.assert_close(both[K.TOTAL_ENERGY][:1], e1, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(both[K.FORCES][:2000], f1, rtol=1e-5, atol=1e-7)
    alone = model(batch_from_arrays(*zip(small), 6.0, 6.0, device=DEV), extras=False)
    torch.testing.assert_close(both[K.TOTAL_ENERGY][1:], alone[K.TOTAL_ENERGY], rtol=1e-6, atol=1e-6)
    # the two precision modes share nothing in their dense chains (exact-fp32 MFMAs / 3 split bf16 MFMAs): at ~60 neighbours
    # per atom, where per-centre sums are longest, they must still agree within the bf16x3 mode's error budget
    current = model.engine.precision
    other = "bf16x3" if current == "fp32" else "fp32"
    model.engine.set_precision(other)
    try:
        out3 = model(g, extras=False)
        assert rel_err(out3[K.TOTAL_ENERGY], e1) < 1e-5, (other, rel_err(out3[K.TOTAL_ENERGY], e1))
        assert rel_err(out3[K.FORCES], f1) < 1e-4, (other, rel_err(out3[K.FORCES], f1))
    finally:
        model.engine.set_precision(current)


# ------------------------------------------------------------------ hyper-parameter sweep (padding paths, template variants)
@pytest.mark.parametrize("l_max,n_max,dim,blocks,cut,tb_cut", [
    (1, 1, 8, 1, 4.0, 3.0),     # smallest everything: C = 1, one three-body k-step
    (2, 2, 33, 4, 4.5, 4.5),    # odd width (zero-padded to 64), 4 blocks, 3-body cutoff == cutoff
    (4, 4, 64, 2, 5.0, 3.5),    # largest supported bases: C = 16
    (3, 4, 48, 1, 4.2, 4.0),    # n_max = 4, C = 12
    (4, 1, 16, 3, 5.0, 4.0),    # C = 4
    # beyond the MFMA kernels' tiles: the any-size path (csrc/m3g_generic.hip) takes over
    (3, 3, 96, 2, 4.5, 4.0),    # embedding_dim 96
    (3, 3, 128, 1, 4.5, 4.0),   # embedding_dim 128
    (5, 5, 32, 2, 4.5, 4.0),    # l_max = n_max = 5, C = 25
    (9, 10, 24, 1, 4.0, 3.5),   # the reference's Bessel-root table used to its limits: C = 90
    (2, 3, 20, 10, 4.0, 3.5),   # more than 8 blocks
])
def test_hyperparameter_sweep_against_oracle(l_max, n_max, dim, blocks, cut, tb_cut):
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.data.material_graph import Batch
    from torch_m3gnet.model.build import build_model

    K = _K()
    torch.manual_seed(11)
    model = build_model(cut, tb_cut, l_max, n_max, 60, dim, blocks, elemental_energies=torch.linspace(-1, 1, 60), energy_scale=1.7)
    for m in model.model:
        if type(m).__name__ == "ThreeBodyInteration":
            m.nsb.factors = m.nsb.documented_factors()
    cells = [random_cell_graph(14 + 3 * s, 6.0 + 0.3 * s, 20 + s, cutoff=cut, tb_cutoff=tb_cut, zmax=59) for s in range(3)]
    big = l_max > 4 or n_max > 4 or dim > 64 or blocks > 8
    for kern in ((2,) if big else (1, 0, 2)):   # 1: MFMA kernels, 0: vector-ALU baseline, 2: any-size path
        model.engine.set_option("edge_kernel", kern)
        out = model(Batch.from_data_list([c.clone() for c in cells]).to(DEV))
        p, cfg, c, og = _oracle_inputs(model, out)
        p = {k: v.double() for k, v in p.items()}
        c = orc.make_constants(cfg, model.model[1].elemental_energies.cpu(), dtype=torch.float64)
        c.factors = model.model[6].nsb.factors.double()
        o = orc.energy_forces(p, cfg, c, og, legendre_backward="exact")
        assert float(((out[K.TOTAL_ENERGY].cpu().double() - o["total_energy"]).abs() / o["total_energy"].abs()).max()) < 1e-5, kern
        assert rel_err(out[K.FORCES], o["forces"]) < 1e-4, kern
        assert rel_err(out[K.STRESSES], o["stresses"]) < 1e-4, kern
        assert rel_err(out[K.EDGE_ATTR], o["edge_attr"]) < 1e-5, kern
        for b in range(blocks):
            assert rel_err(out[K.MID_EDGE_FEATURES][b], o[f"mid_edge_features_{b}"]) < 1e-4, (kern, b)


# ---- PBC-consistent virial -------------------------------------------------------------
def _strained_energy(params, cfg, consts, graph, eps):
    """fp64 oracle energy with positions and lattice strained by (1 + eps), topology and cell shifts fixed."""
    g = dict(graph)
    dfm = torch.eye(3, dtype=torch.float64) + eps
    g["pos"] = graph["pos"].double() @ dfm
    g["lattice"] = graph["lattice"].double() @ dfm
    return orc.energy_forces(params, cfg, consts, g, want_forces=False)["total_energy"].double()


@pytest.mark.parametrize("case", ["cu32", "mix"])
def test_pair_virial_is_the_strain_derivative(case):
    """sigma_ab V = -dE/d eps_ab, checked against central differences of the fp64 oracle energy (doc mode, so the
    three-body term contributes).  Tolerance 2e-3 of max|sigma V|: fp32 engine, h = 1e-4 differences."""
    from torch_m3gnet.nn import Gradient

    params, cfg, consts, graph, _ = load_oracle_case(case, "doc", dtype=torch.float64)
    model, _ = build_engine_model(case, "doc")
    model = Gradient(model.model, pair_virial=True).cuda()
    out = model(engine_graph(graph))
    lat = graph["lattice"].double().reshape(-1, 3, 3)
    vol = torch.linalg.det(lat).abs()
    sv = out["stresses"].double().cpu() * vol[:, None]
    voigt = [(0, 0), (1, 1), (2, 2), (1, 2), (2, 0), (0, 1)]
    h = 1e-4
    fd = torch.zeros_like(sv)
    for k, (a, b) in enumerate(voigt):
        eps = torch.zeros(3, 3, dtype=torch.float64)
        eps[a, b] += 0.5 * h
        eps[b, a] += 0.5 * h
        fd[:, k] = -(_strained_energy(params, cfg, consts, graph, eps) - _strained_energy(params, cfg, consts, graph, -eps)) / (2 * h)
    assert float((sv - fd).abs().max() / fd.abs().max()) < 2e-3, (sv, fd)


def test_pair_virial_equals_reference_formula_without_boundary_crossings():
    """A cluster in a large box (no edge crosses the cell): both formulas are the same sum."""
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph
    from torch_m3gnet.nn import Gradient

    model, cfg = build_engine_model("cu32", "doc")
    rng = np.random.default_rng(3)
    pos = 20.0 + rng.uniform(-3.0, 3.0, (24, 3))
    g = Batch.from_data_list([MaterialGraph.from_arrays(np.eye(3) * 40.0, pos, np.full(24, 29), cfg.cutoff, cfg.threebody_cutoff)]).to("cuda")
    assert int(g["edge_cell_shift"].abs().sum()) == 0
    ref = model.cuda()(g)["stresses"]
    pair = Gradient(model.model, pair_virial=True).cuda()(g)["stresses"]
    # sum pos (x) F cancels to ~1e-6 of its terms (|pos| ~ 20): compare relative to the largest component
    assert float((ref - pair).abs().max() / pair.abs().max()) < 5e-4


def test_pair_virial_is_invariant_under_lattice_translation_of_an_atom():
    """Moving one atom by a lattice vector (with its cell shifts adjusted) changes the reference formula but not the
    pair virial -- the reason row 4 exists (the reference's own stress test is skipped, tests/test_model.py:123)."""
    from torch_m3gnet.data.material_graph import Batch, MaterialGraph
    from torch_m3gnet.nn import Gradient

    model, cfg = build_engine_model("cu32", "doc")
    lat = np.eye(3) * 7.22
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    grid = np.stack(np.meshgrid(*[np.arange(2)] * 3, indexing="ij"), -1).reshape(-1, 1, 3)
    pos = (grid + base[None]).reshape(-1, 3) * 3.61 + np.random.default_rng(0).uniform(-0.05, 0.05, (32, 3))
    pos2 = pos.copy()
    pos2[5] += lat[0] - 2 * lat[2]
    z = np.full(32, 29)
    pv = Gradient(model.model, pair_virial=True).cuda()
    outs = []
    for p in (pos, pos2):
        g = Batch.from_data_list([MaterialGraph.from_arrays(lat, p, z, cfg.cutoff, cfg.threebody_cutoff)]).to("cuda")
        outs.append((pv(g)["stresses"].cpu(), model.cuda()(g)["stresses"].cpu()))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-4, atol=1e-6 * float(outs[0][0].abs().max()) + 1e-9)
    assert float((outs[0][1] - outs[1][1]).abs().max()) > 10 * float((outs[0][0] - outs[1][0]).abs().max())


@pytest.mark.parametrize("edge_kernel,split_node_tiles", [(1, 128), (1, 0), (0, 128), (2, 128)])
def test_out_of_range_species_past_the_host_check_give_nan_and_a_sticky_bit(edge_kernel, split_node_tiles):
    """The library side of the species check (a C caller has no Python host in front of it; reference: IndexError at
    elemental_energies[atom_types], nn/atom_ref.py:25-29): with the host's own check switched off, every engine -- split / persistent
    MFMA kernels, vector-ALU baseline, any-size path -- leaves NaN as the offending structure's energy and the sticky
    M3G_TOPO_ERR_SPECIES bit on the topology, never indexing a table with the value (memory-safe by construction: clamped)."""
    from torch_m3gnet.data.material_graph import Batch
    from torch_m3gnet.nn.modules import _Topology

    K = _K()
    model = _default_model()
    model.engine.set_option("edge_kernel", edge_kernel)
    model.engine.set_option("split_node_tiles", split_node_tiles)
    model.engine._check_species = lambda graph, probe: None
    cells = [random_cell_graph(12, 6.0, s) for s in range(3)]
    g = Batch.from_data_list(cells).to(DEV)
    good = model(g.clone())[K.TOTAL_ENERGY].clone()
    assert torch.isfinite(good).all()
    for bad in (95, -3, 1 << 40):
        h = g.clone()
        types = h[K.ATOM_TYPES].clone()
        types[14] = bad              # an atom of the second structure
        h[K.ATOM_TYPES] = types
        out = model(h)
        e = out[K.TOTAL_ENERGY]
        # (the other structures are untouched; the vector-ALU baseline path adds its per-centre sums with float atomics: not bitwise)
        assert torch.isnan(e[1]) and torch.allclose(e[[0, 2]], good[[0, 2]], rtol=1e-5, atol=0), (bad, e, good)
        assert _Topology.of(out).status() & 4
