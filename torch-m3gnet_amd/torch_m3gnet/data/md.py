"""Device-resident graph of a batch of periodic structures along a trajectory (MD, relaxation): positions stay on the GPU, the
neighbour / triplet lists, the CSR topology and its certificate are REUSED while the lists are provably unchanged.

The reference rebuilds the whole graph for every structure it sees, on the host (data/material_graph.py:133-254: pymatgen
neighbour search + an O(T) Python triplet loop).  `VerletGraph.update(pos)` returns, for the positions given, exactly the
`MaterialGraph` batch a fresh build (`graph_gpu.batch_from_arrays`, or the reference's own construction in canonical order)
would return -- same edges, same order, same shifts, same triplets, hence bit-identical energies and forces -- at the cost of
one small pass over a skin list (C ABI: m3g_verlet_*, csrc/m3g_graph_build.hip):

  reuse    no candidate pair crossed the cutoff or the three-body cutoff and no atom moved further than skin / 2 since the
           candidates were searched: the index tensors, the engine's topology (cached on the graph) and its hints word stay as
           they are, only `pos` is new (written in place: the storage of every tensor of the graph is unchanged);
  refill   some pair crossed a cutoff: the lists are re-derived from the candidates (no search), the engine rebuilds its topology;
  search   an atom moved further than skin / 2 (or the caller asks): a new candidate search with cutoff + skin.

`update(pos)` waits for the skin test's verdict (32 bytes) before it returns.  `evaluate(model, pos)` need not: it queues the
test, writes the positions into the current graph, queues the evaluation behind it ON THE ASSUMPTION that nothing changed, and
reads the verdict afterwards -- two small kernels and no wait; when the verdict says otherwise the lists are rebuilt and the
evaluation runs again (its first results are overwritten).  A wrong guess costs a whole step, so `evaluate` guesses only after
`speculate_after` (4) consecutive "unchanged" verdicts and is `model(update(pos))` until then: a 10,000-atom cell at finite
temperature refills on nearly every step and never pays for a guess, a small or cold cell runs without the wait.

Positions are the trajectory's own, UNWRAPPED coordinates (an atom that crosses a cell face keeps going; wrapping it back by a
lattice vector reads as a jump and takes the search path, which is correct, just slower).
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np
import torch

from .. import _cuda, _lib
from . import MaterialGraphKey as K
from .graph_gpu import _ptr, _stream, neighbor_list_gpu
from .material_graph import Batch


class VerletGraph:
    def __init__(self, lattices: Sequence, atomic_numbers: Sequence, cutoff: float, threebody_cutoff: float, skin: float = 0.5,
                 device="cuda"):
        """lattices: list of [3,3] arrays (rows = lattice vectors), atomic_numbers: list of [n_s] arrays (one per structure);
        cutoff / threebody_cutoff as `MaterialGraph.from_structure`; skin in the same length unit."""
        if threebody_cutoff > cutoff:
            raise ValueError("Three body cutoff raidus should be smaller than two body.")
        self.cutoff, self.threebody_cutoff, self.skin = float(cutoff), float(threebody_cutoff), float(skin)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _lib.load_library()
        lat = np.stack([np.asarray(L, dtype=np.float64).reshape(3, 3) for L in lattices])
        sizes = [len(np.asarray(z).reshape(-1)) for z in atomic_numbers]
        z = np.concatenate([np.asarray(a).reshape(-1) for a in atomic_numbers])
        self._host_lattice = lat
        self.lattice = torch.tensor(lat, device=self.device)                                    # fp64, as the search works
        self.batch = torch.tensor(np.repeat(np.arange(len(sizes)), sizes), dtype=torch.int64, device=self.device)
        self.atom_types = torch.tensor(z - 1, dtype=torch.long, device=self.device)
        self._species_range = (int(z.min()) - 1, int(z.max()) - 1) if len(z) else (0, 0)
        self.lattice32 = self.lattice.to(torch.float)
        self.N, self.S = int(len(z)), len(sizes)
        self.graph = None
        self._cand = None            # (edge_index [2,Ec], shift [Ec,3], row_ptr [N+2], state [Ec] u8, pos_ref [N,3] f64, scratch)
        self.stats = {"reuse": 0, "refill": 0, "search": 0}
        self._verdict = torch.zeros(8, dtype=torch.int64).pin_memory()   # m3g_verlet_update_async: max disp^2 (bits), changed, E, T, -, longest candidate row
        self._verdict_i64 = self._verdict.numpy()                # the same pinned words as numpy views: reading six of them through
        self._verdict_f64 = self._verdict_i64.view(np.float64)   # tensor indexing cost ~15 us of host time per step
        self._verdict_ptr = C.c_void_p(self._verdict.data_ptr())
        self._verdict_ready = torch.cuda.Event()
        self._topo_verdict = torch.zeros(16, dtype=torch.int32).pin_memory()   # m3g_topology_build_canonical_begin / _end
        self._pending = None         # positions of a begin() whose verdict has not been read
        self._state_valid = False    # the candidates' membership bytes describe the current lists
        self._max_row = 0
        self._md = None              # m3g_md handle of `step`
        self._md_for = None          # the candidate set its buffers were made for
        self._md_buffers = None
        self._lists_owner = "py"     # which path wrote cand_state / the lists last: update / evaluate ("py") or step ("c")
        self._cand_dist = None
        self._test_args = None
        self._graph_constants = {K.ATOM_TYPES: self.atom_types, K.LATTICE: self.lattice32, K.BATCH: self.batch, K.NUM_NODES: self.N,
                                 "num_graphs": self.S}
        self._reuse_streak = 0       # consecutive verdicts "lists unchanged" (evaluate's policy)
        self.speculate_after = 4     # evaluate() queues the step ahead of the verdict after this many of them (0: always)
        self.eager_topology = True   # queue the topology build with the fill (False: the engine builds it at its call)
        self.split_fill = False      # True: refill through m3g_verlet_fill + m3g_threebody_build (tests; identical lists)

    def set_lattice(self, lattices: Sequence) -> None:
        """New cell(s) (variable-cell relaxation, NPT): the candidates were searched in the old cell, so the next `update` /
        `evaluate` searches again.  Positions are the caller's to rescale."""
        lat = np.stack([np.asarray(L, dtype=np.float64).reshape(3, 3) for L in lattices])
        if lat.shape != self._host_lattice.shape:
            raise ValueError(f"expected {self._host_lattice.shape[0]} lattices of shape [3, 3]")
        self._host_lattice = lat
        self.lattice = torch.tensor(lat, device=self.device)
        self.lattice32 = self.lattice.to(torch.float)
        self._graph_constants[K.LATTICE] = self.lattice32
        self._cand, self.graph, self._pending, self._state_valid = None, None, None, False

    # ------------------------------------------------------------------------------------------------ candidates
    def _search(self, pos: torch.Tensor) -> None:
        ei, shift, dist = neighbor_list_gpu(self.lattice, pos, self.batch, self.cutoff + self.skin, host_lattice=self._host_lattice)
        self._cand_dist = dist   # (for the triplet capacity of `step`)
        ec = int(ei.size(1))
        rows = torch.empty(self.N + 2, dtype=torch.int32, device=self.device)
        with _cuda.on_device(self.device):
            _lib.check(self.lib.m3g_verlet_rows(self.N, ec, _ptr(ei), _ptr(rows), _stream()))
        nbytes = C.c_size_t()
        _lib.check(self.lib.m3g_verlet_scratch_bytes(self.N, ec, C.byref(nbytes)))
        scratch = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
        state = torch.empty(ec + 16, dtype=torch.uint8, device=self.device)   # written by the first fill
        self._cand = (ei, shift, rows, state, pos.clone(), scratch)
        self._max_row = 1 << 30   # the longest candidate row arrives with the next verdict (m3g_verlet_update_async, word 5)
        self._state_valid = False
        self.stats["search"] += 1

    def _own_lists(self) -> None:
        """update / evaluate after `step` calls: the membership bytes describe the lists `step` keeps on the C side, not `self.graph`."""
        if self._lists_owner != "py":
            self._lists_owner = "py"
            self.graph, self._pending, self._state_valid = None, None, False
            self._reuse_streak = 0   # (evaluate's speculation counted verdicts about lists that are no longer the current ones)

    def _queue_test(self, pos: torch.Tensor) -> None:
        """The skin test at `pos`, queued on the current stream together with the copy of its verdict to pinned host memory."""
        self._own_lists()
        a = self._test_args
        if a is None or a[0] is not self._cand:   # the arguments that change only with a new search, converted once
            ei, shift, rows, state, pos_ref, scratch = self._cand
            a = self._test_args = (self._cand, int(ei.size(1)), _ptr(pos_ref), _ptr(self.lattice), _ptr(self.batch), _ptr(ei), _ptr(shift),
                                   _ptr(rows), _ptr(state), _ptr(scratch), scratch.numel())
        with _cuda.on_device(self.device):
            _lib.check(self.lib.m3g_verlet_update_async(self.N, self.S, a[1], _ptr(pos), a[2], a[3], a[4], a[5], a[6], a[7], self.cutoff,
                                                        self.threebody_cutoff, a[8] if self._state_valid else None, a[9], a[10],
                                                        self._verdict_ptr, _stream()))
            self._verdict_ready.record()

    def _read_verdict(self):
        self._verdict_ready.synchronize()
        v = self._verdict_i64
        disp = float(np.sqrt(self._verdict_f64[0]))
        self._max_row = int(v[5])   # decides whether the two-launch refill applies (m3g_verlet_fill_lists)
        return disp, bool(v[1]), int(v[2]), int(v[3])

    def _update(self, pos: torch.Tensor):
        self._queue_test(pos)
        return self._read_verdict()

    def _fill(self, pos: torch.Tensor, n_e: int, n_t: int) -> None:
        """New index tensors from the candidates at `pos` (m3g_verlet_update has just run on them) -> a NEW graph object (the
        engine's topology cache lives on the graph and is keyed by the tensors' storage)."""
        c_ei, c_shift, rows, state, _, scratch = self._cand
        dev, N = self.device, self.N
        ei = torch.empty(2, n_e, dtype=torch.int64, device=dev)
        shift = torch.empty(n_e, 3, dtype=torch.int32, device=dev)
        tei = torch.empty(2, n_t, dtype=torch.int64, device=dev)
        nti = torch.empty(N, dtype=torch.int64, device=dev)
        ntij = torch.empty(n_e, dtype=torch.int32, device=dev)
        with _cuda.on_device(dev):
            if self._max_row <= _lib.VERLET_FILL_LISTS_MAX_ROW and N <= 262144 and not self.split_fill:
                # edges, shifts, membership bytes, triplets and triplet counts in two launches
                _lib.check(self.lib.m3g_verlet_fill_lists(N, int(c_ei.size(1)), n_e, n_t, self._max_row, _ptr(scratch), _ptr(c_ei), _ptr(c_shift),
                                                          _ptr(rows), _ptr(ei), _ptr(shift), _ptr(state), _ptr(tei), _ptr(nti), _ptr(ntij), _stream()))
                self._state_valid = True
            else:   # very long candidate rows / very many atoms: the general calls (identical lists)
                dist = torch.empty(n_e, dtype=torch.float64, device=dev)
                _lib.check(self.lib.m3g_verlet_fill(N, int(c_ei.size(1)), n_e, _ptr(scratch), _ptr(c_ei), _ptr(c_shift), _ptr(rows), _ptr(ei),
                                                    _ptr(shift), _ptr(dist), _ptr(state), _stream()))
                self._state_valid = True
                d32 = dist.to(torch.float32)
                tb_bytes = C.c_size_t()
                _lib.check(self.lib.m3g_threebody_scratch_bytes(N, n_e, C.byref(tb_bytes)))
                tb_scratch = torch.empty(tb_bytes.value, dtype=torch.uint8, device=dev)
                _lib.check(self.lib.m3g_threebody_build(N, n_e, _ptr(ei), _ptr(d32), float(self.threebody_cutoff), _ptr(tb_scratch),
                                                        tb_bytes.value, n_t, _ptr(tei), _ptr(nti), _ptr(ntij), _stream()))
        g = Batch.__new__(Batch)
        dict.__init__(g, self._graph_constants)   # atom_types, lattice, batch, counts of atoms / structures: the trajectory's own
        dict.update(g, {K.POS: pos.to(torch.float), K.NUM_TRIPLET_I: nti, K.EDGE_INDEX: ei, K.EDGE_CELL_SHIFT: shift, K.NUM_TRIPLET_IJ: ntij,
                        K.TRIPLET_EDGE_INDEX: tei, K.NUM_EDGES: n_e, K.NUM_TRIPLETS: n_t})
        # the species check of the engine is a property of `atom_types`, which every graph of this trajectory shares
        if self.graph is not None and "_m3g_species_ok" in self.graph:
            dict.__setitem__(g, "_m3g_species_ok", self.graph["_m3g_species_ok"])
        from ..nn.modules import _Topology

        # lists of this library's own builder: the topology build skips the checks they pass by construction (graph_gpu.mark_canonical;
        # the signature is formed once here for the mark, the cache key and the build)
        sig = _Topology.signature(g)
        dict.__setitem__(g, "_m3g_canonical_lists", sig)
        # the topology build is queued right behind the fill (and waited for by the engine when it needs the buffer): the device
        # builds while the host walks from here to its m3g_energy_forces call
        if self.graph is not None and self.graph.get("_m3g_topology") is not None:
            self.graph["_m3g_topology"][1].finish()   # one verdict buffer: a build still in flight (a graph that was never evaluated) ends first
        dict.__setitem__(g, "_m3g_pinned_verdict", self._topo_verdict)
        if self.eager_topology:
            dict.__setitem__(g, "_m3g_topology", (sig, _Topology(g, sig)))
        self.graph = g

    # ------------------------------------------------------------------------------------------------ the per-step call
    def update(self, pos: torch.Tensor, force: str | None = None) -> Batch:
        """The graph at positions `pos` ([N,3] device tensor, fp64 or fp32, unwrapped).  `force="search"` / `"refill"` take that
        path whatever the skin test says (benchmarks, tests).  The returned object is `self.graph`; on the reuse path it is the
        SAME object with the same tensors, its `pos` overwritten in place."""
        pos = self._check_pos(pos)
        self._pending = None
        if self._cand is None or force == "search" or self.skin <= 0.0:
            self._search(pos)
        disp, changed, n_e, n_t = self._update(pos)
        return self._settle(pos, disp, changed, n_e, n_t, force)

    def _check_pos(self, pos: torch.Tensor) -> torch.Tensor:
        if pos.device != self.device or tuple(pos.shape) != (self.N, 3):
            raise ValueError(f"pos must be a [{self.N}, 3] tensor on {self.device}")
        return pos.detach().to(torch.float64).contiguous()

    def _settle(self, pos, disp, changed, n_e, n_t, force=None) -> Batch:
        if disp >= 0.5 * self.skin and not (disp == 0.0):
            self._search(pos)
            disp, changed, n_e, n_t = self._update(pos)
        if changed or self.graph is None or force == "refill":
            self._fill(pos, n_e, n_t)
            self.stats["refill"] += 1
            self._reuse_streak = 0
        else:
            self.graph[K.POS].copy_(pos)   # fp64 -> fp32 in place: every tensor of the graph keeps its storage
            self.stats["reuse"] += 1
            self._reuse_streak += 1
        return self.graph

    # ------------------------------------------------------------------------------------------------ without the wait
    def begin(self, pos: torch.Tensor) -> Batch:
        """Queue the skin test at `pos` and return the CURRENT graph with `pos` written into it, on the assumption that the lists
        are unchanged; `confirm()` says whether that held.  (No graph yet: falls back to `update`.)"""
        self._own_lists()   # (`step` wrote the lists last: `self.graph` is gone, take the update path)
        if self._cand is None or self.graph is None or self.skin <= 0.0:
            return self.update(pos)
        pos = self._check_pos(pos)
        self._queue_test(pos)
        self.graph[K.POS].copy_(pos)
        self._pending = pos
        return self.graph

    def confirm(self) -> bool:
        """True: the graph `begin` returned was the right one (whatever was evaluated on it stands).  False: the lists had to be
        rebuilt -- `self.graph` is the right graph now, evaluate again."""
        if self._pending is None:
            return True
        pos, self._pending = self._pending, None
        disp, changed, n_e, n_t = self._read_verdict()
        if not changed and (disp < 0.5 * self.skin or disp == 0.0):
            self.stats["reuse"] += 1
            self._reuse_streak += 1
            return True
        self._settle(pos, disp, changed, n_e, n_t)
        return False

    def evaluate(self, model, pos: torch.Tensor, **kwargs):
        """model(graph at `pos`), without waiting for the skin test first WHILE THAT PAYS (see the module text).  Queueing the step
        before the verdict saves the wait (~0.08 ms on the 10k-atom cell) when the lists turn out unchanged and costs a whole wasted
        step when they do not -- and on a large cell at finite temperature some pair crosses a cutoff on nearly every step.  So the
        step is queued ahead only after `speculate_after` consecutive verdicts of "unchanged" (small cells, cold or frozen
        systems); otherwise this is `model(self.update(pos))`.  Results are the same either way."""
        if self._reuse_streak < self.speculate_after:
            return model(self.update(pos), **kwargs)
        out = model(self.begin(pos), **kwargs)
        if not self.confirm():
            out = model(self.graph, **kwargs)
        return out

    # ------------------------------------------------------------------------------------------------ one call per step
    def __del__(self):
        try:
            if getattr(self, "_md", None):
                self.lib.m3g_md_destroy(self._md)
                self._md = None
        except Exception:
            pass

    def _md_prepare(self, engine, exact: bool = False) -> None:
        """Buffers of the candidates' capacity for `step`: list tensors, topology, workspace.  Made when there are none, when the
        candidates outgrow them, or (`exact`) when m3g_md_step found the triplets beyond their capacity; a new search otherwise keeps
        the buffers it has (their capacity is checked by the library at every refill) and costs no wait for the device here."""
        ei, shift, rows, state, pos_ref, scratch = self._cand
        dev, N, S = self.device, self.N, self.S
        ec = int(ei.size(1))
        b = self._md_buffers
        keep = b is not None and b["plan_key"][0] == engine.precision and ec <= b["cap_e"] and not exact
        cap_t = b["cap_t"] if keep else 0
        if not keep:
            # every configuration within skin / 2 of the reference positions has its three-body edges among the candidates within
            # threebody_cutoff + skin: sum_i c_i (c_i - 1) bounds its triplets
            near = self._cand_dist <= self.threebody_cutoff + self.skin
            c = torch.bincount(ei[0][near], minlength=N)
            cap_t = int((c * (c - 1)).sum())
        if not keep or cap_t > b["cap_t"]:
            # new buffers, with some headroom so that the searches of a trajectory (whose candidate counts drift by a few per cent)
            # keep them
            cap_e, cap_t = max(int(ec * 1.1), 1), max(int(cap_t * 1.1), 1)
            nb_topo, nb_work = C.c_size_t(), C.c_size_t()
            _lib.check(self.lib.m3g_topology_bytes(N, cap_e, cap_t, S, C.byref(nb_topo)))
            _lib.check(self.lib.m3g_workspace_bytes(engine.plan, N, cap_e, cap_t, S, C.byref(nb_work)))
            b = {
                "ei": torch.empty(2 * cap_e, dtype=torch.int64, device=dev), "shift": torch.empty(3 * cap_e, dtype=torch.int32, device=dev),
                "tei": torch.empty(2 * cap_t, dtype=torch.int64, device=dev), "nti": torch.empty(N, dtype=torch.int64, device=dev),
                "ntij": torch.empty(cap_e, dtype=torch.int32, device=dev), "pos32": torch.empty(N, 3, dtype=torch.float, device=dev),
                "topo": torch.empty(nb_topo.value, dtype=torch.uint8, device=dev), "work": torch.empty(nb_work.value, dtype=torch.uint8, device=dev),
                "cap_e": cap_e, "cap_t": cap_t, "plan_key": (engine.precision, nb_work.value),
            }
        cap_e, cap_t = b["cap_e"], b["cap_t"]
        lists = _lib.M3GMdLists(
            n_atoms=N, n_structs=S, n_cand=ec, cap_edges=cap_e, cap_triplets=cap_t, cutoff=self.cutoff, threebody_cutoff=self.threebody_cutoff,
            skin=self.skin, pos_ref=pos_ref.data_ptr(), lattice=self.lattice.data_ptr(), lattice32=self.lattice32.data_ptr(),
            batch=self.batch.data_ptr(), atom_types=self.atom_types.data_ptr(), cand_edge_index=ei.data_ptr(), cand_shift=shift.data_ptr(),
            cand_row_ptr=rows.data_ptr(), cand_state=state.data_ptr(), verlet_scratch=scratch.data_ptr(), verlet_scratch_bytes=scratch.numel(),
            edge_index=b["ei"].data_ptr(), edge_cell_shift=b["shift"].data_ptr(), triplet_edge_index=b["tei"].data_ptr(),
            num_triplet_i=b["nti"].data_ptr(), num_triplet_ij=b["ntij"].data_ptr(), pos32=b["pos32"].data_ptr(), topo=b["topo"].data_ptr(),
            topo_bytes=b["topo"].numel(), workspace=b["work"].data_ptr(), workspace_bytes=b["work"].numel())
        if self._md is None:
            h = C.c_void_p()
            _lib.check(self.lib.m3g_md_create(C.byref(h)))
            self._md = h
        _lib.check(self.lib.m3g_md_set_lists(self._md, C.byref(lists)))
        self._md_buffers, self._md_for = b, self._cand
        self._lists_owner = "c"

    def step(self, model, pos: torch.Tensor, forces: bool = True, force: str | None = None) -> dict:
        """Energies, forces and stresses at `pos` through ONE call of the library per step (m3g_md_step): the skin test, the lists and
        topology when a pair crossed a cutoff, and the engine are sequenced on the C side, on buffers of the candidates' capacity
        made once per search -- a refill allocates nothing and the interpreter runs once per step.  `model`: the `Gradient` built by
        `build_model`.  Returns {total_energy [S], forces [N,3], stresses [S,6]} (new tensors per call), bit-identical to
        `model(self.update(pos))`.  The current lists are `step_lists()`.  `force="refill"` / `"search"` as in `update`.
        Mixing `step` with `update` / `evaluate` on one object is allowed (each re-derives the lists when the other wrote them last)."""
        pos = self._check_pos(pos)
        eng = model.engine
        dev = self.device
        lo, hi = self._species_range
        if lo < 0 or hi >= eng.cfg.num_types:   # (the reference fails on elemental_energies[atom_types], nn/atom_ref.py:27)
            raise IndexError(f"atom_types must lie in [0, {eng.cfg.num_types - 1}] (num_types = {eng.cfg.num_types}); got values in [{lo}, {hi}]")
        sig = eng._signature(dev)
        with _cuda.on_device(dev):
            if sig != eng._sig:
                eng.commit()
                eng._sig = sig
            self._pending = None
            if self._cand is None or force == "search" or self.skin <= 0.0:
                self._search(pos)
            f32 = dict(dtype=torch.float, device=dev)
            e = torch.empty(self.S, **f32)
            f = torch.empty(self.N, 3, **f32) if forces else None
            st = torch.empty(self.S, 6, **f32) if forces else None
            res = _lib.M3GMdResult()
            searched = exact_done = False
            while True:
                b = self._md_buffers
                if self._md_for is not self._cand or b is None or b["plan_key"][0] != eng.precision:
                    self._md_prepare(eng)
                    b = self._md_buffers
                elif self._lists_owner != "c":       # update / evaluate rewrote the membership bytes since
                    _lib.check(self.lib.m3g_md_invalidate(self._md))
                    self._lists_owner = "c"
                _lib.check(self.lib.m3g_md_step(self._md, eng.plan, _ptr(pos), _ptr(e), _ptr(f), _ptr(st), 1 if force == "refill" else 0,
                                                C.byref(res), _stream()))
                if res.path == _lib.MD_NEED_SEARCH and not searched:
                    self._search(pos)
                    searched = True
                    continue
                if res.path == _lib.MD_UNSUPPORTED and not exact_done:
                    self._md_prepare(eng, exact=True)   # buffers kept from an earlier search may be too small for this one
                    exact_done = True
                    continue
                break
            if res.path == _lib.MD_UNSUPPORTED or res.path == _lib.MD_NEED_SEARCH:
                # candidate rows beyond the two-launch refill, or lists beyond the capacity bound: the general path for this step
                out = model(self.update(pos), forces=forces, extras=False)
                return {k: out[k] for k in (K.TOTAL_ENERGY, K.FORCES, K.STRESSES) if k in out}
            self.stats["reuse" if res.path == _lib.MD_REUSE else "refill"] += 1
            self._step_sizes = (int(res.n_edges), int(res.n_triplets))
        out = {K.TOTAL_ENERGY: e}
        if forces:
            out[K.FORCES], out[K.STRESSES] = f, st
        return out

    def step_lists(self) -> dict:
        """The lists `step` ran on last, as views of its capacity buffers (valid until the next `step`)."""
        n_e, n_t = self._step_sizes
        b = self._md_buffers
        return {K.EDGE_INDEX: b["ei"][: 2 * n_e].view(2, n_e), K.EDGE_CELL_SHIFT: b["shift"][: 3 * n_e].view(n_e, 3),
                K.TRIPLET_EDGE_INDEX: b["tei"][: 2 * n_t].view(2, n_t), K.NUM_TRIPLET_I: b["nti"], K.NUM_TRIPLET_IJ: b["ntij"][:n_e]}
