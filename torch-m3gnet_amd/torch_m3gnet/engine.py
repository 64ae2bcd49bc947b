"""Host side of the fused path: turns the module tree built by `build_model` into an m3g_plan and
drives `m3g_energy_forces` (C ABI, include/m3gnet_hip.h).  PyTorch supplies device memory and the
stream only."""
from __future__ import annotations

import ctypes as C
import os
from itertools import chain
from operator import attrgetter

import numpy as np
import torch

from . import _cuda, _lib
from .data import MaterialGraphKey as K
from .nn import modules as M

_DATA_PTR = torch.Tensor.data_ptr
_VERSION = attrgetter("_version")


def _host_f32(t: torch.Tensor) -> np.ndarray:
    return np.ascontiguousarray(t.detach().to(device="cpu", dtype=torch.float32).numpy())


class Engine:
    """One plan per model; re-commits automatically when parameters or captured constants change."""

    def __init__(self, seq: torch.nn.Module):
        mods = list(seq)
        kinds = [type(m).__name__ for m in mods]
        head = ["ScaleLength", "AtomRef", "DistanceAndAngle", "AtomFeaturizer", "EdgeFeaturizer", "EdgeAdjustor"]
        n_blocks = (len(mods) - len(head) - 1) // 2
        expect = head + ["ThreeBodyInteration", "M3GNetConv"] * n_blocks + ["AtomWiseReadout"]
        if kinds != expect:
            raise RuntimeError(
                "Gradient(model): the fused MI355X path needs the module order of build_model "
                f"({' -> '.join(head)} -> [ThreeBodyInteration -> M3GNetConv]* -> AtomWiseReadout); got {kinds}"
            )
        if int(mods[-1].num_layers) != 3:
            raise ValueError("AtomWiseReadout: the MI355X engine implements the readout of build_model (num_layers = 3, "
                             f"model/build.py:69-76); got num_layers = {mods[-1].num_layers}")
        self.seq = seq
        self.mods = mods
        self.num_blocks = n_blocks
        self.scale, self.atom_ref, _, self.atom_feat, self.edge_feat, self.edge_adj = mods[:6]
        self.tb = [mods[6 + 2 * b] for b in range(n_blocks)]
        self.readout = mods[-1]
        tb0 = self.tb[0] if n_blocks else None
        ls = float(self.scale.length_scale)
        self.cfg = _lib.M3GConfig(
            cutoff=float(self.edge_feat.cutoff) * ls,
            threebody_cutoff=(float(tb0.threebody_cutoff) * ls) if tb0 is not None else float(self.edge_feat.cutoff) * ls,
            energy_scale=float(self.readout.scale), length_scale=ls,
            l_max=int(tb0.l_max) if tb0 is not None else 1, n_max=int(self.edge_feat.degree),
            num_types=int(self.atom_feat.num_types), embedding_dim=int(self.atom_feat.linear.out_features),
            num_blocks=n_blocks, reserved=0,
        )
        # build.py divides the cutoffs by length_scale before handing them to the modules; the plan takes
        # the unscaled values and repeats that division in double, so recover them exactly when possible
        self._scaled_cutoff = float(self.edge_feat.cutoff)
        self._scaled_tb_cutoff = float(tb0.threebody_cutoff) if tb0 is not None else self._scaled_cutoff
        self.lib = _lib.load_library()
        self.plan = C.c_void_p()
        _lib.check(self.lib.m3g_plan_create(C.byref(self.cfg), C.byref(self.plan)))
        self._sig = None
        # the per-module parameter dicts, collected once: walking `seq.parameters()` (named_modules + de-duplication) costs ~60 us per
        # call, which a caller that waits for the device every step (MD) pays in full; a Parameter that is REPLACED later
        # (`module.weight = Parameter(...)`) lands in the same dict and is seen
        # Every module's dict is kept (a Parameter registered LATER on a module that had none is then seen too), and the `_modules`
        # dicts are walked per call for their identities: a submodule replaced after construction rebuilds both lists.
        self._collect_param_dicts()
        self.precision = "fp32"   # the reference's arithmetic; the split modes are explicit opt-ins (set_precision / M3G_PRECISION)
        env_prec = os.environ.get("M3G_PRECISION")   # run a whole test suite in another mode without touching it
        if env_prec:
            self.set_precision(env_prec)
        self._workspace = None
        self._graph_replay = False
        # ask every new topology for its m3g_topology_hints word (complete triplet lists -> three-body moment kernels); False for
        # loops that rebuild the neighbour list every step on sparse cells, where the certificate costs more than it returns
        self.topology_hints = True
        self._species_host = None   # pinned [min, max] of the last species check
        self._out_cache = {}
        self._count_next, self.last_launch_count = False, None

    def __del__(self):
        try:
            if getattr(self, "plan", None) and self.plan.value:
                self.lib.m3g_plan_destroy(self.plan)
                self.plan = C.c_void_p()
        except Exception:
            pass

    # ---------------------------------------------------------------- parameters
    def _collect_param_dicts(self) -> None:
        mods = list(self.seq.modules())
        self._dicts = [m._modules for m in mods] + [m._parameters for m in mods]
        self._n_module_dicts = len(mods)
        self._object_ids = self._ids()
        self._params = [p for d in self._dicts[self._n_module_dicts:] for p in d.values() if p is not None]

    def _ids(self):
        """Identities of every submodule and Parameter object (None included) the tree holds now, in one pass."""
        return tuple(map(id, chain.from_iterable(map(dict.values, self._dicts))))

    def _signature(self, dev):
        """(device, storage addresses, version counters) of everything `commit` uploads: a change of any of them -- an optimiser
        step, load_state_dict, a replaced Parameter or submodule, .to(device) -- commits again.  Runs on every call, on the host's
        critical path for callers that wait for the device each step: C-level maps over the cached dicts (a Python loop building
        (ptr, version) pairs cost 87 us per call, profiles/r05_md_host_profile.txt)."""
        ids = self._ids()
        if ids != self._object_ids:   # a submodule or Parameter was replaced / added / removed: collect again
            self._collect_param_dicts()
        params = self._params + [m.nsb.factors for m in self.tb[:1]]
        params.append(self.atom_ref.elemental_energies)
        # the plan's device buffers live on the device it was committed under: a change of device is a change of plan state
        return (dev.index if dev.index is not None else torch.cuda.current_device(), tuple(map(_DATA_PTR, params)),
                tuple(map(_VERSION, params)))

    def commit(self) -> None:
        lib, plan = self.lib, self.plan
        for key, val in self.seq.state_dict().items():
            arr = _host_f32(val)
            _lib.check(lib.m3g_plan_set_param(plan, f"model.{key}".encode(), arr.ctypes.data, arr.size))
        em, dm, coeff = self.edge_feat.host_constants()
        consts = {"em": em, "dm": dm, "coeff": coeff, "elemental_energies": _host_f32(self.atom_ref.elemental_energies)}
        if self.tb:
            nsb = self.tb[0].nsb
            for other in self.tb[1:]:
                if not torch.equal(other.nsb.factors.cpu(), nsb.factors.cpu()):
                    raise RuntimeError("all ThreeBodyInteration blocks must share the same `nsb.factors`")
            consts["factors"] = _host_f32(nsb.factors)
            consts["bessel_zeros"] = _host_f32(nsb.spherical_bessel_zeros[: nsb.l_max, : nsb.n_max])
        else:
            consts["factors"] = np.ones(self.cfg.n_max, dtype=np.float32)
            consts["bessel_zeros"] = np.ones(self.cfg.n_max, dtype=np.float32)
        for name, arr in consts.items():
            _lib.check(lib.m3g_plan_set_const(plan, name.encode(), arr.ctypes.data, arr.size))
        _lib.check(lib.m3g_plan_commit(plan))

    PRECISIONS = {"fp32": 0, "bf16x3": 1, "f16x3": 2}

    def set_precision(self, name: str) -> None:
        """Arithmetic of the dense gated-MLP products (fp32 accumulate in every mode):
        "fp32"   (default) every product on v_mfma_f32_16x16x4_f32: exact fp32 products accumulated in k order, bitwise an fp32
                 `fmaf` chain -- the reference's arithmetic (nn/core.py:61-62, nn/featurizer.py:36);
        "f16x3"  opt-in, NARROWER than fp32: every operand scaled by a power of two and split in two fp16 parts (22-24 significant
                 bits), three v_mfma_f32_16x16x32_f16 products per fp32 product (the lo x lo term is dropped): errors ~1.8 x those
                 of an fp32 fmaf chain; every parity case (golden, LJ-fitted, saturated) inside the 1e-5 / 1e-4 tolerances;
        "bf16x3" opt-in, two bf16 parts (16 significant bits), three bf16 MFMA products: ~2^-16 product error, the fastest mode."""
        if name not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}, got {name!r}")
        self.set_option("precision", self.PRECISIONS[name])
        self.precision = name

    def set_option(self, name: str, value: int) -> None:
        """Engine options, e.g. set_option("edge_kernel", 0) selects the vector-ALU baseline kernels.

        set_option("graph_replay", 1) replays a captured hipGraph of the launch sequence while the call's buffers stay the
        same: the engine then REUSES its output tensors from call to call (copy what you need to keep), and inputs must be
        updated in place (same storage) to hit the cached graph."""
        _lib.check(self.lib.m3g_plan_set_option(self.plan, name.encode(), int(value)))
        if name == "graph_replay":
            self._graph_replay = bool(value)
            self._out_cache = {}

    # ---------------------------------------------------------------- measurement
    def profile(self, enable: bool) -> None:
        _lib.check(self.lib.m3g_profile_enable(self.plan, 1 if enable else 0))

    def count_launches(self, call) -> tuple:
        """(kernel launches, other stream operations) of ONE m3g_energy_forces call as `call()` issues it -- the un-profiled launch
        sequence, counted by the library from a stream capture of the call itself (m3g_count_launches)."""
        self._count_next, self.last_launch_count = True, None
        call()
        return self.last_launch_count

    def profile_read(self) -> dict:
        """{stage: (total_ms, launches)} from the HIP events recorded since the last read."""
        n = C.c_int32()
        names = (C.c_char_p * _lib.MAX_STAGES)()
        ms = (C.c_float * _lib.MAX_STAGES)()
        cnt = (C.c_int32 * _lib.MAX_STAGES)()
        _lib.check(self.lib.m3g_profile_read(self.plan, C.byref(n), names, ms, cnt))
        return {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(n.value)}

    # ---------------------------------------------------------------- the hot call
    def run(self, graph, want_forces: bool = True, extras: bool = True):
        pos = graph[K.POS]
        M._require_cuda(pos, K.POS)
        dev = pos.device
        sig = self._signature(dev)
        if sig != self._sig:
            with _cuda.on_device(dev):
                self.commit()
            self._sig = sig
        with _cuda.on_device(dev):
            types_in = graph[K.ATOM_TYPES]
            types = types_in.contiguous().long()
            probe = self._species_probe(graph, types_in, types)   # queued before the topology build, read after it (no extra wait)
            topo = M._Topology.of(graph, finish=False)   # (a trajectory graph's build may still be running: finished below)
            N, E, T, S = topo.N, topo.E, topo.T, topo.S
            D, R, Cc, B = self.cfg.embedding_dim, self.cfg.n_max, self.cfg.l_max * self.cfg.n_max, self.num_blocks
            pos_c = pos.detach().contiguous().float()
            shift = graph[K.EDGE_CELL_SHIFT].contiguous().to(torch.int32)
            lat = graph[K.LATTICE].contiguous().float()
            nbytes = C.c_size_t()
            _lib.check(self.lib.m3g_workspace_bytes(self.plan, N, E, T, S, C.byref(nbytes)))
            if self._workspace is None or self._workspace.numel() < nbytes.value or self._workspace.device != dev:
                self._workspace = None
                self._workspace = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
            f32 = dict(dtype=torch.float, device=dev)
            cache_key = (N, E, T, S, bool(want_forces), bool(extras), dev)
            out = self._out_cache.get(cache_key) if self._graph_replay else None
            if out is None:
                out = {K.TOTAL_ENERGY: torch.empty(S, **f32)}
                if want_forces:
                    out[K.FORCES] = torch.empty(N, 3, **f32)
                    out[K.STRESSES] = torch.empty(S, 6, **f32)
                out[K.SCALED_TOTAL_ENERGY] = torch.empty(S, **f32)
                out[K.SCALED_ATOMIC_ENERGIES] = torch.empty(N, **f32)
                if extras:
                    out[K.NODE_FEATURES] = torch.empty(N, D, **f32)
                    out[K.EDGE_ATTR] = torch.empty(E, D, **f32)
                    out[K.EDGE_DISTANCES] = torch.empty(E, **f32)
                    out[K.EDGE_WEIGHTS] = torch.empty(E, R, **f32)
                    out[K.TRIPLET_ANGLES] = torch.empty(T, **f32)
                    out[K.MID_EDGE_FEATURES] = torch.empty(B, E, Cc, **f32)
                if self._graph_replay:
                    self._out_cache[cache_key] = out
            topo.finish()   # waits for a queued build (everything above was host work beside it)
            self._check_species(graph, probe)
            p = M._ptr
            io = _lib.M3GIO(
                n_atoms=N, n_edges=E, n_triplets=T, n_structs=S, pos=p(pos_c), atom_types=p(types), edge_cell_shift=p(shift),
                lattice=p(lat), topo=p(topo.buf), triplet_edge_index=p(topo.tei),
                total_energy=p(out[K.TOTAL_ENERGY]), forces=p(out.get(K.FORCES)), stresses=p(out.get(K.STRESSES)),
                scaled_total_energy=p(out[K.SCALED_TOTAL_ENERGY]), scaled_atomic_energies=p(out[K.SCALED_ATOMIC_ENERGIES]),
                node_features=p(out.get(K.NODE_FEATURES)), edge_attr=p(out.get(K.EDGE_ATTR)),
                edge_distances=p(out.get(K.EDGE_DISTANCES)), edge_weights=p(out.get(K.EDGE_WEIGHTS)),
                triplet_angles=p(out.get(K.TRIPLET_ANGLES)), mid_edge_features=p(out.get(K.MID_EDGE_FEATURES)),
                topo_hints=topo.hints_for_call() if self.topology_hints else 0,
            )
            # The hot call leaves sticky error bits on the topology buffer instead of failing (e.g. a hints word that was not
            # certified for this buffer: the moment kernels then touch nothing and the results are INVALID, include/m3gnet_hip.h).
            # Read back once per topology, at its second call (the first one has long finished then: no pipeline stall).
            calls = getattr(topo, "_engine_calls", 0)
            if calls == 1 and topo.status():
                raise RuntimeError(f"m3g_energy_forces flagged topology error bits {topo.status():#x} on the previous call with this "
                                   "graph: its results were invalid (m3g_topology_status, include/m3gnet_hip.h)")
            topo._engine_calls = calls + 1
            if self._count_next:   # measurement (count_launches): this call's launch sequence, captured beside the call -- nothing executes twice
                self._count_next = False
                k, o = C.c_int32(), C.c_int32()
                _lib.check(self.lib.m3g_count_launches(self.plan, C.byref(io), p(self._workspace), self._workspace.numel(), C.byref(k), C.byref(o)))
                self.last_launch_count = (int(k.value), int(o.value))
            _lib.check(self.lib.m3g_energy_forces(self.plan, C.byref(io), p(self._workspace), self._workspace.numel(), M._stream()))
        for key, val in out.items():
            graph[key] = val
        if extras:
            # the remaining keys the reference's leading modules leave on the graph (nn/scale.py:24-29, nn/atom_ref.py:25-29)
            self.scale(graph)
            self.atom_ref(graph)
        return graph

    def _species_probe(self, graph, types_in: torch.Tensor, types: torch.Tensor):
        """The reference fails on a species index outside the model's table (`elemental_energies[atom_types]`,
        nn/atom_ref.py:27, raises IndexError; one_hot raises after it).  Checked once per atom_types tensor and cached on the
        graph like the topology.  The min/max reduction and its copy to pinned host memory are queued here; `_check_species`
        reads them after the topology build, whose own wait for the stream (new graph) has then already covered them.
        The cache key is taken from the tensor the GRAPH holds (kept alive by it, version-counted), never from the int64 /
        contiguous temporary made of it: a temporary's address is recycled and its version is always 0."""
        sig = (types_in.data_ptr(), types_in._version, tuple(types_in.shape), str(types_in.dtype), tuple(types_in.stride()),
               int(self.cfg.num_types))
        if isinstance(graph, dict) and graph.get("_m3g_species_ok") == sig:
            return None
        if not types.numel():
            return (sig, None, None)
        if self._species_host is None:
            self._species_host = torch.empty(2, dtype=torch.int64, pin_memory=True)
        lo, hi = torch.aminmax(types)
        self._species_host.copy_(torch.stack((lo, hi)), non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return (sig, self._species_host, done)

    def _check_species(self, graph, probe) -> None:
        if probe is None:
            return
        sig, host, done = probe
        if host is not None:
            done.synchronize()
            lo, hi = int(host[0]), int(host[1])
            if lo < 0 or hi >= self.cfg.num_types:
                raise IndexError(f"atom_types must lie in [0, {self.cfg.num_types - 1}] (num_types = {self.cfg.num_types}); "
                                 f"got values in [{lo}, {hi}]")
        if isinstance(graph, dict):
            dict.__setitem__(graph, "_m3g_species_ok", sig)
