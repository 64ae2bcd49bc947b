"""ctypes binding of libm3gnet_hip.so (C ABI: include/m3gnet_hip.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_PKG_ROOT = Path(__file__).resolve().parent.parent  # .../torch-m3gnet_amd
LIB_PATH = _PKG_ROOT / "lib" / "libm3gnet_hip.so"

M3G_OK, M3G_ERR_VALUE, M3G_ERR_STATE, M3G_ERR_SIZE, M3G_ERR_HIP, M3G_ERR_UNSUPPORTED = range(6)
ABI_VERSION = 6
VERLET_FILL_LISTS_MAX_ROW = 1024   # M3G_VERLET_FILL_LISTS_MAX_ROW (include/m3gnet_hip.h)


class M3GConfig(C.Structure):
    _fields_ = [
        ("cutoff", C.c_double), ("threebody_cutoff", C.c_double), ("energy_scale", C.c_double),
        ("length_scale", C.c_double), ("l_max", C.c_int32), ("n_max", C.c_int32), ("num_types", C.c_int32),
        ("embedding_dim", C.c_int32), ("num_blocks", C.c_int32), ("reserved", C.c_int32),
    ]


class M3GInfo(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device_count", C.c_int32), ("arch", C.c_char * 32)]


class M3GIO(C.Structure):
    _fields_ = [
        ("n_atoms", C.c_int64), ("n_edges", C.c_int64), ("n_triplets", C.c_int64), ("n_structs", C.c_int64),
        ("pos", C.c_void_p), ("atom_types", C.c_void_p), ("edge_cell_shift", C.c_void_p), ("lattice", C.c_void_p),
        ("topo", C.c_void_p), ("triplet_edge_index", C.c_void_p),
        ("total_energy", C.c_void_p), ("forces", C.c_void_p),
        ("stresses", C.c_void_p), ("scaled_total_energy", C.c_void_p), ("scaled_atomic_energies", C.c_void_p),
        ("node_features", C.c_void_p), ("edge_attr", C.c_void_p), ("edge_distances", C.c_void_p),
        ("edge_weights", C.c_void_p), ("triplet_angles", C.c_void_p), ("mid_edge_features", C.c_void_p),
        ("topo_hints", C.c_int32), ("reserved", C.c_int32),
    ]


class M3GMdLists(C.Structure):   # m3g_md_lists
    _fields_ = [
        ("n_atoms", C.c_int64), ("n_structs", C.c_int64), ("n_cand", C.c_int64), ("cap_edges", C.c_int64), ("cap_triplets", C.c_int64),
        ("cutoff", C.c_double), ("threebody_cutoff", C.c_double), ("skin", C.c_double),
        ("pos_ref", C.c_void_p), ("lattice", C.c_void_p), ("lattice32", C.c_void_p), ("batch", C.c_void_p), ("atom_types", C.c_void_p),
        ("cand_edge_index", C.c_void_p), ("cand_shift", C.c_void_p), ("cand_row_ptr", C.c_void_p), ("cand_state", C.c_void_p),
        ("verlet_scratch", C.c_void_p), ("verlet_scratch_bytes", C.c_size_t),
        ("edge_index", C.c_void_p), ("edge_cell_shift", C.c_void_p), ("triplet_edge_index", C.c_void_p), ("num_triplet_i", C.c_void_p),
        ("num_triplet_ij", C.c_void_p), ("pos32", C.c_void_p),
        ("topo", C.c_void_p), ("topo_bytes", C.c_size_t), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class M3GMdResult(C.Structure):   # m3g_md_result
    _fields_ = [("path", C.c_int32), ("topo_hints", C.c_int32), ("n_edges", C.c_int64), ("n_triplets", C.c_int64),
                ("max_displacement", C.c_double)]


MD_REUSE, MD_REFILL, MD_NEED_SEARCH, MD_UNSUPPORTED = range(4)

# name -> (restype, argtypes); every symbol include/m3gnet_hip.h declares
SYMBOLS = {
    "m3g_get_info": (C.c_int, [C.POINTER(M3GInfo)]),
    "m3g_last_error": (C.c_char_p, []),
    "m3g_plan_create": (C.c_int, [C.POINTER(M3GConfig), C.POINTER(C.c_void_p)]),
    "m3g_plan_destroy": (None, [C.c_void_p]),
    "m3g_plan_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "m3g_plan_set_const": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "m3g_plan_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "m3g_plan_commit": (C.c_int, [C.c_void_p]),
    "m3g_topology_bytes": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_size_t)]),
    "m3g_topology_build": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.c_void_p]),
    "m3g_topology_build_hints": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]),
    "m3g_topology_build_canonical": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]),
    "m3g_topology_build_canonical_begin": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "m3g_topology_build_canonical_end": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]),
    "m3g_topology_hints": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "m3g_topology_status": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "m3g_topology_active_edges": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "m3g_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_size_t)]),
    "m3g_energy_forces": (C.c_int, [C.c_void_p, C.POINTER(M3GIO), C.c_void_p, C.c_size_t, C.c_void_p]),
    "m3g_distance_angle": (C.c_int, [C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_edge_featurizer": (C.c_int, [C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "m3g_atom_featurizer": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_atom_ref": (C.c_int, [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_linear": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "m3g_multiply": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_bessel_basis": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_three_body": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_conv_block_scratch_bytes": (C.c_int, [C.c_int32, C.c_int64, C.POINTER(C.c_size_t)]),
    "m3g_conv_block": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "m3g_readout": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_verlet_scratch_bytes": (C.c_int, [C.c_int64, C.c_int64, C.POINTER(C.c_size_t)]),
    "m3g_verlet_rows": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_verlet_update": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_double, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_double),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    "m3g_verlet_update_async": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_double, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "m3g_verlet_fill": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_verlet_fill_lists": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_neighbor_scratch_bytes": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_size_t)]),
    "m3g_neighbor_count": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                     C.c_size_t, C.POINTER(C.c_int64), C.c_void_p]),
    "m3g_neighbor_count_triplets": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_float,
                                              C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    "m3g_threebody_build": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_neighbor_fill": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_double, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_threebody_scratch_bytes": (C.c_int, [C.c_int64, C.c_int64, C.POINTER(C.c_size_t)]),
    "m3g_threebody_count": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_int64), C.c_void_p]),
    "m3g_threebody_fill": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "m3g_debug_read_stamps": (C.c_int, [C.c_void_p, C.c_void_p]),
    "m3g_debug_live_handles": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "m3g_md_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "m3g_md_destroy": (None, [C.c_void_p]),
    "m3g_md_set_lists": (C.c_int, [C.c_void_p, C.POINTER(M3GMdLists)]),
    "m3g_md_invalidate": (C.c_int, [C.c_void_p]),
    "m3g_md_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(M3GMdResult),
                              C.c_void_p]),
    "m3g_topology_data_bytes": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_size_t)]),
    "m3g_topology_debug_last_path": (C.c_int, [C.POINTER(C.c_int32)]),
    "m3g_debug_exclusive_scan": (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "m3g_debug_radix_sort": (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "m3g_count_launches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "m3g_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "m3g_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_char_p), C.POINTER(C.c_float),
                                   C.POINTER(C.c_int32)]),
}
MAX_STAGES = 16

_lib = None


def build_library(verbose: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 with hipcc (make).  Cross-compiles without a GPU."""
    proc = subprocess.run(["make", "-C", str(_PKG_ROOT), "-j8"], capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout[-4000:])
        print(proc.stderr[-4000:])
    if proc.returncode != 0:
        raise RuntimeError("building libm3gnet_hip.so failed")
    return LIB_PATH


def load_library():
    """Load the HIP engine.  Fails loudly when it is missing -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C {_PKG_ROOT}` (or __graft_entry__.build()); "
            "torch_m3gnet has no CPU/eager fallback"
        )
    lib = C.CDLL(str(LIB_PATH), mode=os.RTLD_GLOBAL if hasattr(os, "RTLD_GLOBAL") else C.DEFAULT_MODE)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    """Translate an m3g_status into the exception the reference would raise."""
    if status == M3G_OK:
        return
    msg = load_library().m3g_last_error().decode(errors="replace")
    if status == M3G_ERR_VALUE:
        raise ValueError(msg)
    raise RuntimeError(f"m3gnet_hip error {status}: {msg}")
