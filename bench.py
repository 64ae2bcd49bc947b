#!/usr/bin/env python3
"""Benchmark of the hot path: energy + forces of M3GNet (default model, fp32) on the HIP engine.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[2], the "10k-atom PBC batch" the metric is quoted on): one jittered
fcc-Cu supercell of 10 x 10 x 25 cells = 10,000 atoms per GPU, cutoff 5 A / three-body cutoff 4 A
(E = 420,000 directed edges, T = 3,060,000 triplets), default model (l_max = n_max = 3, D = 64, 3 blocks,
95 species), random-init weights (seed 0), synthetic data.  A step = one `model(graph)` call = the fused
m3g_energy_forces launch sequence (forward + analytic reverse pass + virial), graph tensors resident in HBM.
With N > 1 each rank owns one independent supercell (structures are independent): weak
scaling, no data-path collective; the per-structure energies are all-gathered over RCCL every step.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      HBM roofline (algorithmic bytes) of the dominant kernel, its duration measured live with HIP events in the
                library; PMC-measured traffic and the matrix-pipe view of the same launch attached
  cpu_baseline  the CPU oracle (oracle/m3gnet_oracle.py, a plain-torch port of the reference) timed on the
                host cores on a bounded sample (2,048-atom Cu supercell), rank 0, N = 1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "torch-m3gnet_amd", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOPS_PER_EDGE_BLOCK = 134_144      # a14 65,536+384, a15 65,536+384, a8 MLP 2,304 (forward)
PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 matrix peak (no xf32/TF32 on gfx950)
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0               # HBM3E spec (about 6.3 TB/s achievable)
# The fused MFMA edge kernels (m3g_edge_mfma.hip).  Algorithmic figures per edge and launch:
#   bytes = edge-feature images the kernel must move under ideal fusion (256 B each): forward read e +
#           write e; reverse read saved e + read and write dL/de;   FLOPs from SURVEY.md 8(d) (backward = 1x forward).
#   MFMA counts per 16-edge tile: v_mfma_f32_16x16x32_bf16 (16,384 FLOP) / v_mfma_f32_16x16x4_f32 (2,048 FLOP).
EDGE_KERNELS = {
    "edge_block_fwd": dict(kernel="k_edge_block_mfma", alg_bytes_per_edge=2 * 256, alg_flops_per_edge=134_144,
                           bf16_mfma_per_tile=192, f32_mfma_per_tile=48),
    "edge_rev_fused": dict(kernel="k_edge_rev_fused", alg_bytes_per_edge=3 * 256, alg_flops_per_edge=134_144,
                           bf16_mfma_per_tile=396, f32_mfma_per_tile=48),
    # split reverse kernels (option rev_kernel = 0)
    "edge_rev_node_mlp": dict(kernel="k_edge_rev_node_mlp", alg_bytes_per_edge=2 * 256, alg_flops_per_edge=65_920,
                              bf16_mfma_per_tile=192, f32_mfma_per_tile=8),
    "edge_rev_edge_mlp": dict(kernel="k_edge_rev_edge_mlp", alg_bytes_per_edge=4 * 256, alg_flops_per_edge=68_224,
                              bf16_mfma_per_tile=204, f32_mfma_per_tile=56),
}
PMC_TRAFFIC_FILE = "r01c_pmc_hbm_traffic.json"   # profiles/: per-kernel FETCH_SIZE / WRITE_SIZE of this build (tools/pmc_traffic.py)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores(cap=64):
    """CPU threads this process may really use: affinity mask, clipped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, cap))


def build_workload(cells, seed, device):
    from helpers import fcc_cu_graph

    g = fcc_cu_graph(*cells, seed=seed)
    return g.to(device)


def default_model(device):
    from torch_m3gnet.model.build import build_model

    torch.manual_seed(0)
    model = build_model(5.0, 4.0, 3, 3, 95, 64, 3)
    return model.to(device)


def cpu_baseline(sample_cells=(8, 8, 8), steps=3):
    """Oracle (port of the reference's CPU path) on a bounded sample of the same workload."""
    from helpers import fcc_cu_graph
    from oracle import m3gnet_oracle as orc
    from torch_m3gnet.model.build import build_model

    cores = min(host_cores(), 16)  # the GPU box gives 16 cores per GPU
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads")
    torch.manual_seed(0)
    model = build_model(5.0, 4.0, 3, 3, 95, 64, 3)
    params = {f"model.{k}": v.detach().clone() for k, v in model.model.state_dict().items()}
    cfg = orc.OracleConfig()
    consts = orc.make_constants(cfg)
    g = fcc_cu_graph(*sample_cells, seed=0)
    graph = {k: g[k] for k in ("pos", "atom_types", "edge_index", "edge_cell_shift", "triplet_edge_index", "lattice", "batch")}
    n = int(g["pos"].size(0))
    orc.energy_forces(params, cfg, consts, graph)  # warm-up
    log("cpu_baseline: warm-up done")
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.energy_forces(params, cfg, consts, graph)
    dt = (time.perf_counter() - t0) / steps
    return {"value": n / dt, "unit": "atom-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n}-atom fcc Cu supercell ({'x'.join(map(str, sample_cells))} cells), fp32, {steps} timed steps "
                      f"after 1 warm-up, {dt * 1e3:.0f} ms/step, torch {torch.__version__} CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", type=int, nargs=3, default=[10, 10, 25], help="fcc cells per axis (4 atoms each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs a torch.distributed launch with WORLD_SIZE={args.gpus} (got {world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # one rank per GPU; the modulo only matters for the rehearsal mode below (several ranks sharing one GPU)
    backend = os.environ.get("M3G_BENCH_BACKEND", "nccl")   # "gloo": control-flow rehearsal of the N > 1 path on a one-GPU box
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from torch_m3gnet.data import MaterialGraphKey as K

    torch.set_num_threads(min(host_cores(), 16))
    model = default_model(device)
    log("building workload graph on the host")
    graph = build_workload(tuple(args.cells), seed=rank, device=device)
    log("graph on device")
    n_atoms = int(graph[K.POS].size(0))
    n_edges = int(graph[K.EDGE_INDEX].size(1))
    n_trip = int(graph[K.TRIPLET_EDGE_INDEX].size(1))
    gather_dev = device if backend == "nccl" else torch.device("cpu")
    energies_all = torch.empty(world, 1, device=gather_dev) if world > 1 else None

    def step():
        model(graph, forces=True, extras=False)
        if world > 1:
            dist.all_gather_into_tensor(energies_all, graph[K.TOTAL_ENERGY].view(1, 1).to(gather_dev))

    t_topo0 = time.perf_counter()
    step()  # first call: plan commit + topology build (index-only, cached on the graph) + workspace allocation
    torch.cuda.synchronize()
    first_call_s = time.perf_counter() - t_topo0
    log(f"first call {first_call_s:.2f} s")
    for _ in range(args.warmup):
        step()
    # index-only CSR build (m3g_topology_build), timed on its own: reused while the neighbour list is unchanged
    from torch_m3gnet.nn.modules import _Topology

    torch.cuda.synchronize()
    t1 = time.perf_counter()
    _Topology(graph)
    torch.cuda.synchronize()
    topo_ms = (time.perf_counter() - t1) * 1e3

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=gather_dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    log(f"timed region done: {ms_per_step:.3f} ms/step")
    value = world * n_atoms * args.steps / elapsed

    # ---- per-kernel device time: HIP events on the launch stream, recorded inside the library ----
    eng = model.engine
    eng.profile(True)
    for _ in range(args.steps):
        model(graph, forces=True, extras=False)
    torch.cuda.synchronize()
    stages = eng.profile_read()
    eng.profile(False)

    if rank == 0:
        per_launch = {k: (ms / max(cnt, 1), cnt) for k, (ms, cnt) in stages.items() if cnt}
        tiles = (n_edges + 15) // 16
        pmc_path = ROOT / "profiles" / PMC_TRAFFIC_FILE
        pmc = json.loads(pmc_path.read_text()) if (pmc_path.exists() and tuple(args.cells) == (10, 10, 25)) else {}

        def kernel_roofline(stage):
            """Roofline views of one fused edge kernel (each stage timer brackets exactly one launch of it)."""
            spec = EDGE_KERNELS[stage]
            ms = per_launch[stage][0]
            alg_bytes = n_edges * spec["alg_bytes_per_edge"]
            alg_flops = n_edges * spec["alg_flops_per_edge"]
            exe_flops = tiles * (spec["bf16_mfma_per_tile"] * 16384 + spec["f32_mfma_per_tile"] * 2048)
            rec = pmc.get(spec["kernel"])
            traffic = (2.0 * rec["fetch_kb"] + rec["write_kb"]) * 1024.0 if rec else None  # gfx950: FETCH_SIZE x 2
            achieved = alg_bytes / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": f"{spec['kernel']} (stage {stage})", "achieved": achieved, "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS, "traffic": traffic, "avg_launch_ms": ms,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "measured_traffic_GBs": (traffic / (ms * 1e-3) / 1e9) if traffic else None,
                    "mfma_view": {"algorithmic_flops_per_launch": alg_flops,
                                  "algorithmic_tflops": alg_flops / (ms * 1e-3) / 1e12,
                                  "executed_flops_per_launch": exe_flops,
                                  "executed_tflops": exe_flops / (ms * 1e-3) / 1e12,
                                  "peak_bf16_dense_tflops": PEAK_BF16_MFMA_TFLOPS, "peak_f32_tflops": PEAK_F32_MFMA_TFLOPS,
                                  "executed_frac_of_bf16_peak": exe_flops / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS}}

        views = {st: kernel_roofline(st) for st in EDGE_KERNELS if st in per_launch}
        dom = max(views, key=lambda k: views[k]["avg_launch_ms"])   # dominant kernel = longest average launch
        roofline = views[dom]
        roofline["note"] = ("dense chains run as 3x bf16 split MFMAs (fp32 accumulate) and the reverse pass recomputes every "
                            "activation in one fused kernel per block: the kernel is instruction-issue/latency-bound (PMC: VALU "
                            "~59 %, MFMA ~31 % of SIMD cycles at 2 waves/SIMD), below both the HBM and the matrix roofline; "
                            "`achieved` uses the ideal-fusion algorithmic bytes of DESIGN.md section 4, `traffic` is PMC-measured "
                            "HBM bytes per launch, `mfma_view` prices the same launch against the matrix peaks; hand-written streaming "
                            "kernels on this box reach 5.9-6.5 TB/s read-only and 4.8 TB/s copy (tools/hbm_bw_probe.hip), so the "
                            "measured read+write traffic rate is ~55 % of what a pure copy reaches")
        out = {
            "metric": "atom-steps/sec (energy+forces) on 10k-atom PBC batch, 1/2/4/8 MI355X", "value": value, "unit": "atom-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (dense chains: 3x bf16 split MFMA, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"single {n_atoms}-atom fcc Cu PBC supercell per GPU ({'x'.join(map(str, args.cells))} cells, "
                                   "a=3.61 A, jitter 0.025 A), r_cut 5 A / 3-body 4 A, default M3GNet (l_max=n_max=3, D=64, "
                                   "3 blocks), energy+forces+stress",
                       "atoms_per_gpu": n_atoms, "edges_per_gpu": n_edges, "triplets_per_gpu": n_trip,
                       "first_call_s_incl_topology_build": first_call_s, "topology_build_ms": topo_ms,
                       "stage_ms_per_step": {k: round(ms / args.steps, 4) for k, (ms, cnt) in stages.items() if cnt}},
            "roofline": roofline,
            "roofline_other_kernels": [v for k, v in views.items() if k != dom],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
