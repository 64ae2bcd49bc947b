#!/usr/bin/env python3
"""Benchmark of the hot path: energy + forces of M3GNet (default model) on the HIP engine.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config3|config4] [--precision fp32|f16x3|bf16x3]

`--gpus N` (N > 1) without a torch.distributed launcher in the environment starts the N ranks itself (one child
process per GPU, before anything touches the GPU in the parent) and relays rank 0's JSON line; under
`python -m torch.distributed.run ... bench.py --gpus N` the ranks come from the launcher.

Workloads
  config3 (default; BASELINE.json configs[2], the "10k-atom PBC batch" the metric is quoted on): one jittered fcc-Cu
          supercell of 10 x 10 x 25 cells = 10,000 atoms per GPU, cutoff 5 A / three-body cutoff 4 A (E = 420,000
          directed edges, T = 3,060,000 triplets).  A single cell does not shard: with N > 1
          every rank owns one independent supercell -- replicas, weak scaling, nothing crosses ranks inside a step; after
          the timed region the replicas' energies are all-gathered over RCCL once and compared.
  config4 (BASELINE.json configs[3]): 512 x N independent 64-atom random-species cells (seeds 0 .. 512 N - 1)
          partitioned over the N ranks by `torch_m3gnet.distributed.ShardedBatch` (greedy by triplet/edge cost, priced
          cooperatively, shard graphs built on the GPU), one RCCL all-gather of the per-structure energies per step.
          Also measured as a secondary figure (`config4_sharded`) after every default run.
Model: default M3GNet (l_max = n_max = 3, D = 64, 3 blocks, 95 species), random-init weights (seed 0), synthetic data.
A step = one `model(graph)` call = the m3g_energy_forces launch sequence (forward + analytic reverse pass + virial),
graph tensors resident in HBM.

Precision: the headline runs the engine's default exact `fp32` mode (every dense product on v_mfma_f32_16x16x4_f32, bitwise an fp32
fmaf chain: the reference's arithmetic); the opt-in split modes -- `f16x3` (every operand as two power-of-two-scaled fp16 parts,
three f16 MFMA products, fp32 accumulate: 22-24 significant bits, NARROWER than fp32) and `bf16x3` (~2^-16 relative product error)
-- are timed beside it and reported under `f16x3` / `bf16x3`, each with its operand format first in `dtype`.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      the dominant kernel against the roofline that bounds it (fp32 mode: the fp32 matrix peak, `frac` on the USEFUL
                FLOPs of the factorised formulation with the executed-MFMA figure beside it; bf16x3: HBM), its duration measured
                live with HIP events in the library.  Bytes on three bases, never mixed: `algorithmic_bytes_8d` (SURVEY.md 8(d):
                the 256-byte edge-feature rows only), `design_bytes` (what the data layout of DESIGN.md section 3 moves) and
                `traffic` (PMC, from the profile set named in `traffic_source` -- null when that set was collected on other
                kernel sources than the ones being timed)
  roofline_other_kernels   every other kernel of the step with its bound, bytes and achieved rate
  step_traffic_bytes       PMC traffic of one whole step against SURVEY.md 8(d)'s ideal-fusion bytes
  beside        (n_gpus = 1, default run) step latency of the 32-atom Cu cell of BASELINE configs[0] and one MD-style iteration on
                the headline cell with device-resident positions: skin-list test (lists reused) or full rebuild, then the step
  cpu_baseline  the CPU oracle (oracle/m3gnet_oracle.py, a plain-torch port of the reference) timed on the host cores ON THE
                HEADLINE CONFIGURATION ITSELF (the 10,000-atom cell; 1 warm-up + 3 steps at all cores, 1 step at 1 thread,
                ~25 s), rank 0, N = 1 only.  `vs_cpu_baseline` = value / cpu_baseline.value (`vs_baseline` stays null:
                BASELINE.md holds no published number for this metric)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "torch-m3gnet_amd"):   # (tests/ is NOT on the path: the bench does not import test modules)
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 matrix peak (no xf32/TF32 on gfx950)
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0               # HBM3E spec (about 6.3 TB/s achievable)
F32_MFMA_FLOP = 2048                # v_mfma_f32_16x16x4_f32
BF16_MFMA_FLOP = 16384              # v_mfma_f32_16x16x32_bf16
# Per-kernel work of the two fused edge kernels, per edge and launch:
#   bytes_8d       SURVEY.md 8(d), ideal fusion: forward reads + writes the 256-byte edge-feature row (512), reverse reads the saved
#                  row and reads + writes its gradient (768) -- the figure `traffic_over_algorithmic` is priced against;
#   design_bytes   what the data layout moves: + the activations the fp32 mode saves / reads back (SiLU'(p1) and p2 of both MLPs,
#                  1 KB each) and the dL/dp1 hand-over rows (blocks > 0 only: averaged over the 3 launches of a step; 1 KB fp32 rows
#                  in the fp32 mode, 768-byte 24-bit rows in the others, + 16 B of scales per row in the f16x3 mode);
#   flops_8d       SURVEY.md 8(d): 134,144 per edge and block forward (a14 65,920 + a15 65,920 + a8 MLP 2,304), the same again for
#                  the input-gradient reverse -- the UNFACTORISED formulation W1 [x_i | x_j | e];
#   flops_useful   what the kernels really have to do after the exact factorisation W1 [x_i | x_j | e] = TA[i] + TB[j] + W1c e
#                  (DESIGN.md section 2): the x_i / x_j thirds of layer 1 (2 x 32,768 FLOP per edge and MLP) become per-NODE
#                  tables built once per block by k_node_pre_mfma (65,536 FLOP per ATOM, counted there), so per edge
#                  2 MLPs x (16,384 layer 1 + 16,384 layer 2 + 384 W_l h) + 2,304 three-body MLP = 68,608: half of flops_8d;
#   mfma           MFMAs issued per 16-edge tile (f32 16x16x4, bf16 16x16x32): executed FLOPs incl. zero padding.
EDGE_KERNELS = {
    "edge_block_fwd": dict(kernel="k_edge_block_mfma", flops_8d=134_144, flops_useful=68_608, bytes_8d=2 * 256,
                           design_bytes={"bf16x3": 2 * 256, "f16x3": 2 * 256, "fp32": 2 * 256 + 2048},
                           mfma={"bf16x3": (48, 192), "f16x3": (8, 216), "fp32": (545, 0)}),   # f16x3: the three-body MLP is on the f16 pipe too
    "edge_rev_fused": dict(kernel={"bf16x3": "k_edge_rev_fused", "f16x3": "k_edge_rev_fused", "fp32": "k_edge_rev_f32"}, flops_8d=134_144,
                           flops_useful=68_608, bytes_8d=3 * 256,
                           design_bytes={"bf16x3": 3 * 256 + 256 + (2 * 768) / 3, "f16x3": 3 * 256 + 256 + (2 * (768 + 16)) / 3,
                                         "fp32": 2048 + (2 * (512 + 1024) + 256) / 3},
                           mfma={"bf16x3": (48, 396), "f16x3": (8, 444), "fp32": (577, 0)}),
    # split reverse kernels (option rev_kernel = 0; fp32: layer 1 saved, layer 2 recomputed)
    "edge_rev_node_mlp": dict(kernel="k_edge_rev_node_mlp", flops_8d=65_920, flops_useful=33_152, bytes_8d=2 * 256,
                              design_bytes={"bf16x3": 2 * 256, "f16x3": 2 * 256, "fp32": 512 + 256 + 512},
                              mfma={"bf16x3": (8, 192), "f16x3": (8, 192), "fp32": (388, 0)}),
    "edge_rev_edge_mlp": dict(kernel="k_edge_rev_edge_mlp", flops_8d=68_224, flops_useful=35_456, bytes_8d=3 * 256,
                              design_bytes={"bf16x3": 4 * 256, "f16x3": 4 * 256, "fp32": 512 + 3 * 256 + 512},
                              mfma={"bf16x3": (56, 204), "f16x3": (56, 204), "fp32": (444, 0)}),
}
FLOPS_RATIO_NOTE = ("flops_8d / flops_useful = 2: SURVEY.md 8(d) counts W1 [x_i | x_j | e] per edge; the kernels use the exact "
                    "factorisation TA[i] + TB[j] + W1c e, whose x_i / x_j parts are per-node tables (k_node_pre_mfma, 65,536 FLOP "
                    "per atom and block, 42 x fewer rows than edges)")
BYTES_8D_PER_STEP = lambda E, T, N: 4008 * E + 48 * T + 3504 * N   # noqa: E731  SURVEY.md 8(d), D = 64, B = 3
PEAK_F16_MFMA_TFLOPS = 2500.0       # dense f16 MFMA peak (same rate as bf16)
PMC_SQ_FILE = "r06_pmc_sq_counters.json"        # profiles/: per-kernel SQ counters (tools/pmc_sq_json.py), stamped like the traffic set
PMC_TRAFFIC_FILE = "r06_pmc_hbm_traffic.json"   # profiles/: per-kernel FETCH_SIZE / WRITE_SIZE (tools/pmc_traffic.py), stamped with
                                                # the digest of the kernel sources it was collected on
DEFAULT_PRECISION = "fp32"   # the reference's arithmetic (fp32 end to end): the headline `value` / `dtype` / `roofline`
DTYPE = {"fp32": "f32",
         "f16x3": "2xf16 split operands (power-of-two scaled, 22-24 significant bits; lo x lo dropped), 3 f16 MFMA products, f32 accumulate",
         "bf16x3": "2xbf16 split operands (16 significant bits), 3 bf16 MFMA products, f32 accumulate"}
METRIC = "atom-steps/sec (energy+forces) on 10k-atom PBC batch, 1/2/4/8 MI355X"


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


def cpu_model():
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def default_model(device):
    from torch_m3gnet.config import ModelConfig   # the reference's defaults (config.py:10-17)

    torch.manual_seed(0)
    return ModelConfig().build().to(device)


def csrc_digest():
    """sha256 over the kernel sources of this tree (csrc/*.hip, csrc/*.h, include/*.h): ties a PMC profile set to the build it
    was collected on (the GPU box has no .git)."""
    h = hashlib.sha256()
    files = sorted((ROOT / "torch-m3gnet_amd" / "csrc").glob("*.h*")) + sorted((ROOT / "include").glob("*.h"))
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def physical_cores():
    """Physical cores among the CPUs this process may run on (distinct (package, core) pairs of /proc/cpuinfo inside the affinity
    mask), clipped by the cgroup quota like host_cores()."""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cpu, pkg = set(), None, 0
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            key, _, val = line.partition(":")
            key = key.strip()
            if key == "processor":
                cpu = int(val)
            elif key == "physical id":
                pkg = int(val)
            elif key == "core id" and cpu in allowed:
                seen.add((pkg, int(val)))
        if seen:
            return max(1, min(len(seen), host_cores(cap=1 << 16)))
    except Exception:
        pass
    return host_cores(cap=1 << 16)


def cpu_baseline(cells=(10, 10, 25), steps=3):
    """Oracle (port of the reference's CPU path) on the headline configuration itself, as BASELINE.md's "CPU-baseline plan" prescribes:
    1 warm-up + `steps` steps at ALL physical host cores (`value`), then 1 step at 16 threads (the per-GPU host share of a box of this
    pool: the figure of rounds 1-5, kept as `threads_16`) and 1 step at 1 thread."""
    from oracle import m3gnet_oracle as orc   # the checker, used here only as the thing timed (allowed: cpu_baseline leg)
    from torch_m3gnet.config import ModelConfig
    from torch_m3gnet.data.synthetic import fcc_cu_graph

    cores = physical_cores()
    torch.manual_seed(0)
    model = ModelConfig().build()
    params = {f"model.{k}": v.detach().clone() for k, v in model.model.state_dict().items()}
    cfg = orc.OracleConfig()
    consts = orc.make_constants(cfg)
    g = fcc_cu_graph(*cells, seed=0)
    graph = {k: g[k] for k in ("pos", "atom_types", "edge_index", "edge_cell_shift", "triplet_edge_index", "lattice", "batch")}
    n = int(g["pos"].size(0))

    def run(threads, n_steps, warm):
        torch.set_num_threads(threads)
        for _ in range(warm):
            orc.energy_forces(params, cfg, consts, graph)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            orc.energy_forces(params, cfg, consts, graph)
        return (time.perf_counter() - t0) / n_steps

    log(f"cpu_baseline: {n} atoms, {cores} threads (all physical cores this process may use)")
    dt = run(cores, steps, 1)
    log(f"cpu_baseline: {dt * 1e3:.0f} ms/step; 16 threads")
    rec = {"value": n / dt, "unit": "atom-steps/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(), "ms_per_step": dt * 1e3,
           "host_logical_cpus": os.cpu_count(), "usable_logical_cpus": host_cores(cap=1 << 16),
           "cores_note": "cores = physical cores inside this process's affinity mask / cgroup quota: everything the box lets one job use "
                         "(a one-GPU box of this pool exposes 16 of its host's cores)"}
    if cores > 16:
        dt16 = run(16, 1, 0)   # (code and allocator are warm from the runs above)
        rec["threads_16"] = {"value": n / dt16, "unit": "atom-steps/s", "cores": 16, "ms_per_step": dt16 * 1e3,
                             "note": "the per-GPU host-core share of a one-GPU box of this pool (the figure rounds 1-5 reported as `value`)"}
        log(f"cpu_baseline: {dt16 * 1e3:.0f} ms/step; 1 thread")
    dt1 = run(1, 1, 0)
    torch.set_num_threads(min(cores, 16))
    rec["threads_1"] = {"value": n / dt1, "unit": "atom-steps/s", "cores": 1, "ms_per_step": dt1 * 1e3}
    rec["threads_all"] = {"value": n / dt, "unit": "atom-steps/s", "cores": cores, "ms_per_step": dt * 1e3}   # (= `value`; named for the record)
    rec["sample"] = (f"the headline workload itself: {n}-atom fcc Cu supercell ({'x'.join(map(str, cells))} cells), fp32, {steps} timed steps after 1 warm-up at "
                     f"{cores} threads = all physical cores of the host this process may use ({dt * 1e3:.0f} ms/step), then "
                     f"{'1 timed step at 16 threads (%.0f ms/step) and ' % rec['threads_16']['ms_per_step'] if 'threads_16' in rec else ''}1 timed step at 1 thread ({dt1 * 1e3:.0f} ms/step), "
                     f"torch {torch.__version__} CPU")
    return rec


# ---------------------------------------------------------------------------------------------- self launch
def supervise(procs, poll_s=0.2, grace_s=5.0):
    """Wait for all children; as soon as ANY exits non-zero, terminate the others and return (its code, its rank).  A rank that
    dies before or inside the rendezvous / a collective would otherwise leave its peers waiting until a store or RCCL time-out
    of tens of minutes.  Returns (0, None) when every child exited cleanly."""
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            return 0, None
        time.sleep(poll_s)
    for p in procs:
        if p.poll() is None:
            p.terminate()
    deadline = time.time() + grace_s
    for p in procs:
        try:
            p.wait(timeout=max(0.1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    rank, code = failed
    return (abs(code) if code else 1), rank


def self_launch(args) -> int:
    """Start one FRESH child per rank (before any GPU call in this process -- a GPU-initialised process is never re-exec'ed),
    supervise them, relay rank 0's JSON line; any rank failing ends the job with its exit code and the tail of its stderr."""
    import tempfile

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, outs, errs = [], [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = tempfile.TemporaryFile(mode="w+") if r == 0 else subprocess.DEVNULL
        err = tempfile.TemporaryFile(mode="w+")
        outs.append(out)
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env, stdout=out, stderr=err, text=True))
    code, rank = supervise(procs)
    for r, err in enumerate(errs):   # relay the children's logs: rank 0 in full, the tails of the others
        err.seek(0)
        lines = err.read().splitlines()
        for line in (lines if r == 0 else lines[-15:]):
            sys.stderr.write(f"[rank {r}] {line}\n")
    if code:
        log(f"rank {rank} exited with code {code}: the other ranks were terminated")
        return code
    outs[0].seek(0)
    sys.stdout.write(outs[0].read())
    sys.stdout.flush()
    return 0


# ---------------------------------------------------------------------------------------------- measurement
class Job:
    """torch.distributed context of this rank."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if args.gpus != self.world:
            raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
        # "gloo": control-flow rehearsal of the N > 1 path on a one-GPU box (several ranks share the GPU)
        self.backend = os.environ.get("M3G_BENCH_BACKEND", "nccl")
        dev_index = local_rank if self.backend == "nccl" else local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        self.device = torch.device("cuda", dev_index)
        self.dist = None
        # M3G_BENCH_FORCE_DIST=1: a ONE-rank process group, so that the RCCL code of the N > 1 path (init with device_id,
        # device-buffer all-gather, all-reduce, barrier) executes on a one-GPU box (tests/test_gpu_sharded.py)
        if self.world > 1 or os.environ.get("M3G_BENCH_FORCE_DIST") == "1":
            import datetime

            import torch.distributed as dist

            if os.environ.get("M3G_BENCH_TEST_DIE_RANK") == str(self.rank):   # test hook: this rank dies before the rendezvous
                os._exit(7)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            timeout = datetime.timedelta(seconds=float(os.environ.get("M3G_BENCH_INIT_TIMEOUT_S", "180")))
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.device, timeout=timeout, rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group(self.backend, timeout=timeout, rank=self.rank, world_size=self.world)
            self.dist = dist
        self.comm_device = self.device if self.backend == "nccl" else torch.device("cpu")

    def sync_all(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, step, steps, warmup):
        """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize; max over ranks (s)."""
        for _ in range(warmup):
            step()
        self.sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.sync_all()
        elapsed = time.perf_counter() - t0
        if self.dist is not None:
            tmax = torch.tensor([elapsed], device=self.comm_device, dtype=torch.float64)
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def series_and_clock(step, steps, sample_clock=True):
    """A pass of `steps` further steps AFTER the timed region (which stays untouched): every step bracketed by HIP events on the
    launch stream -> min / median / max step time, and the shader clock sampled from the SMU on a second thread while the steps
    run (torch.cuda.clock_rate -> amdsmi: MHz of the current device).  Tells a slow box / a throttled clock from warm-up ramp."""
    import statistics
    import threading

    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                samples.append(int(torch.cuda.clock_rate()))
            except Exception:
                return
            stop.wait(0.004)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler, daemon=True)
    if sample_clock:   # (rank 0 only: one SMU client per node is enough)
        th.start()
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    stop.set()
    if sample_clock:
        th.join(timeout=2.0)
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    rec = {"ms_per_step_min": min(ms), "ms_per_step_median": statistics.median(ms), "ms_per_step_max": max(ms), "steps": steps,
           "how": "a second pass of the same steps, each bracketed by HIP events on the launch stream (the timed region itself carries no events)"}
    busy = [c for c in samples if c > 0]
    if busy:
        rec["clock_mhz"] = {"median": statistics.median(busy), "min": min(busy), "max": max(busy), "samples": len(busy),
                            "source": "torch.cuda.clock_rate() (amdsmi current sclk) sampled every ~4 ms during that pass"}
    return rec


def load_stamped(name, digest, cells):
    """A profile set under profiles/ that carries the digest of the kernel sources it was collected on; ({}, source record)."""
    path = ROOT / "profiles" / name
    source = {"file": f"profiles/{name}", "csrc_sha256_of_this_build": digest, "valid": False}
    if path.exists() and tuple(cells) == (10, 10, 25):
        loaded = json.loads(path.read_text())
        src = loaded.get("_source", {})
        source.update({k: src.get(k) for k in ("csrc_sha256", "git_commit", "command")})
        if src.get("csrc_sha256") == digest:
            source["valid"] = True
            return loaded, source
        source["note"] = "profile set collected on other kernel sources than this build: fields taken from it are null"
    else:
        source["note"] = "no profile set for this workload"
    return {}, source


def stage_times(model, call, steps):
    """{stage: (ms per launch, launches per step)} from HIP events recorded on the launch stream inside the library."""
    eng = model.engine
    eng.profile(True)
    for _ in range(steps):
        call()
    torch.cuda.synchronize()
    stages = eng.profile_read()
    eng.profile(False)
    return {k: (ms / cnt, cnt / steps) for k, (ms, cnt) in stages.items() if cnt}


def rooflines(per_launch, n_atoms, n_edges, n_trip, n_active, precision, pmc, moments=True):
    """Per-kernel roofline records (DESIGN.md section 4 states every byte / FLOP figure used here).  `pmc` = {kernel: {fetch_kb,
    write_kb}} of the profile set that matches this build, or {} (then every `traffic` is null)."""
    tiles = (n_edges + 15) // 16
    views = {}

    def pmc_bytes(kernel):
        rec = pmc.get(kernel)
        return (2.0 * rec["fetch_kb"] + rec["write_kb"]) * 1024.0 if rec else None   # gfx950: FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM)

    for stage, spec in EDGE_KERNELS.items():
        if stage not in per_launch:
            continue
        ms = per_launch[stage][0]
        sec = ms * 1e-3
        n_f32, n_bf16 = spec["mfma"][precision]
        kname = spec["kernel"][precision] if isinstance(spec["kernel"], dict) else spec["kernel"]
        bytes_8d = n_edges * spec["bytes_8d"]
        design = n_edges * spec["design_bytes"][precision]
        useful = n_edges * spec["flops_useful"]
        exe_flops = tiles * (n_f32 * F32_MFMA_FLOP + n_bf16 * BF16_MFMA_FLOP)
        traffic = pmc_bytes(kname)
        common = {"kernel": f"{kname} (stage {stage})", "avg_launch_ms": ms, "launches_per_step": per_launch[stage][1],
                  "algorithmic_bytes_8d": bytes_8d, "design_bytes": design, "traffic": traffic,
                  "traffic_over_algorithmic": (traffic / bytes_8d) if traffic else None,
                  "traffic_over_design": (traffic / design) if traffic else None,
                  "measured_traffic_GBs": (traffic / sec / 1e9) if traffic else None,
                  "algorithmic_flops": useful, "algorithmic_flops_8d_unfactorised": n_edges * spec["flops_8d"],
                  "executed_mfma_flops": exe_flops}
        hbm_rate = bytes_8d / sec / 1e9
        hbm_view = {"bound": "hbm", "achieved": hbm_rate, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_rate / PEAK_HBM_GBS}
        if precision == "fp32":
            # exact-fp32 MFMA chains: priced on the USEFUL FLOPs of the factorised formulation against the fp32 matrix peak, the
            # executed-MFMA rate (zero padding of the K = 9 / 3 products included) beside it.  fp32 MFMAs and vector instructions
            # share the SIMD's fp32 datapath on gfx950 (profiles/r03_mfma_filler_probe.txt): the kernel's own floor is
            # 32 cycles x MFMAs + its vector instructions, not the MFMAs alone.
            t_useful, t_exe = useful / sec / 1e12, exe_flops / sec / 1e12
            mfma_view = {"bound": "mfma", "achieved": t_useful, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": t_useful / PEAK_F32_MFMA_TFLOPS, "executed_mfma_tflops": t_exe,
                         "executed_mfma_frac": t_exe / PEAK_F32_MFMA_TFLOPS, "flops_note": FLOPS_RATIO_NOTE}
            views[stage] = dict(common, **mfma_view, other_view=hbm_view)
        else:
            # split modes: the products execute on the f16 / bf16 matrix pipe, which (unlike the fp32 MFMA) co-executes with the
            # vector instructions -- `frac` is priced against THAT pipe on the EXECUTED FLOPs (3 part products per fp32 product, zero
            # padding included), the HBM view on PMC traffic beside it (`hbm_traffic_view`; falls back to the algorithmic bytes when
            # no profile set matches this build), and the useful fp32 FLOPs against the fp32 matrix peak only as a yardstick
            # (`useful_vs_fp32_peak`: that peak does not bound a kernel on the f16 pipe).  Neither pipe bounds these kernels: they
            # are bound by their vector instruction stream (operand scaling / splitting, activations) -- `valu_per_mfma`,
            # `valu_active_frac` from the SQ counter set say so in the line.
            t_useful, t_exe = useful / sec / 1e12, exe_flops / sec / 1e12
            peak = PEAK_F16_MFMA_TFLOPS if precision == "f16x3" else PEAK_BF16_MFMA_TFLOPS
            traffic_rate = (traffic / sec / 1e9) if traffic else None
            mfma_view = {"bound": "mfma", "achieved": t_exe, "peak": peak, "unit": "TFLOP/s", "frac": t_exe / peak,
                         "frac_is": "executed MFMA FLOPs (incl. the 3x of the split and zero padding) / dense %s peak" % ("f16" if precision == "f16x3" else "bf16"),
                         "useful_vs_fp32_peak": t_useful / PEAK_F32_MFMA_TFLOPS, "useful_tflops": t_useful,
                         "hbm_traffic_view": {"achieved": traffic_rate if traffic_rate else hbm_rate, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                              "frac": (traffic_rate if traffic_rate else hbm_rate) / PEAK_HBM_GBS,
                                              "bytes_basis": "PMC traffic (2 x FETCH_SIZE + WRITE_SIZE)" if traffic_rate else "algorithmic_bytes_8d"},
                         "flops_note": FLOPS_RATIO_NOTE}
            views[stage] = dict(common, **mfma_view, other_view=hbm_view)
    # HBM-bound kernels: bytes the data layout of DESIGN.md section 3 makes each launch move (every array the kernel must read or
    # write once; gathers from L2/MALL-resident node tables not counted)
    E, T, N, A = n_edges, n_trip, n_atoms, n_active
    dp1_row = {"fp32": 1024, "bf16x3": 768, "f16x3": 768 + 16}[precision]   # dL/dp1 hand-over row per edge
    hbm_kernels = {
        "geometry_basis": ("k_geometry", E * (8 + 12 + 12 + 4 + 16 + 16) + A * (64 + 64 + 8), 0),
        # three-body aggregate: the moment kernels (complete partner lists) read no partner ids; the list kernels
        # one byte per triplet and role.  Which pair ran is read off the profile set (tools/pmc_traffic.py names them apart).
        **({"threebody_fwd": ("k_threebody_moments_fwd", A * (64 + 12 + 4 + 4 + 64), 0),
            "threebody_rev": ("k_threebody_moments_rev", A * (64 + 64 + 12 + 4 + 4 + 64 + 64 + 16), 0)} if moments else
           {"threebody_fwd": ("k_threebody_fwd", A * (64 + 12 + 4 + 4 + 64) + T * 1, 0),
            "threebody_rev": ("k_threebody_rev", A * (64 + 64 + 12 + 4 + 4 + 64 + 64 + 16) + 2 * T * 1, 0)}),
        "node_rev": ("k_node_reverse", E * (dp1_row + 8) + N * (64 + 256 + 256) * 4, N * 2 * 256 * 64 * 2),
        # (x in, TA / TB / v / x out, and a third of the 135-KB weight image per workgroup: 3 x min(ceil(tiles / 4), 256) workgroups)
        "node_pre": ("k_node_pre_mfma", N * (3 * 256 + 2 * 1024 + 64 + 256) + 47 * 1024 * 3 * min((N + 63) // 64, 256), N * 2 * 528 * 64),
        "geometry_rev_forces": ("k_geometry_reverse+k_force_gather+k_struct_stress", E * (16 * 3 + 12 + 4 + 12 + 12 * 2 + 8) + N * 12, 0),
    }
    for stage, (kernel, nbytes, flops) in hbm_kernels.items():
        if stage not in per_launch:
            continue
        ms = per_launch[stage][0]
        parts = [pmc_bytes(k) for k in kernel.split("+")]
        traffic = sum(parts) if all(x is not None for x in parts) else None
        rate = nbytes / (ms * 1e-3) / 1e9
        views[stage] = {"bound": "hbm", "kernel": f"{kernel} (stage {stage})", "achieved": rate, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": rate / PEAK_HBM_GBS, "traffic": traffic, "avg_launch_ms": ms, "launches_per_step": per_launch[stage][1],
                        "design_bytes": nbytes, "measured_traffic_GBs": (traffic / (ms * 1e-3) / 1e9) if traffic else None}
        if flops:
            views[stage]["table_flops_per_launch"] = flops   # the per-node share of the factorised layer 1 (see FLOPS_RATIO_NOTE)
    return views


def measure_config4(job, model, steps, warmup):
    """BASELINE config 4: 512 x world independent 64-atom cells sharded by ShardedBatch; returns the record."""
    from torch_m3gnet.data import MaterialGraphKey as K
    from torch_m3gnet.data.synthetic import random_cell_arrays
    from torch_m3gnet.distributed import ShardedBatch

    n_structs = 512 * job.world
    t0 = time.perf_counter()
    sb = ShardedBatch.from_structures(n_structs, lambda i: random_cell_arrays(64, 9.1, seed=i), 5.0, 4.0, device=job.device)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    evaluate = lambda b: model(b, forces=True, extras=False)   # noqa: E731
    energies = None

    def step():
        nonlocal energies
        energies, _ = sb.evaluate(evaluate)

    step()
    elapsed = job.timed(step, steps, warmup)
    n_atoms = sum(sb.sizes)
    loads = [sum(sb.costs[i] for i in s) for s in sb.shards]
    return {"workload": f"{n_structs} independent 64-atom random-species cells (L = 9.1 A, seeds 0..{n_structs - 1}), "
                        f"{512} per GPU, partitioned by ShardedBatch (greedy by triplets + 32 edges), energies all-gathered "
                        f"({job.backend}) every step",
            "value": n_atoms * steps / elapsed, "unit": "atom-steps/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
            "structures": n_structs, "atoms": n_atoms, "local_atoms": sb.n_local_atoms,
            "local_edges": int(sb.batch[K.NUM_EDGES]), "local_triplets": int(sb.batch[K.NUM_TRIPLETS]),
            "partition_imbalance": max(loads) / (sum(loads) / len(loads)) - 1.0, "shard_build_s": build_s,
            "energy_checksum": float(energies.double().sum())}


def measure_beside(model, device):
    """Figures beside the throughput headline (n_gpus = 1 only): the step latency of BASELINE configs[0]'s 32-atom Cu cell, and one
    MD-style iteration on the headline cell with the positions RESIDENT ON THE DEVICE (torch_m3gnet.data.md.VerletGraph over the
    C ABI's m3g_verlet_*): fresh jittered positions every iteration -> skin-list test -> energies and forces.
      reuse    the lists are unchanged (the jitter moves no shell across 5 A / 4 A): index tensors, CSR topology and its
               certificate are reused, only the positions are new -- bit-identical to a fresh build (tests/test_gpu_md.py);
               `reuse_verdict_read_after_the_step`: the same with the evaluation queued behind the skin test before its verdict
               is read (VerletGraph.evaluate: no wait in front of the step; a changed list would re-run the step);
      refill   the same iteration forced to re-derive the lists from the skin list (no search) and rebuild triplets, topology and
               certificate: what a step costs when some pair has crossed a cutoff -- every step of a liquid or a hot crystal;
      rebuild  the same iteration forced through a new candidate search (cutoff + skin), list fill, triplets, topology and
               certificate: what a step costs when an atom has moved further than skin / 2.
    Both in the headline's arithmetic mode and in the opt-in f16x3 mode."""
    import numpy as np
    from torch_m3gnet.data.md import VerletGraph
    from torch_m3gnet.data.synthetic import fcc_cu_graph

    def per_call(fn, reps, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    small = fcc_cu_graph(2, 2, 2).to(device)
    rec = {"step_ms_32_atom_cu_cell": per_call(lambda: model(small, forces=True, extras=False), 200, warm=20)}
    mid = fcc_cu_graph(6, 6, 6).to(device)
    rec["step_ms_864_atom_cu_cell"] = per_call(lambda: model(mid, forces=True, extras=False), 100, warm=10)
    for cells in ((8, 8, 8), (10, 10, 10)):   # 2,048 / 4,000 atoms: the sizes between the small-system launches and the headline cell
        gm = fcc_cu_graph(*cells).to(device)
        rec[f"step_ms_{4 * cells[0] ** 3}_atom_cu_cell"] = per_call(lambda: model(gm, forces=True, extras=False), 50, warm=5)
        del gm
    # MD-style iteration on the two small cells (VerletGraph.evaluate, reuse path: what an MD user of a small cell pays per step)
    for n_cells, key in ((2, "md_iteration_ms_32_atom_cell"), (6, "md_iteration_ms_864_atom_cell")):
        gi_s = np.stack(np.meshgrid(np.arange(n_cells), np.arange(n_cells), np.arange(n_cells), indexing="ij"), -1)
        base_s = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
        p0 = torch.tensor((gi_s.reshape(-1, 1, 3) + base_s[None]).reshape(-1, 3) * 3.61, device=device)
        vgs = VerletGraph([np.eye(3) * n_cells * 3.61], [np.full(p0.size(0), 29)], 5.0, 4.0, skin=0.5, device=device)
        gen_s = torch.Generator(device=device)
        gen_s.manual_seed(0)

        def it_small():
            vgs.evaluate(model, p0 + (torch.rand(p0.shape, generator=gen_s, device=device, dtype=torch.float64) - 0.5) * 0.05, forces=True, extras=False)

        rec[key] = per_call(it_small, 200, warm=20)

        def it_small_refill():   # the lists re-derived on every step (what a cell at finite temperature does): host-bound
            model(vgs.update(p0 + (torch.rand(p0.shape, generator=gen_s, device=device, dtype=torch.float64) - 0.5) * 0.05, force="refill"),
                  forces=True, extras=False)

        rec[key.replace("md_iteration_ms", "md_refill_iteration_ms")] = per_call(it_small_refill, 100, warm=10)

        def it_small_step_refill():   # the same through one library call per step (VerletGraph.step -> m3g_md_step)
            vgs.step(model, p0 + (torch.rand(p0.shape, generator=gen_s, device=device, dtype=torch.float64) - 0.5) * 0.05, force="refill")

        rec[key.replace("md_iteration_ms", "md_step_refill_iteration_ms")] = per_call(it_small_step_refill, 100, warm=10)
    # BASELINE config 5 (2,000 atoms in L = 31.1 A, cutoff 6 A, three-body cutoff 4 A and 6 A): step time on a model of those cutoffs
    from torch_m3gnet.data.graph_gpu import batch_from_arrays
    from torch_m3gnet.data.synthetic import random_cell_arrays
    from torch_m3gnet.model.build import build_model

    lat5, pos5, z5 = random_cell_arrays(2000, 31.1, seed=0)
    for tb in (4.0, 6.0):
        torch.manual_seed(0)
        m5 = build_model(6.0, tb, 3, 3, 95, 64, 3).to(device)
        m5.engine.set_precision(model.engine.precision)
        g5 = batch_from_arrays([lat5], [pos5], [z5], 6.0, tb)
        rec[f"step_ms_config5_r3_{int(tb)}A"] = per_call(lambda: m5(g5, forces=True, extras=False), 20, warm=3)
        del m5, g5
    a = 3.61
    base = np.array([[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]])
    gi = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(25), indexing="ij"), -1)
    pos0 = torch.tensor((gi.reshape(-1, 1, 3) + base[None]).reshape(-1, 3) * a, device=device)   # fp64, on the device
    lat = np.diag([10 * a, 10 * a, 25 * a]).astype(float)
    z = np.full(pos0.size(0), 29)
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    vg = VerletGraph([lat], [z], 5.0, 4.0, skin=0.5, device=device)
    t_graph = [0.0]

    def iteration(force=None, split=False):
        pos = pos0 + (torch.rand(pos0.shape, generator=gen, device=device, dtype=torch.float64) - 0.5) * 0.05   # +-0.025 A, on the device
        if force == "no_wait":   # the skin test queued in front of the evaluation, its verdict read afterwards (VerletGraph.evaluate)
            vg.evaluate(model, pos, forces=True, extras=False)
            return
        if force in ("step", "step_refill"):
            vg.step(model, pos, force="refill" if force == "step_refill" else None)
            return
        if not split:            # what a trajectory loop does: graph at the new positions, then the step on it
            model(vg.update(pos, force=force), forces=True, extras=False)
            return
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = vg.update(pos, force=force)
        torch.cuda.synchronize()
        t_graph[0] += time.perf_counter() - t0
        model(g, forces=True, extras=False)

    def md_loop(force, reps=10):
        for _ in range(3):
            iteration(force)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            iteration(force)
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) / reps * 1e3
        if force in ("no_wait", "step", "step_refill"):
            return {"total": total}
        # the graph update on its own, from a second pass with a wait for the device on either side of it (the waits cost the
        # overlap of the queued topology build with the host's way to the engine call: that pass is slower than `total`)
        t_graph[0] = 0.0
        for _ in range(reps):
            iteration(force, split=True)
        torch.cuda.synchronize()
        return {"graph_update": t_graph[0] / reps * 1e3, "total": total}

    md = {}
    current = model.engine.precision
    for mode in dict.fromkeys((current, "f16x3")):
        model.engine.set_precision(mode)
        md[mode] = {"reuse": md_loop(None), "reuse_verdict_read_after_the_step": {"total": md_loop("no_wait", reps=20)["total"]},
                    "refill": md_loop("refill"), "rebuild": md_loop("search"),
                    # one library call per step (VerletGraph.step -> m3g_md_step)
                    "step_reuse": {"total": md_loop("step")["total"]}, "step_refill": {"total": md_loop("step_refill")["total"]}}
    model.engine.set_precision(current)
    md["paths_taken"] = dict(vg.stats)
    md["note"] = ("positions generated and kept on the device; `total`: model(vg.update(pos)) per iteration, jitter kernel and the wait "
                  "inside the skin-list test included; `graph_update`: the update alone, timed in a second pass between two waits")
    rec["md_iteration_ms_10k_atom_cell"] = md
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)   # (the clocks settle over the first dozen steps: 3 -> 30 warm-up steps is 1.5 % of the step)
    ap.add_argument("--workload", choices=("config3", "config4"), default="config3")
    ap.add_argument("--precision", choices=("f16x3", "fp32", "bf16x3"), default=DEFAULT_PRECISION, help="arithmetic of the headline figure")
    ap.add_argument("--cells", type=int, nargs=3, default=[10, 10, 25], help="config3: fcc cells per axis (4 atoms each)")
    ap.add_argument("--engine-option", action="append", default=[], metavar="NAME=INT",
                    help="engine option for A/B timing (m3g_plan_set_option), e.g. threebody_moments=0; recorded in config.engine_options")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other-precision and config4 secondary figures")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    job = Job(args)
    world, rank, device = job.world, job.rank, job.device
    from torch_m3gnet.data import MaterialGraphKey as K

    torch.set_num_threads(min(host_cores(), 16))
    model = default_model(device)
    model.engine.set_precision(args.precision)
    for opt in args.engine_option:
        name, _, val = opt.partition("=")
        model.engine.set_option(name, int(val))
    other_modes = [m for m in ("fp32", "f16x3", "bf16x3") if m != args.precision]
    out = {"metric": METRIC, "unit": "atom-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic", "dtype": DTYPE[args.precision]}

    if args.workload == "config4":
        rec = measure_config4(job, model, args.steps, args.warmup)
        out.update(value=rec["value"], ms_per_step=rec["ms_per_step"],
                   config={"workload": rec.pop("workload"), **{k: v for k, v in rec.items() if k not in ("value", "unit", "ms_per_step", "steps")}})
        if rank == 0:
            print(json.dumps(out), flush=True)
        job.close()
        return

    from torch_m3gnet.data.synthetic import fcc_cu_graph

    log("building workload graph on the host")
    graph = fcc_cu_graph(*args.cells, seed=rank).to(device)
    n_atoms = int(graph[K.POS].size(0))
    n_edges = int(graph[K.EDGE_INDEX].size(1))
    n_trip = int(graph[K.TRIPLET_EDGE_INDEX].size(1))
    def step():   # replicas: nothing crosses ranks inside a step (a single cell does not shard)
        model(graph, forces=True, extras=False)

    t_first = time.perf_counter()
    step()  # first call: plan commit + topology build (index-only, cached on the graph) + workspace allocation
    torch.cuda.synchronize()
    first_call_s = time.perf_counter() - t_first
    log(f"first call {first_call_s:.2f} s")
    elapsed = job.timed(step, args.steps, args.warmup)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n_atoms * args.steps / elapsed
    log(f"timed region done ({args.precision}): {ms_per_step:.3f} ms/step")
    replica_spread = None
    if job.dist is not None:   # after the timed region: the replicas' energies side by side (same cell, same weights: they must agree)
        energies_all = torch.empty(world, 1, device=job.comm_device)
        job.dist.all_gather_into_tensor(energies_all, graph[K.TOTAL_ENERGY].view(1, 1).to(job.comm_device))
        replica_spread = float(((energies_all - energies_all[0]).abs().max() / energies_all[0].abs().clamp_min(1e-30)).item())

    # index-only CSR build (m3g_topology_build), timed on its own: reused while the neighbour list is unchanged
    from torch_m3gnet.nn.modules import _Topology

    torch.cuda.synchronize()
    t1 = time.perf_counter()
    topo = _Topology(graph)
    torch.cuda.synchronize()
    topo_ms = (time.perf_counter() - t1) * 1e3
    n_active = topo.n_active()
    topo_hints = topo.query_hints()   # bit 0: complete partner lists -> the three-body moment kernels ran (include/m3gnet_hip.h)

    # PMC traffic is a property of a build: the profile set carries the digest of the kernel sources it was collected on
    # (tools/pmc_traffic.py); a set from other sources than the ones being timed yields `traffic: null`
    digest = csrc_digest()
    pmc_all, traffic_source = load_stamped(PMC_TRAFFIC_FILE, digest, args.cells)
    sq_all, sq_source = load_stamped(PMC_SQ_FILE, digest, args.cells)

    def record(precision, ms_step):
        """Roofline objects of one precision mode from live stage timers (the mode must be the engine's current one)."""
        per_launch = stage_times(model, lambda: model(graph, forces=True, extras=False), args.steps)
        views = rooflines(per_launch, n_atoms, n_edges, n_trip, n_active, precision, pmc_all.get(precision, {}),
                          moments=bool(topo_hints & 1) and "threebody_moments=0" not in args.engine_option
                          and "legendre_backward=1" not in args.engine_option)
        edge = {k: v for k, v in views.items() if k in EDGE_KERNELS}
        dom = max(edge, key=lambda k: edge[k]["avg_launch_ms"] * per_launch[k][1])   # dominant kernel = largest share of the step
        stage_ms = {k: round(ms * cnt, 4) for k, (ms, cnt) in per_launch.items()}
        views[dom]["traffic_source"] = traffic_source
        # vector instructions per MFMA and the share of SIMD cycles the vector ALU is active, from the SQ counter set of this build
        # (tools/collect_sq_counters.sh + tools/pmc_sq_json.py): SQ_ACTIVE_INST_VALU counts units of 4 cycles summed over SIMDs,
        # SQ_BUSY_CU_CYCLES cycles summed over CUs (4 SIMDs each) -> active share = ACTIVE_INST_VALU / BUSY_CU_CYCLES
        kname = views[dom]["kernel"].split(" ")[0]
        sq = sq_all.get(precision, {}).get(kname)
        views[dom]["sq_counters"] = ({"valu_per_mfma": sq["SQ_INSTS_VALU"] / max(sq["SQ_INSTS_MFMA"], 1.0),
                                      "valu_active_frac": sq["SQ_ACTIVE_INST_VALU"] / max(sq["SQ_BUSY_CU_CYCLES"], 1.0),
                                      "mfma_busy_frac": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4.0 * sq["SQ_BUSY_CU_CYCLES"], 1.0),
                                      "insts_valu_per_launch": sq["SQ_INSTS_VALU"], "insts_mfma_per_launch": sq["SQ_INSTS_MFMA"],
                                      "source": sq_source} if sq else {"valu_per_mfma": None, "valu_active_frac": None, "source": sq_source})
        pm = pmc_all.get(precision, {})
        # (kernels that run less than once per step belong to the topology build of the first call, not to the step)
        total = sum(r["launches_per_step"] * (2.0 * r["fetch_kb"] + r["write_kb"]) * 1024.0 for r in pm.values()
                    if r["launches_per_step"] >= 0.99) if pm else None
        ideal = BYTES_8D_PER_STEP(n_edges, n_trip, n_atoms)
        step_bytes = {"traffic": total, "algorithmic_bytes_8d": ideal, "traffic_over_algorithmic": (total / ideal) if total else None,
                      "source": traffic_source["file"] if total else None}
        # launches of ONE step as the timed region issues it: counted by the library from a stream capture of the un-profiled call
        # (m3g_count_launches; the stage profiler above runs a slightly different sequence -- no fused tail launches)
        launches = model.engine.count_launches(lambda: model(graph, forces=True, extras=False))
        return views[dom], [v for k, v in views.items() if k != dom], stage_ms, step_bytes, launches

    timing = series_and_clock(step, args.steps, sample_clock=rank == 0)
    roofline, others, stage_ms, step_bytes, launches = record(args.precision, ms_per_step)
    roofline["formula"] = ("achieved = algorithmic_flops (fp32) or executed_mfma_flops (split modes) / avg_launch_ms; frac = achieved / peak; "
                           "avg_launch_ms = HIP events around the kernel's launches on the launch stream, inside the library (m3g_profile_*), "
                           "which the rocprofv3 average of profiles/r06_<mode>_kernel_stats.csv must agree with")
    clk = (timing.get("clock_mhz") or {}).get("median")
    if clk:   # the peaks of MI355X_MICROARCH.md are quoted at 2,400 MHz; the card holds less under this load
        roofline["frac_at_measured_clock"] = roofline["frac"] * 2400.0 / clk
    out.update(value=value, ms_per_step=ms_per_step, ms_per_step_min=timing["ms_per_step_min"], ms_per_step_median=timing["ms_per_step_median"],
               clock_mhz=(timing.get("clock_mhz") or {}).get("median"), step_timing=timing, kernel_launches_per_step=launches[0],
               other_stream_operations_per_step=launches[1], roofline=roofline, roofline_other_kernels=others, step_traffic_bytes=step_bytes,
               config={"workload": f"single {n_atoms}-atom fcc Cu PBC supercell per GPU ({'x'.join(map(str, args.cells))} cells, "
                                   "a=3.61 A, jitter 0.025 A), r_cut 5 A / 3-body 4 A, default M3GNet (l_max=n_max=3, D=64, "
                                   "3 blocks), energy+forces+stress",
                       "precision": args.precision, **({"engine_options": args.engine_option} if args.engine_option else {}),
                       "atoms_per_gpu": n_atoms, "edges_per_gpu": n_edges, "triplets_per_gpu": n_trip,
                       "active_edges_per_gpu": n_active, "topology_hints": topo_hints, "kernel_launches_per_step": launches[0], "first_call_s_incl_topology_build": first_call_s,
                       "topology_build_ms": topo_ms, "stage_ms_per_step": stage_ms,
                       "multi_gpu": "replicas (a single cell does not shard; no collective inside the timed steps); config4_sharded below runs the sharded path",
                       **({"replica_energy_rel_spread": replica_spread} if replica_spread is not None else {})})
    if not args.no_secondary:
        for other in other_modes:   # the other arithmetic modes of the same engine, same workload, same timed-region rules
            model.engine.set_precision(other)
            step()
            el2 = job.timed(step, args.steps, args.warmup)
            t2 = series_and_clock(step, args.steps, sample_clock=rank == 0)
            r2, o2, st2, sb2, _ = record(other, el2 / args.steps * 1e3)
            out[other] = {"value": world * n_atoms * args.steps / el2, "unit": "atom-steps/s", "ms_per_step": el2 / args.steps * 1e3,
                          "ms_per_step_min": t2["ms_per_step_min"], "ms_per_step_median": t2["ms_per_step_median"],
                          "clock_mhz": (t2.get("clock_mhz") or {}).get("median"),
                          "dtype": DTYPE[other], "roofline": r2, "stage_ms_per_step": st2, "step_traffic_bytes": sb2}
            log(f"{other}: {el2 / args.steps * 1e3:.3f} ms/step")
        model.engine.set_precision(args.precision)
        out["config4_sharded"] = measure_config4(job, model, max(5, args.steps // 2), 2)
        log(f"config4_sharded: {out['config4_sharded']['ms_per_step']:.3f} ms/step")
        if world == 1 and tuple(args.cells) == (10, 10, 25):
            out["beside"] = measure_beside(model, device)
            log(f"beside: {out['beside']}")
        # the secondary figures as SCALAR keys directly under `config` (the driver's record keeps the scalar keys of `config` and drops
        # nested objects; the full sub-records stay in the top-level keys above)
        bs = out.get("beside", {})
        md = bs.get("md_iteration_ms_10k_atom_cell", {}).get(args.precision, {})
        sec = {**{f"sec_{m}_ms": out[m]["ms_per_step"] for m in other_modes if m in out},
               "sec_config4_ms": out["config4_sharded"]["ms_per_step"]}
        for key, short in (("step_ms_32_atom_cu_cell", "sec_step_ms_32_atom"), ("step_ms_864_atom_cu_cell", "sec_step_ms_864_atom"),
                           ("step_ms_2048_atom_cu_cell", "sec_step_ms_2048_atom"), ("step_ms_4000_atom_cu_cell", "sec_step_ms_4000_atom"),
                           ("md_iteration_ms_32_atom_cell", "sec_md_ms_32_atom"), ("md_iteration_ms_864_atom_cell", "sec_md_ms_864_atom"),
                           ("md_refill_iteration_ms_32_atom_cell", "sec_md_refill_ms_32_atom"),
                           ("md_refill_iteration_ms_864_atom_cell", "sec_md_refill_ms_864_atom"),
                           ("md_step_refill_iteration_ms_32_atom_cell", "sec_md_step_refill_ms_32_atom"),
                           ("md_step_refill_iteration_ms_864_atom_cell", "sec_md_step_refill_ms_864_atom"),
                           ("step_ms_config5_r3_4A", "sec_config5_ms"), ("step_ms_config5_r3_6A", "sec_config5_r3_6A_ms")):
            if key in bs:
                sec[short] = bs[key]
        for k, short in (("reuse", "sec_md_10k_reuse_ms"), ("refill", "sec_md_10k_refill_ms"), ("rebuild", "sec_md_10k_search_ms"),
                         ("step_reuse", "sec_md_10k_step_reuse_ms"), ("step_refill", "sec_md_10k_step_refill_ms"),
                         ("reuse_verdict_read_after_the_step", "sec_md_10k_reuse_no_wait_ms")):
            if k in md:
                sec[short] = md[k]["total"]
        out["config"].update({k: round(float(v), 4) for k, v in sec.items()})
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tuple(args.cells))
            out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]   # (vs_baseline stays null: nothing published)
            out["config"]["sec_cpu_all_cores_ms"] = round(out["cpu_baseline"]["ms_per_step"], 1)
            out["config"]["sec_cpu_cores"] = out["cpu_baseline"]["cores"]
        print(json.dumps(out), flush=True)
    job.close()


if __name__ == "__main__":
    main()
