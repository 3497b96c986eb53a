#!/usr/bin/env python3
"""Benchmark of the ATM-VFI forward hot path on MI355X (contract: see the task brief).

Workload (BASELINE.json `metric`): network_base, 1080p frames replicate-padded to 1088x1920,
batch 1 per GPU, global + local branches on, fp32.  One step = one ``Network.forward`` on one
synthetic frame pair that is already resident in HBM, producing one interpolated frame.
With N GPUs every rank interpolates its own pair per step (frame-batch sharding, weak scaling)
and the N output frames are all-gathered over RCCL/xGMI -- the only collective on the path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  roofline     : the kernel family with the largest share of the forward (today conv3x3_planes_kernel),
                 achieved = sum of algorithmic FLOPs / sum of launch durations, measured live with HIP
                 events on the launch stream in a separate instrumented pass after the timed region;
                 `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), reported only
                 while the kernel sources hash to the build those passes were taken on;
  cpu_baseline : the CPU oracle (oracle/atmvfi_oracle.py, a port of the reference's algorithm) timed on this
                 node's host cores (cgroup quota) on a bounded sample, 1 warm-up + median of 3 (rank 0, N=1 only);
  kernels      : per-kernel-family time split of one forward (ms), for DESIGN.md / profiles/.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

_REAL_STDOUT = sys.stdout          # main() replaces it by a private copy of file descriptor 1 (see there)

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FLOPS_PER_PAIR = {  # SURVEY.md section 8d, algorithmic FLOPs per frame pair (2 FLOP/MAC)
    ("base", 1088, 1920, True): 5925.0e9,
    ("base", 576, 960, True): 1566.3e9,
    ("lite", 256, 448, False): 87.8e9,
    ("lite", 256, 256, True): 56.2e9,
    ("base", 2176, 4096, True): 25300.0e9,
}
# BASELINE.json `configs`, in order: (variant, height, width, global branch on, description)
CONFIGS = {
    "c1": ("lite", 256, 256, True, "network_lite 256x256 (configs[0], the reference's CPU-runnable case)"),
    "c2": ("lite", 256, 448, False, "network_lite 256x448 Vimeo90K shape, global off (configs[1])"),
    "c3": ("base", 540, 960, True, "network_base 540x960 (configs[2])"),
    "c4": ("base", 1080, 1920, True, "network_base 1080x1920, one pair per GPU (configs[3]; the headline metric)"),
    "c5": ("base", 2160, 4096, True, "network_base 2160x4096, one pair per GPU (configs[4], run with --gpus 4)"),
}
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, dense fp32 matrix peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense fp16/bf16 matrix peak (no sparsity)
PEAK_HBM_GBS = 8000.0


def csrc_digest() -> str:
    """sha256 over everything that determines the library (tools/source_digest.py: kernel sources, headers, generated includes, the
    Makefile, the C-ABI header) -- the value the Makefile bakes into it (``atmvfi_source_digest()``); stamps the committed PMC traffic
    figures with the build they were measured on."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import source_digest
        return source_digest.digest()
    finally:
        sys.path.pop(0)


def host_cores():
    """(threads to use, description): the cgroup CPU quota if there is one, else the affinity mask; plus the CPU model string."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, int(q / per + 0.5))
        except Exception:
            pass
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    use = min(n, quota) if quota else n
    return use, f"{model}; {n} logical CPUs visible, cgroup quota {quota if quota else 'none'}"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="a BASELINE.json configuration by name (sets --variant/--height/--width/--global-off): " +
                         "; ".join(f"{k} = {v[4]}" for k, v in sorted(CONFIGS.items())) + ".  Default: c4, the headline metric")
    ap.add_argument("--variant", default="base", choices=["base", "lite"])
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--global-off", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher self-test (tests/test_host_logic.py): the ranks rendezvous over gloo on the CPU and rank 0 prints a line; "
                         "no GPU, no model, nothing is measured")
    ap.add_argument("--cpu-baseline-crop", action="store_true",
                    help="time the CPU oracle on a quarter-pixel crop and scale by the pixel ratio (marked extrapolated) instead of the full frame pair")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip timing BASELINE.json's other configurations (c1, c2, c3, c5) after the headline one")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f32"])
    ap.add_argument("--no-plans", action="store_true", help="direct launches (one C-ABI call per op) instead of launch-plan replay: the A/B of tools/small_configs.sh")
    ap.add_argument("--no-host-io", action="store_true", help="skip the PCIe-inclusive measurement (uint8 frames in host memory in and out)")
    ap.add_argument("--gather-u8", action="store_true", help="N > 1: all-gather the output frames rounded to uint8 (4x fewer bytes over xGMI)")
    ap.add_argument("--graph", action="store_true", help="replay the forward from a captured HIP graph (Network.enable_graphs); measured "
                    "within 0.5 %% of eager launches at 1080p and at 256x448: the stream is already back to back")
    args = ap.parse_args()
    if args.config:
        args.variant, args.height, args.width, g_on, _ = CONFIGS[args.config]
        args.global_off = not g_on
    return args


def time_config(pkg, host_io, pairs, dev, cname, steps, warmup, precision, shared=None, ensemble=False, inflight=()):
    """One BASELINE.json configuration on one GPU, timed like the headline: resident synthetic pairs, ``warmup`` untimed forwards (they
    also record the launch plan), ``steps`` timed ones between synchronisations.  -> the entry of the bench line's ``configs`` block."""
    variant, height, width, g_on, desc = CONFIGS[cname]
    if shared is not None:
        net, _ = shared
    else:
        sd = pkg.synthetic_state_dict(variant, seed=1)
        net = (pkg.NetworkBase if variant == "base" else pkg.NetworkLite)()
        net.load_state_dict(sd, strict=True)
        net.to(dev).eval()
        net.set_precision(precision)
    net.global_motion = g_on
    net.ensemble_global_motion = bool(ensemble)          # multiscale_global_motion_ensemble (network_base.py:564-605): three input scales
    padder = host_io.InputPadder((1, 3, height, width), divisor=64)
    frames = []
    for i in range(2):
        a, b = pairs.random_pair(1, height, width, seed=2000 + i)
        a, b = padder.pad(a.to(dev), b.to(dev))
        frames.append((a.contiguous(), b.contiguous()))
    H, W = frames[0][0].shape[-2:]
    for i in range(warmup):
        net(*frames[i & 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        net(*frames[i & 1])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    fps = steps / el
    net.ensemble_global_motion = False
    out = {"workload": desc + f" (padded {H}x{W})" + (", ensemble_global_motion on" if ensemble else ""), "value": round(fps, 3), "unit": "frames/s", "ms_per_step": round(1e3 * el / steps, 4),
           "steps": steps, "warmup": warmup}
    fl = None if ensemble else FLOPS_PER_PAIR.get((variant, H, W, g_on))
    if fl:
        out["forward_tflops"] = round(fl * fps / 1e12, 2)
        out["forward_frac_of_f16x3_peak"] = round(fl * fps / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3.0), 4)
    if inflight:
        # K independent forwards in flight (host_io.PairStreams: K replicas -- shared weights, own workspace + launch plan -- on K
        # streams, pairs round-robin, no cross-stream event inside a forward; outputs bit-identical to the single-stream forward:
        # tests/test_gpu_e2e.py::test_pair_streams_equal_single_stream).  Reported BESIDE the single-stream value above, never
        # instead of it.  Timed like it: resident pairs, the same number of forwards, one synchronisation at each end; every forward
        # returns fresh output tensors.  (`record_outputs=False`: nothing here consumes the outputs on another stream, so the
        # allocator's cross-stream bookkeeping -- ~25 record_stream calls per forward -- is not needed; `..._with_bookkeeping` has it.)
        kin = {}
        for k in inflight:
            with host_io.PairStreams(net, k) as ps:
                # every replica: two eager forwards, the recording one (+ its self-check replay), two replays -- all before the clock
                for _ in ps.map((frames[i & 1] for i in range(6 * k)), wait_inputs=False, record_outputs=False):
                    pass
                ps.synchronize()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = sum(1 for _ in ps.map((frames[i & 1] for i in range(3 * steps)), wait_inputs=False, record_outputs=False))
                ps.synchronize()
                kin[str(k)] = round(n / (time.perf_counter() - t0), 2)
                if k == inflight[-1] and k > 2:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n = sum(1 for _ in ps.map((frames[i & 1] for i in range(3 * steps))))
                    ps.synchronize()
                    out["frames_per_s_k_inflight_with_bookkeeping"] = {str(k): round(n / (time.perf_counter() - t0), 2)}
            torch.cuda.empty_cache()
        out["frames_per_s_k_inflight"] = kin
        if fl:
            best = max(kin.values())
            out["k_inflight_frac_of_f16x3_peak"] = round(fl * best / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3.0), 4)
    net.release_workspace()
    del frames
    torch.cuda.empty_cache()
    return out


def self_launch(args) -> int:
    """``python bench.py --gpus N`` (N > 1) outside a launcher: start the N ranks as FRESH child processes through
    ``python -m torch.distributed.run`` on 127.0.0.1 with a free port -- before this process has made any GPU call (it never
    makes one: ``import torch`` does not initialise HIP) and without exec'ing -- relay rank 0's JSON line to stdout, everything
    else the children print to stderr, and return the launcher's exit code (non-zero if any rank failed)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    # host budget per rank: the ranks' CPU work is launching (one thread each); OpenMP pools of every rank's
    # torch are capped so that N ranks never oversubscribe the node's CPU quota
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, host_cores()[0] // max(1, args.gpus)))))
    env["MASTER_ADDR"], env["MASTER_PORT"] = "127.0.0.1", str(port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in child.stdout:
        is_result = line.startswith("{") and '"metric"' in line
        print(line, end="", file=sys.stdout if is_result else sys.stderr, flush=True)
    return child.wait()


def dry_run(args, dist):
    """The rendezvous and reporting skeleton of ``main`` without a GPU: what the CPU test of the launcher branch runs."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if os.environ.get("ATMVFI_BENCH_DRY_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    group_world, per_rank = 1, [0.0]
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        # the reporting collective of main(): every rank's elapsed time into one [world, 1] tensor
        allt = torch.empty(dist.get_world_size(), 1, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, torch.tensor([[float(rank)]], dtype=torch.float64))
        per_rank, group_world = allt[:, 0].tolist(), dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    if os.environ.get("ATMVFI_BENCH_DRY_NOISE") == "1":                  # (launcher self-test) what RCCL's version banner does
        os.write(1, b"noise written to file descriptor 1 by a native library\n")
    print(f"rank {rank} of {world}: rendezvous at {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')} ok", flush=True)
    if rank == 0:
        print(json.dumps({"metric": "dry-run (launcher self-test, nothing measured)", "value": None, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "collective_world_size": group_world, "per_rank_ms_per_step": per_rank}),
              file=_REAL_STDOUT, flush=True)


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (read when the runtime starts): RCCL's intra-node transport needs it
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    # Rank 0's stdout carries ONE JSON line and nothing else.  Native libraries write there too (RCCL prints a five-line version banner to
    # file descriptor 1 when its communicator starts), so descriptor 1 is pointed at stderr for the run and the line goes out through a saved
    # copy of the real stdout at the end.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch.distributed as dist

    if args.dry_run:
        return dry_run(args, dist)
    import pairs
    pkg = importlib.import_module("atm-vfi_amd")
    host_io = importlib.import_module("atm-vfi_amd.host_io")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.set_grad_enabled(False)
    # rehearsal of the N > 1 code path on a one-GPU box: ATMVFI_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo
    # (RCCL refuses two ranks on one device); never set by the driver
    rehearsal = os.environ.get("ATMVFI_BENCH_REHEARSAL") == "1"
    dev = torch.device("cuda:0" if rehearsal else f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    # ATMVFI_BENCH_FORCE_COLLECTIVE=1 (never set by the driver): take the collective code path with ONE rank too, so the RCCL calls
    # (communicator init, async all_gather into a tensor list, barrier, all_reduce) can be exercised on a one-GPU box
    collective = world > 1 or os.environ.get("ATMVFI_BENCH_FORCE_COLLECTIVE") == "1"
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        tmo = datetime.timedelta(seconds=600)      # a stalled rank fails the run instead of hanging it
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)   # "nccl" is RCCL on ROCm

    # ---- model + synthetic inputs (resident in HBM before the timed region) ----
    variant = args.variant
    sd = pkg.synthetic_state_dict(variant, seed=1)
    net = (pkg.NetworkBase if variant == "base" else pkg.NetworkLite)()
    net.load_state_dict(sd, strict=True)
    net.to(dev).eval()
    net.global_motion = not args.global_off
    padder = host_io.InputPadder((1, 3, args.height, args.width), divisor=64)
    n_in = 4
    frames = []
    for i in range(n_in):
        a, b = pairs.random_pair(1, args.height, args.width, seed=1000 + 17 * rank + i)
        a, b = padder.pad(a.to(dev), b.to(dev))
        frames.append((a.contiguous(), b.contiguous()))
    H, W = frames[0][0].shape[-2:]

    # The collective is sharding.PipelinedGather -- the class the gloo world-size-2 tests exercise (tests/test_sharding_gloo.py):
    # the all-gather of step k (RCCL, its own stream) is issued from a send buffer of its own, runs under the forward of step k+1
    # and is waited for one step later, so a rank that is momentarily slower delays the others' *gather*, not their next forward,
    # and a captured graph may overwrite its static output at once.  Every step's collective is issued and completed inside the
    # timed region (the last one is waited for before the closing synchronize).  --gather-u8 sends the frame rounded to uint8.
    sharding = importlib.import_module("atm-vfi_amd.sharding")
    gather = None
    if collective:
        if args.gather_u8:
            u8 = torch.empty(H, W, 3, dtype=torch.uint8, device=dev)

            def enc(x):                      # fp32 [1,3,H,W] -> uint8 [H,W,3] (np.round(x * 255)), one HIP kernel
                net._ops_obj.frame_f32_to_u8(x[0], u8, 0, 0, False)
                return u8
            gather = sharding.PipelinedGather(world, (H, W, 3), dev, torch.uint8, encode=enc)
        else:
            gather = sharding.PipelinedGather(world, (1, 3, H, W), dev, torch.float32)

    def drain():
        if gather is not None:
            gather.drain()

    def step(i):
        a, b = frames[i % n_in]
        out = net(a, b)["I_t"]
        if gather is not None:
            gather.submit(out)                  # per-rank output frames only; returns step i-1's frames
        return out

    net.set_precision(args.precision)
    if args.no_plans:
        net.enable_plans(False)
    net.enable_graphs(args.graph)             # one hipGraphLaunch per forward; the collective stays outside the graph
    # Set-up, before the W warm-up steps: forwards until this shape's launch plan is recorded and has replayed once (workspace, window
    # maps, packed weights, kernel attributes, the plan's record-time self-check, first-use allocations of the output tensors) -- so
    # that the W warm-up steps and the K timed steps are all the steady-state step, whatever W is.
    for i in range(5):
        step(i)
    drain()
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    if collective:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # per-step GPU time (evidence; not the metric)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step(i)
        marks[i + 1].record()
    drain()
    torch.cuda.synchronize()
    if collective:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank_ms = None
    if collective:
        # every rank's own elapsed time (one small all-gather): the line reports the MAX (the contract) and the list, so a reader sees
        # that the communicator had N ranks and whether they were balanced
        mine = torch.tensor([[elapsed]], device=dev, dtype=torch.float64)
        allt = torch.empty(dist.get_world_size(), 1, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, mine)
        per_rank_ms = [round(1e3 * float(x) / args.steps, 3) for x in allt[:, 0].tolist()]
        elapsed = float(allt.max().item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        fps = world * args.steps / elapsed
        key = (variant, H, W, net.global_motion)
        flops = FLOPS_PER_PAIR.get(key)
        result = {
            "metric": "interpolated frames/s at 1080p (network_base, bs=1)" if key == ("base", 1088, 1920, True)
            else f"interpolated frames/s ({variant} {args.height}x{args.width})",
            "value": round(fps, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "per_step_gpu_ms": [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(args.steps)],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (contractions as fp16 hi/lo split x3 MFMA, fp32 accumulate; ~22 significand bits)" if getattr(net._ops_obj, "precision", "") == "f16x3" else "f32",
            "data": "synthetic",
            "config": {"workload": f"network_{variant} {args.height}x{args.width} (padded {H}x{W}) bs=1 per GPU, "
                                   f"global {'on' if net.global_motion else 'off'}, fp32, random frame pairs, stress weights seed 1",
                       "pairs_per_step": world, "parallelism": f"frame-batch dp{world} + all-gather of the {'uint8' if args.gather_u8 else 'fp32'} output frames, one step behind the forward (sharding.PipelinedGather)" if collective else "single GPU"},
        }
        if args.graph:
            result["launch"] = "HIP graph replay (captured forward, inputs copied into its static buffers inside the timed region)"
        elif getattr(net, "use_plans", False) and getattr(net, "_plans", None):
            result["launch"] = "launch plan (the recorded forward replayed by one atmvfi_plan_run call per step: the same ~120 kernel launches, fresh output tensors)"
        else:
            result["launch"] = "eager (one C-ABI call per op)"
        if flops:
            result["forward_tflops"] = round(flops * fps / world / 1e12, 2)     # per-GPU algorithmic rate of the whole forward
            result["forward_frac_of_f32_mfma_peak"] = round(flops * fps / world / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
            result["forward_frac_of_f16x3_peak"] = round(flops * fps / world / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3.0), 4)
        # ---- instrumented pass: per-launch HIP events on the launch stream ----
        if not args.no_profile:
            ops = net._ops_obj
            net.enable_graphs(False)      # per-launch events need the individual launches
            for rep in range(2):          # second pass is the one reported (first warms the event pool)
                ops.profile = []
                net(*frames[0])           # forward only: the other ranks are not in this pass, so no collective here
                torch.cuda.synchronize()
                prof = ops.profile
                ops.profile = None
            agg = {}
            for name, meta, s, e in prof:
                d = agg.setdefault(name, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0})
                d["ms"] += s.elapsed_time(e)
                d["launches"] += 1
                d["flops"] += meta.get("flops", 0.0)
                d["bytes"] += meta.get("bytes", 0.0)
            total_ms = sum(d["ms"] for d in agg.values())
            # kernel families and the MFMA peak that bounds them: the f16x3 engines issue three
            # 16-bit MFMAs per algorithmic multiply-add, so their algorithmic peak is 2500/3 TFLOP/s.
            families = {
                "conv3x3_planes_kernel (3x3 s1 convs on split-plane input, LDS-DMA halo, ping-pong wave groups)": (["conv3x3_planes"], PEAK_F16_MFMA_TFLOPS / 3.0),
                "conv3x3_f16x3_row_kernel": (["conv3x3_f16x3"], PEAK_F16_MFMA_TFLOPS / 3.0),
                "stem_kernel (feat_extracts.0.0 -> 0.1 -> 1.0 in one launch, the full-resolution maps in LDS)": (["stem_fused"], PEAK_F16_MFMA_TFLOPS / 3.0),
                "gemm_f16x3_kernel (fp32-input rows: strided conv2d)": (["linear_f16x3", "deconv2x2_f16x3", "conv2d_f16x3"], PEAK_F16_MFMA_TFLOPS / 3.0),
                "gemm_pp_kernel + gemm_duo_kernel (nn.Linear / ConvTranspose2d / strided Conv2d rows from split planes by LDS-DMA: 256x128 tiles, ping-pong wave groups, persistent grid; 128x128 tiles, two workgroups per CU, where that grid would be under-filled)": (["linear_split", "deconv2x2_split", "conv2d_split"], PEAK_F16_MFMA_TFLOPS / 3.0),
                "gemm_mfma_f32 (exact-fp32 engine)": (["conv2d", "linear", "deconv2x2"], PEAK_F32_MFMA_TFLOPS),
            }
            fam_out = {}
            for fam, (names, peak) in families.items():
                ms = sum(agg[n]["ms"] for n in names if n in agg)
                if ms <= 0:
                    continue
                fl = sum(agg[n]["flops"] for n in names if n in agg)
                nl = sum(agg[n]["launches"] for n in names if n in agg)
                ach = fl / (ms * 1e-3) / 1e12
                fam_out[fam] = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                                "frac": round(ach / peak, 4), "traffic": None, "kernel": fam, "launches": nl,
                                "avg_launch_ms": round(ms / nl, 4), "algorithmic_gflop_per_forward": round(fl / 1e9, 1),
                                "share_of_forward_time": round(ms / total_ms, 4)}
            # The plane-input GEMM launches that their BYTES bound, not their flops (short K, wide scatter epilogues: the last decoder
            # deconvs, the refiner's first strided convs): algorithmic bytes / 8 TB/s exceeds algorithmic flops / 833 TFLOP/s.  They get an
            # HBM entry of their own (a traffic-bound launch graded against MFMA says nothing); the family entry
            # above keeps every launch and also says what the MFMA-bound ones alone reach.
            gemm_names = ("linear_split", "deconv2x2_split", "conv2d_split")
            hb = [(n, m, s.elapsed_time(e)) for n, m, s, e in prof if n in gemm_names and m.get("bytes", 0.0) / (PEAK_HBM_GBS * 1e9) > m.get("flops", 0.0) / (PEAK_F16_MFMA_TFLOPS / 3.0 * 1e12)]
            if hb:
                ms_h = sum(t for _, _, t in hb)
                by_h = sum(m.get("bytes", 0.0) for _, m, _ in hb)
                fl_h = sum(m.get("flops", 0.0) for _, m, _ in hb)
                ach = by_h / (ms_h * 1e-3) / 1e9
                kh = "gemm_pp_kernel / gemm_duo_kernel launches bounded by their bytes (algorithmic bytes / 8 TB/s > algorithmic flops / 833 TFLOP/s)"
                fam_out[kh] = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                               "frac_of_achievable_6300": round(ach / 6300.0, 4), "traffic": None, "kernel": kh, "launches": len(hb),
                               "avg_launch_ms": round(ms_h / len(hb), 4), "algorithmic_gbytes_per_forward": round(by_h / 1e9, 3),
                               "mfma_tflops_of_these_launches": round(fl_h / (ms_h * 1e-3) / 1e12, 1),
                               "shapes": sorted({m.get("shape", "?") for _, m, _ in hb}), "share_of_forward_time": round(ms_h / total_ms, 4)}
                for fam in fam_out:
                    if fam.startswith("gemm_pp_kernel + gemm_duo_kernel"):
                        ms_all = sum(agg[n]["ms"] for n in gemm_names if n in agg)
                        fl_all = sum(agg[n]["flops"] for n in gemm_names if n in agg)
                        if ms_all > ms_h:
                            fam_out[fam]["frac_without_the_byte_bound_launches"] = round((fl_all - fl_h) / ((ms_all - ms_h) * 1e-3) / 1e12 / (PEAK_F16_MFMA_TFLOPS / 3.0), 4)
            # HBM traffic per launch: PMC counters cannot be read from inside this process (rocprofv3 wraps the run), so the value
            # comes from the committed passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled per the
            # gfx950 correction; tools/pmc_hbm.sh) -- and only while the kernel sources are the ones those passes were taken on
            # (sha256 over atm-vfi_amd/csrc, stamped into the file); otherwise `traffic` stays null rather than going stale.
            try:
                if key == ("base", 1088, 1920, True):
                    import glob
                    digest = csrc_digest()
                    pmc_all, pmc_file = None, None
                    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.json")), reverse=True):
                        try:
                            j = json.load(open(cand))
                        except Exception:
                            continue
                        if j.get("csrc_sha256") == digest:
                            pmc_all, pmc_file = j, os.path.basename(cand)
                            break
                    for fam in fam_out:
                        # the family's kernels as rocprofv3 names them: the words ending in "_kernel" in front of the parenthesis
                        kns = [w for w in fam.split(" (")[0].split(" ") if w.endswith("_kernel")]
                        rows = [pmc_all["per_forward"][k] for k in kns if pmc_all is not None and k in pmc_all["per_forward"]]
                        if rows:
                            fam_out[fam]["traffic"] = round(sum(r["traffic_GB_per_launch"] * r["launches"] for r in rows) /
                                                            sum(r["launches"] for r in rows) * 1e9)
                            # measured HBM bytes (PMC) over the algorithmic bytes of the same launches ("each layer reads its
                            # input once and writes its output once"): > 1 = re-reads (halo, column blocks, L2 misses)
                            alg = sum(agg[n]["bytes"] for n in families.get(fam, ([], 0))[0] if n in agg)
                            if alg > 0:
                                fam_out[fam]["algorithmic_bytes_per_forward"] = round(alg)
                                fam_out[fam]["traffic_ratio"] = round(sum(r["traffic_GB_per_launch"] * r["launches"] for r in rows) * 1e9 / alg, 3)
                            fam_out[fam]["traffic_unit"] = f"bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/{pmc_file}, taken on this build: same source digest)"
                        else:
                            fam_out[fam]["traffic_unit"] = "null: no committed PMC pass was taken on this build (source digest differs)"
            except Exception:
                pass
            if fam_out:
                dom = max(fam_out.values(), key=lambda d: d["share_of_forward_time"])
                result["roofline"] = dom
                result["roofline_all_families"] = fam_out
            result["kernels"] = {k: {"ms": round(d["ms"], 3), "launches": d["launches"],
                                     **({"tflops": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2)} if d["flops"] and d["ms"] > 0 else {}),
                                     **({"gbs": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1)} if d["bytes"] and d["ms"] > 0 else {})}
                                 for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
            result["kernels"]["_sum_ms"] = round(total_ms, 3)
        # ---- PCIe-inclusive rate (never `value`): uint8 frames in host memory -> uint8 frame in host memory through
        #      host_io.FramePipeline (pinned slots, H2D / D2H on side streams under the previous / next forward) ----
        if world == 1 and not args.no_host_io:
            rng = np.random.default_rng(0)
            u8 = [rng.integers(0, 256, (args.height, args.width, 3), dtype=np.uint8) for _ in range(3)]
            pairs_u8 = [(u8[i % 3], u8[(i + 1) % 3]) for i in range(max(args.steps, 30))]      # long enough to amortise fill and drain
            hio = {}
            for depth, nstreams in ((1, 1), (3, 1), (3, 2)):
                pipe = host_io.FramePipeline(net, args.height, args.width, isBGR=True, divisor=64, depth=depth, streams=nstreams)
                for _ in pipe.run(pairs_u8[:2 if nstreams == 1 else 10]):      # (with streams: every replica builds its workspace and plan)
                    pass
                torch.cuda.synchronize()
                tp = time.perf_counter()
                n_out = sum(1 for _ in pipe.run(pairs_u8))
                torch.cuda.synchronize()
                tp = time.perf_counter() - tp
                hio["sequential" if depth == 1 else "overlapped_depth3" if nstreams == 1 else "overlapped_depth3_two_forwards_in_flight"] = round(n_out / tp, 3)
                if pipe.lanes is not None:
                    pipe.lanes.release()
                del pipe
                torch.cuda.empty_cache()
            result["host_io"] = {"unit": "frames/s, uint8 HWC BGR frames in pageable host memory in and out (pre/post kernels, pinned staging, PCIe both ways)",
                                 **hio, "bytes_over_pcie_per_frame": 3 * args.height * args.width * 3}
        # ---- the other BASELINE.json configurations, timed the same way (default run only: N = 1, config c4) ----
        if world == 1 and key == ("base", 1088, 1920, True) and not args.no_configs:
            net.release_workspace()
            torch.cuda.empty_cache()
            result["configs"] = {"c4": {"value": result["value"], "ms_per_step": result["ms_per_step"], "steps": args.steps,
                                        "forward_frac_of_f16x3_peak": result.get("forward_frac_of_f16x3_peak")}}
            for cname, csteps, cwarm in (("c1", 200, 20), ("c2", 200, 20), ("c3", 60, 6), ("c5", 4, 3)):
                result["configs"][cname] = time_config(pkg, host_io, pairs, dev, cname, csteps, cwarm, args.precision,
                                                       shared=(net, sd) if CONFIGS[cname][0] == variant else None,
                                                       inflight=(2, 3, 4) if cname in ("c1", "c2", "c3") else ())
            # two forwards in flight at the headline size too (two 13 GB workspaces): kernel tails and tile-boundary gaps of one forward
            # under the other's kernels.  Beside the single-stream headline, never instead of it.
            c4k = time_config(pkg, host_io, pairs, dev, "c4", args.steps, 3, args.precision, shared=(net, sd), inflight=(2,))
            result["configs"]["c4"]["frames_per_s_k_inflight"] = c4k.get("frames_per_s_k_inflight")
            result["configs"]["c4"]["k_inflight_frac_of_f16x3_peak"] = c4k.get("k_inflight_frac_of_f16x3_peak")
            # configs[4] as BASELINE words it -- "2160x4096 TILED": host_io.forward_tiled (four 1144 x 2112 tiles = 1080 x 2048 cores + 64 px
            # of context, padded to 1152 x 2112; the build-defined mode whose parity oracle is the oracle on the same tiles,
            # tests/test_gpu_e2e.py::test_tiled_inference_vs_oracle_on_identical_tiles), one tile at a time and with two in flight
            try:
                a5, b5 = (t.to(dev) for t in pairs.random_pair(1, 2160, 4096, seed=2005))
                tiled = {}
                for k in (1, 2):
                    fwd = net if k == 1 else host_io.PairStreams(net, k)       # kept across calls: the replicas' workspaces and plans
                    for _ in range(4):
                        host_io.forward_tiled(fwd, a5, b5, tile=(1088, 2048), overlap=64)
                    torch.cuda.synchronize()
                    t5 = time.perf_counter()
                    for _ in range(4):
                        host_io.forward_tiled(fwd, a5, b5, tile=(1088, 2048), overlap=64)
                    torch.cuda.synchronize()
                    tiled[str(k)] = round(4.0 / (time.perf_counter() - t5), 3)
                    if k > 1:
                        fwd.release()
                    net.release_workspace()
                    torch.cuda.empty_cache()
                result["configs"]["c5_tiled"] = {"workload": "network_base 2160x4096, host_io.forward_tiled: 4 tiles of 1144x2112 (1080x2048 cores, 64 px overlap) padded to 1152x2112, stitched on the device",
                                                 "value": tiled["1"], "unit": "frames/s", "frames_per_s_k_inflight": {"2": tiled["2"]},
                                                 "steps": 4, "warmup": 4}
                del a5, b5
            except Exception as e:          # never lose the headline line to an auxiliary configuration
                result["configs"]["c5_tiled"] = {"error": repr(e)}
            # the large-motion mode of the API on the c3 frame size: planned like every other mode since its pick moved
            # into the C ABI (atmvfi_ensemble_select)
            result["configs"]["c3_ensemble"] = time_config(pkg, host_io, pairs, dev, "c3", 40, 6, args.precision, shared=(net, sd), ensemble=True)
            net.global_motion = not args.global_off
        # ---- CPU baseline: the oracle on this node's host cores, bounded sample ----
        if world == 1 and not args.no_cpu_baseline:
            # threads = the cores this process may really use (cgroup quota), CPU model stated, one warm-up,
            # median of three, on the timed workload itself: the same padded frame pair, full size (~12 s per forward at 1088x1920
            # on 16 threads).  --cpu-baseline-crop: a quarter-pixel crop scaled by the pixel ratio instead, marked as extrapolated.
            from oracle import atmvfi_oracle as O
            ncore, cpu_desc = host_cores()
            torch.set_num_threads(ncore)
            a, b = frames[0]
            if args.cpu_baseline_crop:
                sh, sw = (H // 2 + 15) // 16 * 16, (W // 2 + 15) // 16 * 16
            else:
                sh, sw = H, W
            # the HIP path's result on exactly that pair (frames[0] of the timed set), for the `parity` block below
            g_out = net(a[..., :sh, :sw].contiguous(), b[..., :sh, :sw].contiguous())
            torch.cuda.synchronize()
            g_keys = ("I_t", "opt_flow_0", "opt_flow_1", "occ_mask1", "I_t_0", "I_t_1")
            g_cpu = {k: g_out[k].cpu() for k in g_keys}
            g_lists = [t.cpu() for t in g_out["im_t_list"]]
            del g_out
            a, b = a[..., :sh, :sw].cpu().contiguous(), b[..., :sh, :sw].cpu().contiguous()
            runs = []
            o_out = None
            for rep in range(4):
                tc = time.perf_counter()
                o = O.forward(sd, a, b, global_motion=net.global_motion)
                runs.append(time.perf_counter() - tc)
                if o_out is None:
                    o_out = o                # the forwards are deterministic: the first one's outputs are the checker's values
                del o
            med = float(np.median(runs[1:]))
            scale = (H * W) / float(sh * sw)
            what = (f"a {sh}x{sw} crop of the same frame pair (1/{scale:.2f} of the {H}x{W} pixels), scaled by the pixel ratio" if scale != 1.0
                    else f"the timed workload itself: the same {H}x{W} frame pair, full size")
            result["cpu_baseline"] = {"value": round(1.0 / (med * scale), 5), "unit": "frames/s", "cores": ncore, "kind": "port",
                                      "cpu": cpu_desc, "extrapolated": scale != 1.0, "scale_factor": round(scale, 4),
                                      "sample": f"oracle forward (oracle/atmvfi_oracle.py) on {what}; {ncore} threads, 1 warm-up + median of 3: "
                                                f"{med:.2f} s per forward (runs {', '.join(f'{t:.2f}' for t in runs)})"}
            # ---- parity ON THE TIMED WORKLOAD: the HIP forward against the oracle forward that was just timed, same pair, every pixel
            #      (north star: |d| <= 1e-3 per pixel on fp32 I_t; flows 2e-3 px as in tests/test_gpu_e2e.py) ----
            errs = {k: float((g_cpu[k] - o_out[k]).abs().max()) for k in g_keys}
            e_lists = max(float((x - y).abs().max()) for x, y in zip(g_lists, o_out["im_t_list"]))
            e_flow = max(errs["opt_flow_0"], errs["opt_flow_1"])
            mse = float(((g_cpu["I_t"] - o_out["I_t"]).double() ** 2).mean())
            result["parity"] = {"max_abs_I_t": errs["I_t"], "max_abs_flow": e_flow, "tol": 1e-3, "tol_flow": 2e-3,
                                "max_abs_other": {k: errs[k] for k in ("occ_mask1", "I_t_0", "I_t_1")}, "max_abs_im_t_list": e_lists,
                                "psnr_vs_oracle_db": None if mse == 0 else round(-10.0 * float(np.log10(mse)), 2),
                                "flow_abs_max": float(max(o_out["opt_flow_0"].abs().max(), o_out["opt_flow_1"].abs().max())),
                                "pass": bool(errs["I_t"] <= 1e-3 and e_flow <= 2e-3 and e_lists <= 1e-3),
                                "inputs": f"frames[0] of the timed set ({what}); every pixel of every returned tensor, HIP forward "
                                          f"({args.precision}) vs the CPU oracle forward timed for cpu_baseline"}
            del g_cpu, g_lists, o_out
        if collective:
            result["collective_backend"] = dist.get_backend()
            result["collective_world_size"] = dist.get_world_size()          # what the communicator itself reports, not WORLD_SIZE
            result["per_rank_ms_per_step"] = per_rank_ms
            if rehearsal:
                # every rank on ONE card, gloo instead of RCCL: the code path, not a scaling measurement -- the line must not read as one
                result["rehearsal_frames_per_s"] = result["value"]
                result["value"] = None
                result["metric"] = "rehearsal (all ranks on cuda:0 over gloo: the N > 1 code path, not a scaling measurement)"
        print(json.dumps(result), file=_REAL_STDOUT, flush=True)
    if collective:
        dist.barrier()                      # rank 0's instrumented pass is over: nobody tears the communicator down under it
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
