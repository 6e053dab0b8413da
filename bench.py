#!/usr/bin/env python3
"""bench.py -- stereo pairs/sec of the PL-SLAM matching front end on MI355X.

A "step" is one pass of the hot path over one device-resident batch of synthetic stereo pairs
(BASELINE.json config 2: 752x480-shaped stream, 1500 ORB + 200 LBD per image; per pair
ORB L<->R, ORB prev<->curr, LBD L<->R, LBD prev<->curr, each a mutual + ratio StVO::match, followed by
StereoFrame's epipolar / disparity / overlap gates over the two L<->R tables -- SURVEY 8 a1-a5).
With N > 1 ranks (one per GPU, torch.distributed 'nccl' == RCCL) every rank runs its own shard of
pairs (weak scaling) and the per-pair match tables are gathered to rank 0 inside the step.

Prints ONE compact JSON line on rank 0 (the contract's keys, < 6 kB) and writes the full record -- secondary
records, distributions, notes -- to gpurun_out/bench_full.json (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

# HIP backs its streams with a few hardware queues (GPU_MAX_HW_QUEUES, default 4 per priority); streams that share one run their
# kernels in order.  This process keeps several in flight -- scan stream, stage stream, the process group's collective stream,
# torch's own -- and with 4 queues the collective of the N > 1 step lands behind the scans for some stream-creation orders: the
# one-rank RCCL step measured 0.84-0.88 of the plain step with 4 and 8 queues, 0.98 with 16 and 32 (profiles/
# r4_e_hw_queues.txt).  Must be set before the HIP runtime initialises (torch is imported in main()).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
VALU_LANES_PER_CLK_PER_CU = 128  # 4 SIMD-32 per CU
FP4_MFMA_OPS_PER_CLK_PER_SIMD = 4096  # v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 operands: 131072 ops in 8 passes of 4 clk


def usable_cpus() -> int:
    """CPUs this process may actually use: scheduler affinity capped by the cgroup CPU quota
    (containers: os.cpu_count() reports the host's CPUs, not the quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse:
                quota, period = parse(txt)
                if quota != "max":
                    n = min(n, max(1, int(int(quota) / int(period))))
            else:
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if int(txt) > 0:
                    n = min(n, max(1, int(int(txt) / period)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


SCAN_KERNEL_SOURCES = ("common.hpp", "mfma_h_common.hpp", "hamming.hip", "hamming_mfma.hip", "hamming_mfma_g.hip", "hamming_mfma_h.hip",
                       "hamming_mfma_i.hip", "hamming_mfma_d.hip")


def kernel_source_hash() -> str:
    """Identifies the sources of the scan kernels a PMC pass was taken on (profiles/pmc_traffic.json is keyed by it): the
    files that define K1a-K1f and their shared declarations."""
    h = hashlib.sha256()
    d = os.path.join(_ROOT, "plslam_amd", "csrc")
    for name in SCAN_KERNEL_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


NATIVE_STEP = None             # --gather-step: None = the one-call C-ABI step whenever the process group exposes its communicator
COMPACT_LINE_CAP = 6144        # bytes of the one stdout line (target <= 6 kB, the driver failed at 22 kB)


def _short(s, n=120):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


def compact_line(full: dict, side_path) -> dict:
    """The ONE stdout line: the bench contract's keys, `roofline`, `hbm_roofline`, `cpu_baseline`, `verified` -- numbers and
    short strings only.  Everything else of the full record (secondary records, step-time distribution, the one-batch
    comparison, executed-instruction figures, notes) is in the side file `details`."""
    c = full["config"]
    r = full["roofline"]
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": _short(c["workload"], 200), "pairs_per_gpu_per_step": c["pairs_per_gpu_per_step"],
                     "pairs_per_step_all_gpus": c["pairs_per_step_all_gpus"], "nnr_p": c["nnr_p"], "nnr_l": c["nnr_l"],
                     "mutual": c["mutual"], "distinct_batches_in_rotation": c["distinct_batches_in_rotation"],
                     "stereo_gates": c["stereo_gates"] is not None, "kernel": c["kernel"], "mfma_form": c["mfma_form"],
                     "kernel_source_hash": c["kernel_source_hash"], "rccl_ranks_seen": c["rccl_ranks_seen"],
                     "gather_wire": None if c["gather_wire"] is None else {k: c["gather_wire"][k] for k in ("format", "comm")},
                     "parallelism": _short(c["parallelism"], 100)}
    out["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
                                             "frac_profiles", "frac_profiles_source")}
    h = full["hbm_roofline"]
    out["hbm_roofline"] = {k: h.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    out["kernel_ms"] = {k: v for k, v in full["kernel_ms"].items() if k != "note"}
    if "cpu_baseline" in full:
        b = full["cpu_baseline"]
        out["cpu_baseline"] = {"value": b["value"], "unit": b["unit"], "cores": b["cores"],
                               "host_logical_cpus": b.get("host_logical_cpus"), "kind": b["kind"],
                               "cores_note": f"{b['cores']} usable of {b.get('host_logical_cpus')} logical CPUs (cgroup quota)",
                               "sample": _short(b.get("sample"), 160)}
    out["verified"] = {k: _short(v, 160) for k, v in full["verified"].items()}
    c4 = (full.get("secondary") or {}).get("config4_strong")
    if c4 is not None and full["n_gpus"] >= 1 and "value" in c4:
        out["config4_strong"] = {k: c4.get(k) for k in ("value", "unit", "n_gpus", "scaling", "pairs_per_gpu_per_step", "ms_per_step",
                                                        "host_ms_per_step")}
    out["device"] = full.get("device")
    out["details"] = side_path
    return out


def oracle_tables(stream, n_orb, n_lbd, nnr_p, nnr_l, threads=None):
    """The oracle's match tables of a WHOLE batch, (B, 2 n_orb + 2 n_lbd) in the GPU table's row layout, on all usable cores:
    the checker of a timed output (every pair, not a sample).  Returns (tables, seconds)."""
    from oracle import oracle as O
    L = O.native_lib()
    B = stream["orb_l"].shape[0] - 1
    threads = threads or usable_cpus()

    def pack(lst, n):
        return np.ascontiguousarray(np.concatenate(lst)), np.arange(0, (len(lst) + 1) * n, n, dtype=np.int32)
    d1o, d2o, d1l, d2l = [], [], [], []
    for i in range(B):
        d1o += [stream["orb_l"][i + 1], stream["orb_l"][i]]
        d2o += [stream["orb_r"][i + 1], stream["orb_l"][i + 1]]
        d1l += [stream["lbd_l"][i + 1], stream["lbd_l"][i]]
        d2l += [stream["lbd_r"][i + 1], stream["lbd_l"][i + 1]]
    (a, oa), (b, ob), (c, oc), (d, od) = pack(d1o, n_orb), pack(d2o, n_orb), pack(d1l, n_lbd), pack(d2l, n_lbd)
    t0 = time.perf_counter()
    mo, _ = O.match_batched(a, oa, b, ob, nnr_p, True, nthreads=threads, L=L)
    ml, _ = O.match_batched(c, oc, d, od, nnr_l, True, nthreads=threads, L=L)
    dt = time.perf_counter() - t0
    return np.concatenate([mo.reshape(B, 2 * n_orb), ml.reshape(B, 2 * n_lbd)], axis=1), dt


def cpu_baseline(stream, n_orb, n_lbd, nnr_p, nnr_l, budget_s=15.0):
    """The CPU restatement of the reference path (oracle, -O3 -march=native, popcnt) timed on this
    host's cores on a bounded sample of the SAME workload.  kind = "port".  Also returns the match tables of its first
    all-cores pass over the WHOLE batch -- (B, 2 n_orb + 2 n_lbd), the layout of the GPU table -- so that the caller can
    verify every pair of the timed output, not a sample."""
    from oracle import oracle as O
    L = O.native_lib()
    cores = usable_cpus()      # threads actually used = CPUs the container may use (cgroup quota)

    def pack(lst, n):
        return np.ascontiguousarray(np.concatenate(lst)), np.arange(0, (len(lst) + 1) * n, n, dtype=np.int32)

    def problems(pairs):                                # packing is NOT part of the timed region
        d1o, d2o, d1l, d2l = [], [], [], []
        for i in pairs:
            d1o += [stream["orb_l"][i + 1], stream["orb_l"][i]]
            d2o += [stream["orb_r"][i + 1], stream["orb_l"][i + 1]]
            d1l += [stream["lbd_l"][i + 1], stream["lbd_l"][i]]
            d2l += [stream["lbd_r"][i + 1], stream["lbd_l"][i + 1]]
        return pack(d1o, n_orb), pack(d2o, n_orb), pack(d1l, n_lbd), pack(d2l, n_lbd)

    def run(packed, threads):
        (a, oa), (b, ob), (c, oc), (d, od) = packed
        t0 = time.perf_counter()
        mo, _ = O.match_batched(a, oa, b, ob, nnr_p, True, nthreads=threads, L=L)
        ml, _ = O.match_batched(c, oc, d, od, nnr_l, True, nthreads=threads, L=L)
        return time.perf_counter() - t0, mo, ml

    B = stream["orb_l"].shape[0] - 1
    n_1 = min(B, 8)
    one = problems(list(range(n_1)))
    t_1, reps_1 = 0.0, 0
    while t_1 < 0.25 * budget_s:                        # single-thread leg, bounded by wall clock
        t_1 += run(one, 1)[0]
        reps_1 += 1
    full = problems(list(range(B)))
    t_mt, reps, tables = 0.0, 0, None
    while t_mt < 0.75 * budget_s:                       # all-cores leg: whole passes over the batch
        dt, mo, ml = run(full, cores)
        t_mt += dt
        reps += 1
        if tables is None:                              # (pair, [L->R, prev->curr], row) -> the GPU table's row layout
            tables = np.concatenate([mo.reshape(B, 2 * n_orb), ml.reshape(B, 2 * n_lbd)], axis=1)
    rec = {"value": B * reps / t_mt, "unit": "stereo pairs/s", "cores": cores, "kind": "port",
           "host_logical_cpus": os.cpu_count(),
           "sample": f"{reps} pass(es) over the same {B}-pair batch on {cores} threads ({t_mt:.1f} s); "
                     f"1 thread: {reps_1} pass(es) over {n_1} pairs ({t_1:.1f} s)",
           "value_1thread": n_1 * reps_1 / t_1,
           "note": "a restatement of the reference path (oracle/plslam_oracle.c: per-query scalar scan with popcnt, no "
                   "tiling, no AVX-512 vpopcnt), not the reference binary, which cannot be built in this image; a "
                   "reported baseline -- the GPU/CPU ratio says nothing about kernel quality"}
    return rec, tables


def self_launch(n: int, argv) -> int:
    """Re-executes this script as `n` ranks of one node: python -m torch.distributed.run --nnodes=1 --nproc-per-node n
    --master-addr 127.0.0.1 --master-port <a free port> bench.py <the same arguments>.  The ranks inherit stdout: rank 0's one
    JSON line is this command's one JSON line.  Returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // max(n, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] self-launch: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def launch_check() -> int:
    """The launch path without a GPU (tests/test_dist_cpu.py): every rank joins a gloo group, the ranks' numbers are summed,
    rank 0 prints one JSON line."""
    sys.stdout.flush()
    real_stdout = os.dup(1)                 # (gloo / RCCL print banners on fd 1: stdout carries the JSON line only)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        os.write(real_stdout, (json.dumps({"launch_check": True, "world_size": world, "rank_sum": int(t.item()),
                                           "argv": sys.argv[1:]}) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
    return 0


def gather_probe(bm, world, rank, dev, cands, note):
    """A few untimed-by-the-bench steps of every (wire format, communication stream) candidate; seconds per step, max over ranks
    (None = a rank could not set the candidate up).  Every rank takes every collective of this function or none: the local
    set-up (allocations) is the only thing inside a try, its outcome is agreed on by an all-reduce OUTSIDE it, and the timed
    steps run unguarded -- an error there must end the run, not let one rank skip ahead while the others wait in a gather."""
    import torch
    import torch.distributed as dist
    from plslam_amd import frontend
    out = {}
    for wire, comm in cands:
        name, g_, ok = f"{wire}/{comm}", None, 1
        try:
            g_ = frontend.PipelinedGather(bm, world, rank, root=0, compact=(wire == "int16"), comm_on_stage_stream=(comm == "stage"), native=NATIVE_STEP)
        except Exception as e:                       # (an allocation failed on THIS rank)
            ok = 0
            note(f"gather probe {name}: set-up failed on rank {rank}: {type(e).__name__}: {e}")
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            out[name] = None
            if g_ is not None:
                g_.close()
            del g_
            continue
        for k in range(2):
            g_.step(k)
        g_.finish()
        bm.synchronize_all()
        dist.barrier()
        t0_ = time.perf_counter()
        for k in range(2, 8):
            g_.step(k)
        g_.finish()
        bm.synchronize_all()
        tt = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        out[name] = float(tt.item()) / 6
        g_.close()
        del g_
    torch.cuda.synchronize(dev)
    return out


def shard_gather_record(ctx, dev, args, world, rank, pairs_total, wire, comm, steps, warm, reps, note, tag, streams=None):
    """ONE batch of `pairs_total` stereo pairs per step, sharded contiguously over the ranks (BASELINE config 4 as written:
    4096 pairs -> 512 per GPU at N = 8), stepped exactly as the headline's N > 1 loop steps (PipelinedGather.step + one event per
    step on the stage stream) -- and, in the same process on the same matcher, the plain step without the gather.  Every rank
    calls this (the timings are all-reduced); the record comes back on rank 0, whose check covers >= 64 pairs of EVERY rank's
    shard in both gathered buffers.  world == 1: a (forced) one-rank RCCL group -- the step's own overhead, not a link."""
    import torch
    import torch.distributed as dist
    from plslam_amd import frontend, synth
    n_orb, n_lbd = args.n_orb, args.n_lbd
    lo, hi = frontend.shard_range(pairs_total, world, rank)
    Bs = hi - lo
    if pairs_total % world:
        return {"skipped": f"{pairs_total} pairs do not divide over {world} ranks"}
    st = synth.stereo_stream(Bs, n_orb, n_lbd, seed=synth.SEED0, first_pair=lo)
    # streams: the [scan stream, stage stream] pair of the headline's matcher (N > 1: created after the process group, and known
    # to be a good pair -- the headline ran on it), or None for a pair of this record's own.  HIP deals streams to a few hardware
    # queues in creation order, and two of {scan stream, stage stream, the collective's internal stream} on one queue serialise
    # (round 4: 0.84 of the plain step "for some creation orders"; round 5: this record inside an N > 1 run with streams of its
    # own gave stretches of 1.48 / 1.13 / 1.25 M pairs/s, and inside the N = 1 run on the headline's OLDER pair -- older than
    # the process group -- 0.62).  The rule that measured 0.98-0.99 every time: process group first, then the streams.
    bm = frontend.StereoBatchMatcher(ctx, st, nnr_p=args.nnr_p, nnr_l=args.nnr_l, mutual=True, device=dev, n_buffers=2,
                                     geometry=synth.stereo_geometry(st, first_pair=lo), gates=dict(synth.KITTI_GATES),
                                     streams=streams)
    stage = bm.stage_stream

    def maxed(dt):
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def stretches(step_fn, sync_fn):
        for k in range(warm):
            step_fn(k)
        sync_fn()
        dts = []
        for r_ in range(reps):
            dist.barrier()
            torch.cuda.synchronize(dev)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
            t0 = time.perf_counter()
            for k in range(steps):
                step_fn(warm + r_ * steps + k)
                evs[k].record(stage)
            sync_fn()
            dts.append(maxed(time.perf_counter() - t0))
        return dts

    def sync_plain():
        bm.synchronize_all()
        torch.cuda.synchronize(dev)

    plain = stretches(lambda k: bm.run_overlapped(k), sync_plain)
    pg = frontend.PipelinedGather(bm, world, rank, root=0, compact=(wire == "int16"), comm_on_stage_stream=(comm == "stage"), native=NATIVE_STEP)

    def sync_gather():
        pg.finish()
        bm.synchronize_all()
        torch.cuda.synchronize(dev)

    pg.host_ms_per_step()
    gath = stretches(pg.step, sync_gather)
    host_ms = pg.host_ms_per_step()
    rec = None
    if rank == 0:
        from oracle import oracle as O
        sample = frontend.spread_sample(Bs, 64)
        firsts = [frontend.shard_range(pairs_total, world, r_)[0] for r_ in range(world)]
        for b_ in range(2):
            bad = frontend.verify_gathered_tables(pg.gathered(b_).cpu().numpy(), world, Bs, n_orb, n_lbd, args.nnr_p, args.nnr_l, sample,
                                                  lambda d1, d2, nnr: O.match(d1, d2, nnr, True)[0], local_stream=st, first_pairs=firsts)
            if bad:
                raise SystemExit(f"{tag}: gathered buffer {b_} differs from the oracle: (rank, pair, problem) = {bad[:4]}")
        dp, dg = float(np.median(plain)), float(np.median(gath))
        rec = {"metric": f"stereo pairs/sec ({n_orb} ORB + {n_lbd} LBD BF-match)", "value": pairs_total * steps / dg,
               "unit": "stereo pairs/s", "n_gpus": world, "scaling": "strong", "pairs_per_step_all_gpus": pairs_total,
               "pairs_per_gpu_per_step": Bs, "steps": steps, "ms_per_step": 1e3 * dg / steps,
               "workload": f"{pairs_total} pairs per step in contiguous shards of {Bs} per rank with a one-pair halo, gate stage "
                           "included, RCCL gather of the match tables to rank 0 under the next step's scan "
                           "(BASELINE config 4 when pairs_per_step_all_gpus = 4096 and n_gpus = 8)",
               "gather": {"format": wire, "comm": comm, "int16_written_by": "k_finalize" if pg.kernel_wire16 else None,
                          "step": "plslam_match_plan_step_gather" if pg.native else "torch"},
               "plain_step_same_run": {"value": pairs_total * steps / dp, "ms_per_step": 1e3 * dp / steps,
                                       "what": "the same matcher, the same stepping, no gather: what the shard's step costs by itself"},
               "over_plain_step_same_run": dp / dg,
               "host_ms_per_step": host_ms,
               "stretches_pairs_per_s": [pairs_total * steps / d for d in gath],
               "plain_stretches_pairs_per_s": [pairs_total * steps / d for d in plain],
               "how": f"median of {reps} stretches of {steps} steps after {warm} warm-up steps, each stretch between barriers, max "
                      "over ranks; host_ms_per_step = wall time inside PipelinedGather.step (enqueueing only)",
               "verified": f"{len(sample)} pairs x 4 problems of each of {world} rank(s), both GATHERED buffers, bit-exact vs the oracle",
               "note": ("one rank: what it measures is the step's own overhead (events, the collective's launch, the int16 table), "
                        "not a link; " if world == 1 else "") + "no 1 -> 8 curve has been measured on hardware by the builder"}
        note(f"  {tag}: {rec['value']:.0f} pairs/s ({rec['over_plain_step_same_run']:.3f} of the plain step), host {host_ms:.3f} ms/step")
    pg.close()
    bm.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-gpu", type=int, default=4096,
                    help="stereo pairs per GPU per step: the device-resident batch (BASELINE config 4: 4096)")
    ap.add_argument("--n-orb", type=int, default=1500)
    ap.add_argument("--n-lbd", type=int, default=200)
    ap.add_argument("--nnr-p", type=float, default=0.75)
    ap.add_argument("--nnr-l", type=float, default=0.75)
    ap.add_argument("--scan-variant", type=int, default=0)
    ap.add_argument("--scan-block", type=int, default=0)
    ap.add_argument("--sym-rows", type=int, default=0)
    ap.add_argument("--group-cap", type=int, default=0)
    ap.add_argument("--mfma-form", type=int, default=0, help="0 = auto (K1i), 1 = K1e (best-2 push per tile), 2 = K1f (group minima, rows), 3 = K1g (two directed scans per mutual problem), 4 = K1h (group minima both ways), 5 = K1i (K1h, M-tiles pipelined against each other, unscaled MFMA, f16 three-input minima)")
    ap.add_argument("--fuse", type=int, default=0, help="K1f: 0 = auto, 1 = never, 2 = always one workgroup per problem incl. merge + finalize")
    ap.add_argument("--post-wgs", type=int, default=0, help="cap on the workgroups of the stages behind a scan when they run beside the next scan (option post_workgroups; 0 = the library's default)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT", help="any other plslam_ctx option, e.g. --opt exact_second=1")
    ap.add_argument("--no-gates", action="store_true", help="leave the stereo-gate stage out of the step (tables only)")
    ap.add_argument("--step-streams", type=int, default=2, help="output buffers / HIP streams the steps alternate over")
    ap.add_argument("--no-overlap", action="store_true",
                    help="single GPU: run the steps strictly one after another on one stream")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --pairs-per-gpu pairs on EVERY rank per step (default); strong: --pairs-per-gpu is the TOTAL per step, "
                         "sharded contiguously over the ranks (BASELINE config 4: 4096 pairs -> 512 per GPU at N = 8)")
    ap.add_argument("--batches", type=int, default=3,
                    help="N = 1: distinct input batches rotated through the timed loop (1 = the same batch every step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary records (tables only, popcount kernels, C5, C3) of the N = 1 line")
    ap.add_argument("--gather-wire", choices=("auto", "int16", "int32"), default="auto",
                    help="N > 1: wire format of the table gather (auto: both are tried for a few untimed steps, the faster one runs)")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the NCCL(RCCL) process group and run the table gather even with one rank")
    ap.add_argument("--gather-comm", choices=("auto", "stage", "own"), default="auto",
                    help="N > 1: where the wait for the collective (and the root's widening) is enqueued -- the matcher's stage stream "
                         "or a communication stream of its own (auto: both are tried by the probe; without a probe: stage at one "
                         "rank, own at N > 1)")
    ap.add_argument("--gather-step", choices=("auto", "native", "torch"), default="auto",
                    help="N > 1: the step as ONE C-ABI call (plslam_match_plan_step_gather on the process group's communicator) or "
                         "torch's streams / events / c10d gather (round 5); auto = native whenever the communicator can be had")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    ap.add_argument("--full-json", default=None, metavar="PATH",
                    help="where the FULL record goes (secondary records, distributions, notes); default gpurun_out/bench_full.json. "
                         "stdout carries one compact line of the contract's keys")
    ap.add_argument("--launch-check", action="store_true",
                    help="(tests) after the self-launch: a gloo rendezvous of the ranks and ONE JSON line from rank 0 -- no GPU work")
    args = ap.parse_args()
    global NATIVE_STEP
    NATIVE_STEP = {"auto": None, "native": True, "torch": False}[args.gather_step]

    # `python bench.py --gpus N` by itself: one process per GPU under torch.distributed.run on this node (the driver's
    # `python -m torch.distributed.run ... bench.py --gpus N` form arrives with WORLD_SIZE set and runs as it is).  --force-dist
    # takes the same road at N = 1, so that the one-rank RCCL run exercises the launch path a SCALE run would take.
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_dist or args.launch_check):
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.launch_check:
        raise SystemExit(launch_check())

    # stdout must carry exactly ONE JSON line; libraries (e.g. RCCL's version banner, flushed at
    # exit) also write to fd 1.  Keep the real stdout aside and point fd 1 at stderr meanwhile.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    T0 = time.perf_counter()

    def note(msg):                                      # progress on stderr (stdout carries the JSON)
        print(f"[bench +{time.perf_counter() - T0:6.1f}s] {msg}", file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    import plslam_amd
    from plslam_amd import frontend, synth
    note("imports done")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world                               # (the launcher's world size is the truth)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one rank per GPU: LOCAL_RANK is the device index -- unless the launcher masks the devices per process (each then sees one)
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        import datetime
        # (a rank that dies leaves the others in a collective: they give up after this long instead of hanging the node)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(minutes=10))

    n_orb, n_lbd = args.n_orb, args.n_lbd
    if args.scaling == "strong":
        # BASELINE config 4 as written: ONE batch of --pairs-per-gpu pairs per step, contiguous shards of total / N pairs
        if args.pairs_per_gpu % world:
            raise SystemExit(f"--scaling strong: {args.pairs_per_gpu} pairs do not divide over {world} ranks")
        first_pair, hi_ = frontend.shard_range(args.pairs_per_gpu, world, rank)
        B = hi_ - first_pair
    else:
        B = args.pairs_per_gpu                   # weak scaling: rank r owns pairs [r*B, (r+1)*B) of one global stream
        first_pair = rank * B
    baseline_cfg = {(800, 100): "C1-shaped (KITTI 800 ORB + 100 LBD)", (1500, 200): "C2", (4000, 600): "C5"}
    cfg_name = baseline_cfg.get((n_orb, n_lbd), "custom")
    # every shard carries a one-pair halo (the left descriptors of the pair before its first)
    stream = synth.stereo_stream(B, n_orb, n_lbd, seed=synth.SEED0, first_pair=first_pair)
    gates = None if args.no_gates else dict(synth.KITTI_GATES)
    geo = None if args.no_gates else synth.stereo_geometry(stream, first_pair=first_pair)

    # N > 1: every rank must sit on its own GPU (a launcher that maps two ranks to one device would still "scale")
    ranks_seen = None
    if use_dist:
        props = torch.cuda.get_device_properties(dev)
        hw = "|".join(str(getattr(props, k)) for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id") if hasattr(props, k))
        ident = hw or f"unidentified:{rank}"          # (no hardware identifier exposed: the ranks cannot be told apart, nor accused)
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        ranks_seen = {"world_size": dist.get_world_size(), "distinct_devices": len(set(idents))}
        if rank == 0 and ranks_seen["distinct_devices"] != world:
            raise SystemExit(f"{world} ranks on {ranks_seen['distinct_devices']} distinct devices: {idents}")
    note(f"synthetic stream of {B} pairs generated")
    ctx = plslam_amd.Context(dev_index)      # raises if libplslam_hip.so / a gfx950 device is missing
    for key, val in (("scan_variant", args.scan_variant), ("scan_block", args.scan_block), ("sym_rows", args.sym_rows),
                     ("group_cap", args.group_cap), ("mfma_form", args.mfma_form), ("fuse", args.fuse), ("post_workgroups", args.post_wgs)):
        if val:
            ctx.set_option(key, val)
    for kv in args.opt:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    overlap = not use_dist and not args.no_overlap
    n_buf = max(2, args.step_streams) if (use_dist or overlap) else 1
    bm = frontend.StereoBatchMatcher(ctx, stream, nnr_p=args.nnr_p, nnr_l=args.nnr_l, mutual=True, device=dev,
                                     n_buffers=n_buf, geometry=geo, gates=gates)
    info = bm.plan.info()
    devinfo = ctx.device_info()
    # N = 1: consecutive steps are DIFFERENT batches (round 2 alternated two plans over one batch, whose LBD sets and tables
    # fit the 256 MB Infinity Cache): `--batches` matchers with their own descriptors, tables and plans step through the
    # same pair of streams in rotation.  (N > 1 keeps one batch per rank: the gather pipeline owns the table buffers.)
    nbatch = max(1, args.batches) if (overlap and not use_dist) else 1
    extra_streams, extra_bms = [], []
    for k_ in range(1, nbatch):
        st_k = synth.stereo_stream(B, n_orb, n_lbd, seed=synth.SEED0 + 7919 * k_, first_pair=first_pair)
        geo_k = None if args.no_gates else synth.stereo_geometry(st_k, first_pair=first_pair)
        extra_streams.append(st_k)
        extra_bms.append(frontend.StereoBatchMatcher(ctx, st_k, nnr_p=args.nnr_p, nnr_l=args.nnr_l, mutual=True, device=dev,
                                                     n_buffers=1, geometry=geo_k, gates=gates, streams=bm.streams))
    rotation = [bm.plans[0]] + [e.plans[0] for e in extra_bms]
    # N > 1: the table of step k is gathered to rank 0 over RCCL on a communication stream while
    # step k+1 computes into the other table buffer (steps are independent batches of a stream).
    # Wire format of the gather, chosen by measurement before the warm-up: int16 (half the bytes on every xGMI link, but a
    # narrowing copy on every rank and the widening of world x the table on the root: 43 + 39 us per 4096-pair table at one
    # rank, i.e. ~0.3 ms per step on the root of 8) or int32 (no extra kernels, 20 GB/s per link at 1.5 M pairs/s per GPU).
    # Which is cheaper depends on the links: a few untimed steps of each, the slowest rank's clock decides, every rank takes the
    # same decision (all-reduce of the times).  A failure of the probe keeps int16.
    gather_wire = None
    pg = None
    if use_dist:
        wires = ("int16", "int32") if args.gather_wire == "auto" else (args.gather_wire,)
        if max(n_orb, n_lbd) > 32767:
            wires = ("int32",)
        comms = ("stage", "own") if args.gather_comm == "auto" else (args.gather_comm,)
        default = ("int16" if "int16" in wires else wires[0], ("stage" if world == 1 else "own") if len(comms) > 1 else comms[0])
        cands = [(w_, c_) for w_ in wires for c_ in comms]
        probe = gather_probe(bm, world, rank, dev, cands, note) if len(cands) > 1 else {}
        choice = default
        timed_ok = {k: v for k, v in probe.items() if v is not None}
        if timed_ok:
            best = min(timed_ok, key=timed_ok.get)
            dkey = "/".join(default)
            # the default keeps its place unless another combination wins by 2 % (the probe is six steps long)
            if dkey not in timed_ok or timed_ok[best] < 0.98 * timed_ok[dkey]:
                choice = tuple(best.split("/"))
        pg = frontend.PipelinedGather(bm, world, rank, root=0, compact=(choice[0] == "int16"),
                                      comm_on_stage_stream=(choice[1] == "stage"), native=NATIVE_STEP)
        gather_wire = {"format": choice[0], "comm": choice[1], "int16_written_by": "k_finalize" if pg.kernel_wire16 else None,
                       "step": "plslam_match_plan_step_gather (one C-ABI call)" if pg.native else "torch (streams, events, c10d gather)",
                       "probe_s_per_step": probe,
                       "how": "six untimed steps per (wire format, communication stream) before the warm-up, max over ranks; the "
                              f"default ({'/'.join(default)}) keeps its place unless another wins by 2 %"}
        note(f"gather: {gather_wire}")
    scan_stream, stage_stream = bm.streams[0], bm.stage_stream

    def run_steps(matcher, gather, n, k0=0, events=None, rotate=True):
        for k in range(k0, k0 + n):
            if gather is not None:
                gather.step(k)
            elif overlap and rotate and nbatch > 1:
                rotation[k % nbatch].run_split(scan_stream.cuda_stream, stage_stream.cuda_stream)
            elif overlap:
                # consecutive steps are independent batches: every scan on one HIP stream, the stages behind a scan on a
                # second one (they run under the next step's scan)
                matcher.run_overlapped(k)
            else:
                matcher.run()
            if events is not None:
                events[k - k0 + 1].record(stage_stream if (overlap or gather is not None) else matcher.stream)

    def sync(matcher, gather):
        if gather is not None:
            gather.finish()
        if overlap:
            matcher.synchronize_all()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    note(f"plan built: {info}; {nbatch} distinct batch(es) in rotation")
    run_steps(bm, pg, args.warmup)
    sync(bm, pg)
    note("warmup done")
    all_plans = list(bm.plans) + [e.plans[0] for e in extra_bms]
    for p_ in all_plans:
        p_.set_profiling(True)
        p_.elapsed()                          # reset the accumulators
    # per-step completion events on the stream the step's last stage runs on: steady-state step times (median, p10, p90)
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    step_events[0].record(stage_stream if (overlap or pg is not None) else bm.stream)
    t0 = time.perf_counter()
    run_steps(bm, pg, args.steps, args.warmup, events=step_events)
    sync(bm, pg)
    elapsed = time.perf_counter() - t0
    step_ms = np.array([step_events[i].elapsed_time(step_events[i + 1]) for i in range(args.steps)])
    scan_ms = fin_ms = 0.0
    runs = 0
    for p_ in all_plans:
        a_, b_, n_ = p_.elapsed()
        scan_ms, fin_ms, runs = scan_ms + a_, fin_ms + b_, runs + n_
        p_.set_profiling(False)
    note(f"timed {args.steps} steps in {elapsed:.4f}s; scan {scan_ms / max(runs, 1):.3f} ms/launch")
    # the same K steps over ONE repeated batch (round 2's stepping): how much of the number is cache residency
    repeated = None
    if nbatch > 1:
        run_steps(bm, pg, 2, 0, rotate=False)
        sync(bm, pg)
        t1 = time.perf_counter()
        run_steps(bm, pg, args.steps, 0, rotate=False)
        sync(bm, pg)
        dt1 = time.perf_counter() - t1
        repeated = {"value": B * args.steps / dt1, "unit": "stereo pairs/s", "ms_per_step": 1e3 * dt1 / args.steps,
                    "note": "the same plan pair over one batch every step (round 2's timed loop); the headline rotates "
                            f"{nbatch} distinct batches"}

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # (before the config-4 record below: rank 0's check of that record is seconds of CPU work, and five launches on a chip that
    # has idled meanwhile measure its clock ramp -- a forced one-rank run printed 2.65-2.90 ms here for a 2.3 ms kernel)
    # In the timed region consecutive steps overlap on two streams, so a kernel's start-to-end time
    # there includes the share of the GPU the other step's kernels took.  Measure the scan kernel's
    # EXCLUSIVE duration too: a few strictly serial launches, same plan, same data, HIP events.
    excl_scan_ms = excl_fin_ms = None
    if rank == 0:
        p0 = bm.plans[0]
        p0.set_profiling(True)
        p0.elapsed()
        for _ in range(5):
            p0.run(bm.streams[0].cuda_stream)
            bm.streams[0].synchronize()
        a_, b_, n_ = p0.elapsed()
        p0.set_profiling(False)
        excl_scan_ms, excl_fin_ms = a_ / max(n_, 1), b_ / max(n_, 1)
        note(f"exclusive kernel times: scan {excl_scan_ms:.3f} ms, post-scan stages {excl_fin_ms:.3f} ms")

    # N > 1: BASELINE config 4 AS WRITTEN beside the weak-scaling headline, from the same process group -- ONE batch of 4096
    # pairs per step sharded over the ranks (512 per GPU at N = 8), gathered with the format the probe chose.  Every rank takes
    # part; rank 0's (long) verification of the headline's tables comes after it, when no collective is left.  A forced one-rank
    # group runs config 4's per-GPU shard (512 pairs) instead: the step's own overhead.
    config4 = None
    if use_dist and not args.no_secondary:
        total4 = 4096 if world > 1 else 512
        note(f"config 4 as written: {total4} pairs per step over {world} rank(s) ...")
        config4 = shard_gather_record(ctx, dev, args, world, rank, total4, gather_wire["format"], gather_wire["comm"],
                                      steps=150, warm=10, reps=3, note=note, tag="config4_strong", streams=bm.streams)

    # full-size determinism check: every output buffer was computed from the same inputs (under
    # overlap / contention), so whole tables and count arrays must be bit-identical
    if args.steps + args.warmup >= 2:
        for b_ in range(1, len(bm.tables)):
            same = torch.equal(bm.tables[0], bm.tables[b_]) and torch.equal(bm.count_bufs[0], bm.count_bufs[b_])
            if gates is not None:
                same = same and torch.equal(bm.stereo_tabs[0], bm.stereo_tabs[b_]) and \
                    torch.equal(bm.stereo_disps[0].view(torch.int64), bm.stereo_disps[b_].view(torch.int64)) and \
                    torch.equal(bm.stereo_cnts[0], bm.stereo_cnts[b_])
            if not same:
                raise SystemExit(f"rank {rank}: output buffers 0 and {b_} differ (nondeterministic result)")

    out = None
    if rank == 0:
        from oracle import oracle as O
        sl = frontend.table_slices(n_orb, n_lbd)
        # ---- CPU baseline (N = 1 only) and, with it, the oracle's tables for the WHOLE batch --------------------------
        cpu_rec, cpu_tables = None, None
        if world == 1 and not args.no_cpu_baseline:
            note("cpu baseline ...")
            cpu_rec, cpu_tables = cpu_baseline(stream, n_orb, n_lbd, args.nnr_p, args.nnr_l, args.cpu_budget_s)
        # ---- verification of the timed output against the oracle (the checker, not the product) -----------------------
        bufs = list(range(len(bm.tables))) if args.steps + args.warmup >= 2 else [0]
        verified = {"match_tables": None, "stereo_gates": None}
        if cpu_tables is not None:
            # every pair, every problem of rank 0's batch: the table the cpu_baseline leg computed anyway
            for b_ in bufs:
                got = bm.tables[b_].cpu().numpy()
                if not np.array_equal(got, cpu_tables):
                    bad = np.argwhere(got != cpu_tables)
                    raise SystemExit(f"bench output differs from the oracle: buffer {b_}, {len(bad)} entries, first at "
                                     f"pair {bad[0][0]} column {bad[0][1]}")
            verified["match_tables"] = f"all {B} pairs x 4 problems bit-exact vs the oracle ({len(bufs)} buffer(s))"
        # N > 1 (or no CPU baseline): pair 0 and a spread of pairs of EVERY rank's table as it arrived on rank 0
        # (through the RCCL gather, both buffers)
        sample = frontend.spread_sample(B, 64) if cpu_tables is None else [0]
        firsts = [frontend.shard_range(args.pairs_per_gpu, world, r_)[0] if args.scaling == "strong" else r_ * B for r_ in range(world)]
        for b_ in bufs:
            full = (pg.gathered(b_) if pg is not None else bm.tables[b_]).cpu().numpy()
            bad = frontend.verify_gathered_tables(full, world if pg is not None else 1, B, n_orb, n_lbd, args.nnr_p,
                                                  args.nnr_l, sample, lambda d1, d2, nnr: O.match(d1, d2, nnr, True)[0],
                                                  local_stream=stream, first_pairs=firsts)
            if bad:
                raise SystemExit(f"bench output differs from the oracle: buffer {b_}, (rank, pair, problem) = {bad[:4]}")
        if verified["match_tables"] is None:
            verified["match_tables"] = (f"{len(sample)} pairs (the shard's ends + four runs spread over it) x 4 problems of EACH of "
                                        f"{world if pg is not None else 1} rank(s) bit-exact vs the oracle ({len(bufs)} buffer(s)"
                                        f"{', as gathered on rank 0' if pg is not None else ''})")
        if gates is not None:
            gsample = sorted(set(range(0, B, max(1, B // 64))) | {B - 1})
            ref_tab = cpu_tables if cpu_tables is not None else bm.tables[0].cpu().numpy()
            st_, sd_, sc_ = (x.cpu().numpy() for x in (bm.stereo_tabs[0], bm.stereo_disps[0], bm.stereo_cnts[0]))
            with np.errstate(all="ignore"):
                for i_ in gsample:
                    ep, dp, cp = O.stereo_point_gate(ref_tab[i_, sl["orb_lr"]], geo["kp_l"][i_ + 1], geo["kp_r"][i_ + 1],
                                                     gates["max_dist_epip"], gates["min_disp"])
                    el, dl, cl = O.stereo_line_gate(ref_tab[i_, sl["lbd_lr"]], geo["seg_l"][i_ + 1], geo["seg_r"][i_ + 1],
                                                    gates["min_disp"], gates["line_horiz_th"], gates["stereo_overlap_th"],
                                                    gates["ls_min_disp_ratio"])
                    ok = (np.array_equal(st_[i_, :n_orb], ep) and np.array_equal(st_[i_, n_orb:], el) and
                          np.array_equal(sd_[i_, :n_orb].view(np.uint64), dp.view(np.uint64)) and
                          np.array_equal(sd_[i_, n_orb:].view(np.uint64), dl.reshape(-1).view(np.uint64)) and
                          sc_[i_].tolist() == [cp, cl])
                    if not ok:
                        raise SystemExit(f"bench stereo-gate output differs from the oracle: pair {i_}")
            verified["stereo_gates"] = (f"{len(gsample)} pairs spread over the batch bit-exact vs the oracle (tables, "
                                        f"disparities as raw words, counts); kept {int(sc_[:, 0].sum())} points + "
                                        f"{int(sc_[:, 1].sum())} lines of the batch")
        # the other batches of the rotation: whole batches when the CPU leg runs anyway, a spread of pairs otherwise
        for k_, (eb, st_k) in enumerate(zip(extra_bms, extra_streams), start=1):
            got = eb.tables[0].cpu().numpy()
            if cpu_tables is not None:
                ref_k, _ = oracle_tables(st_k, n_orb, n_lbd, args.nnr_p, args.nnr_l)
                if not np.array_equal(got, ref_k):
                    bad = np.argwhere(got != ref_k)
                    raise SystemExit(f"bench output differs from the oracle: batch {k_}, {len(bad)} entries, first at pair {bad[0][0]}")
            else:
                bad = frontend.verify_gathered_tables(got, 1, B, n_orb, n_lbd, args.nnr_p, args.nnr_l,
                                                      frontend.spread_sample(B, 64),
                                                      lambda d1, d2, nnr: O.match(d1, d2, nnr, True)[0], local_stream=st_k)
                if bad:
                    raise SystemExit(f"bench output differs from the oracle: batch {k_}, (rank, pair, problem) = {bad[:4]}")
        if extra_bms:
            verified["match_tables"] += (f"; the {len(extra_bms)} other batch(es) of the rotation: " +
                                         ("every pair" if cpu_tables is not None else "the same spread of pairs"))
        note(f"output verified: {verified}")

        pairs_total = B * world * args.steps
        # duration of the dominant kernel: exclusive (serial launches) for the roofline; the in-region
        # figure (overlapped with the neighbouring step) is reported next to it
        scan_s = excl_scan_ms / 1e3
        achieved_gbs = info["algorithmic_bytes"] / scan_s / 1e9
        mfma = info["scan_variant"] == 4
        form = ctx.get_option("mfma_form")
        kernel_name = {4: {1: "k_scan_sym_mfma", 2: "k_scan_sym_mfma_g", 3: "k_scan_dir_mfma", 4: "k_scan_sym_mfma_h"}.get(form, "k_scan_sym_mfma_i"),
                       3: "k_scan_symmetric" + ("_r4" if info["scan_block_threads"] == 64 else ""),
                       2: "k_scan_wave_per_query", 1: "k_scan_lane_per_query"}.get(info["scan_variant"], "k_scan")
        # HBM bytes / executed instructions of the dominant kernel per launch: PMC counters cannot be read from inside
        # this process, so the figures come from the committed rocprofv3 passes of this same command
        # (profiles/pmc_traffic.json) -- and ONLY when they were taken on these kernel sources (hash) and this workload.
        src_hash = kernel_source_hash()
        wkey = f"{n_orb}+{n_lbd}:pairs{B}:{kernel_name}"
        traffic, executed, pmc_note = None, None, "no PMC entry for these kernel sources / this workload"
        try:
            with open(os.path.join(_ROOT, "profiles", "pmc_traffic.json")) as f:
                pm = json.load(f).get("entries", {}).get(wkey)
        except OSError:
            pm = None
        trace_us, trace_file = None, None
        if pm and pm.get("kernel_source_hash") == src_hash:
            trace_us, trace_file = pm.get("trace_avg_us"), pm.get("trace_file")
            traffic = pm["traffic_bytes_per_launch"]
            valu_insts = pm["sq_insts_valu_per_launch"] - pm.get("sq_insts_mfma_per_launch", 0)
            lane_ops = valu_insts * 64 / scan_s
            executed = {"lane_ops_per_s": lane_ops, "sq_insts_valu_per_launch": valu_insts,
                        "sq_insts_mfma_per_launch": pm.get("sq_insts_mfma_per_launch", 0),
                        "measured_ceiling_lane_ops_per_s": pm["measured_int_valu_ceiling_lane_ops_per_s"],
                        "frac_of_measured_ceiling": lane_ops / pm["measured_int_valu_ceiling_lane_ops_per_s"],
                        "source": pm.get("source"),
                        "note": "executed wave64 VALU instructions, MFMAs excluded (PMC) x 64 / exclusive scan time, against "
                                "the issue rate measured for this integer instruction class (16 lanes/clk/SIMD)"}
            pmc_note = pm.get("source")
        elif pm:
            pmc_note = "profiles/pmc_traffic.json holds this workload for OTHER kernel sources (stale): not reported"
        timing_note = ("kernel_ms = exclusive duration (5 serial launches after the timed region, HIP events on the launch "
                       "stream); in the timed region consecutive steps overlap on two streams, so start-to-end times there "
                       "include the other step's share of the GPU")
        hbm_roofline = {
            "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel_name,
            "kernel_ms": 1e3 * scan_s, "kernel_ms_in_timed_region": scan_ms / max(runs, 1), "timing": timing_note,
            "algorithmic_bytes_per_launch": info["algorithmic_bytes"],
            "note": "compulsory-byte model 32(Q+T)+16Q per directed scan; at ~17 k distance evaluations per compulsory "
                    "kilobyte the path is compute-bound on any formulation, so this fraction is small by construction",
        }
        if mfma:
            # distances come from v_mfma_scale_f32_32x32x64_f8f6f4 over fp4 +-1 codes; 256 multiply-accumulates =
            # 512 ops per executed distance
            mx_peak = devinfo["cu_count"] * 4 * FP4_MFMA_OPS_PER_CLK_PER_SIMD * devinfo["clock_khz"] * 1e3
            mfma_ops = info["distance_evals"] * 512
            roofline = {
                "bound": "mfma", "achieved": mfma_ops / scan_s / 1e12, "peak": mx_peak / 1e12, "unit": "TFLOP/s",
                "frac": mfma_ops / scan_s / mx_peak, "traffic": traffic, "kernel": kernel_name,
                "kernel_ms": 1e3 * scan_s, "kernel_ms_in_timed_region": scan_ms / max(runs, 1), "timing": timing_note,
                "algorithmic_ops_per_launch": mfma_ops, "pmc": pmc_note,
                # the same fraction from the COMMITTED rocprofv3 kernel trace of this command (another box, another day: boxes
                # differ by a few per cent), when it was taken on these kernel sources
                "frac_profiles": (mfma_ops / (trace_us * 1e-6) / mx_peak) if trace_us else None,
                "frac_profiles_source": trace_file if trace_us else None,
                "note": "fp4 MFMA v_mfma_f32_32x32x64_f8f6f4 (operands are the e2m1 codes of +-4 -- K1i, unscaled -- or of +-1 "
                        "with a 2^6 block scale -- K1e..K1h; fp32 accumulation of integers below 2^24: exact); dense fp4 peak = "
                        "CUs x 4 SIMDs x 4096 ops/clk x max clock (MI355X_MICROARCH.md measures 9099 T for the 32x32x64 "
                        "shape); 512 ops per executed 256-bit distance (each serves both match directions).  The kernel is "
                        "co-limited by the VALU bookkeeping that consumes the accumulators: see valu_executed",
            }
        else:
            roofline = dict(hbm_roofline, pmc=pmc_note)
        workload = (f"{cfg_name}: synthetic 752x480 stereo stream, {n_orb} ORB + {n_lbd} LBD per image, ORB+LBD L<->R and "
                    "prev<->curr, mutual + ratio (StVO::match)" +
                    ("" if gates is None else ", then StereoFrame's gates over the L<->R tables (config_kitti.yaml:25-36)") +
                    ", device-resident")
        out = {
            "metric": f"stereo pairs/sec ({n_orb} ORB + {n_lbd} LBD BF-match)",
            "value": pairs_total / elapsed,
            "unit": "stereo pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "ms_per_step_distribution": {"median": float(np.median(step_ms)), "p10": float(np.percentile(step_ms, 10)),
                                         "p90": float(np.percentile(step_ms, 90)), "min": float(step_ms.min()),
                                         "max": float(step_ms.max()), "n": int(step_ms.size),
                                         "how": "HIP events on the stream of each step's last stage, rank 0; an interval is the "
                                                "time between the completions of consecutive steps (steady state: the steps "
                                                "overlap), the first one includes the pipeline fill"},
            "vs_baseline": None,
            "dtype": "fp4" if mfma else "u32",
            "dtype_note": ("the operand CODE of +-1 values on the matrix cores; every distance is an exact integer (fp32 accumulation "
                           "of integers below 2^24) and every table of the timed run is compared bit for bit with the oracle's"
                           if mfma else "exact: XOR + popcount on 32-bit words"),
            "data": "synthetic",
            "config": {
                "workload": workload,
                "pairs_per_gpu_per_step": B, "pairs_per_step_all_gpus": B * world, "nnr_p": args.nnr_p, "nnr_l": args.nnr_l,
                "mutual": True, "distinct_batches_in_rotation": nbatch, "rccl_ranks_seen": ranks_seen,
                "stereo_gates": gates,
                "scan_variant": info["scan_variant"], "scan_block_threads": info["scan_block_threads"],
                "mfma_form": form, "kernel": kernel_name, "kernel_source_hash": src_hash,
                "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "gather_wire": gather_wire,
                "parallelism": f"pairs sharded over {world} rank(s); per-step RCCL gather of the match tables "
                               "to rank 0, overlapped with the next step" if use_dist else
                               ("single GPU; consecutive steps alternate two output buffers; every scan on one HIP stream, the stages behind a scan (merge, finalize, gates) on a second, high-priority one: they run under the next step's scan" if overlap
                                else "single GPU"),
            },
            "roofline": roofline,
            "hbm_roofline": hbm_roofline,
            "valu_executed": executed,
            "kernel_ms": {"scan": excl_scan_ms, "post_scan_stages": excl_fin_ms,
                          "scan_in_timed_region": scan_ms / max(runs, 1),
                          "post_scan_stages_in_timed_region": fin_ms / max(runs, 1),
                          "note": "post-scan stages = column-partial merge + ratio/mutual finalize" +
                                  ("" if gates is None else " + stereo gates")},
            "verified": verified,
            "device": devinfo["name"],
        }
        if cpu_rec is not None:
            out["cpu_baseline"] = cpu_rec
        if repeated is not None:
            repeated["rotating_over_repeated"] = out["value"] / repeated["value"]
            out["one_repeated_batch"] = repeated

    for eb in extra_bms:
        eb.close()
    bm.close()
    # ---- secondary records (N = 1): the other single-GPU configurations, timed by this same command ---------------------
    if rank == 0 and world == 1 and not args.no_secondary and not use_dist:
        note("secondary records ...")
        out["secondary"] = secondary_records(ctx, dev, args, note, cpu_tables if out is not None else None, streams=bm.streams)
    elif rank == 0 and config4 is not None:
        out["secondary"] = {"config4_strong": config4}
    if rank == 0:
        # The driver parses the LAST stdout line and keeps only a bounded tail of stdout (round 5's 22 kB line came back
        # unparsed): the line is the contract's keys and nothing else; every secondary record, distribution and prose note
        # goes to a side file whose path the line names.
        side = args.full_json or os.path.join(_ROOT, "gpurun_out", "bench_full.json" if world == 1 and not use_dist else f"bench_full_n{world}.json")
        try:
            os.makedirs(os.path.dirname(side), exist_ok=True)
            with open(side, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:                         # (a read-only tree: the line still goes out)
            note(f"side file {side} not written: {e}")
            side = None
        line = json.dumps(compact_line(out, side and os.path.relpath(side, _ROOT)), separators=(",", ":"))
        if len(line) >= COMPACT_LINE_CAP:
            raise SystemExit(f"bench line of {len(line)} bytes: the driver parses at most {COMPACT_LINE_CAP}")
        os.write(real_stdout, (line + "\n").encode())

    ctx.close()
    if use_dist:
        dist.destroy_process_group()


def secondary_records(ctx, dev, args, note, main_tables=None, streams=None):
    """Driver-timed numbers for the other BASELINE configurations and kernel forms (they existed only as
    builder-run files).  Each is a short run: a few hundred ms of GPU time."""
    import torch
    import plslam_amd
    from oracle import oracle as O
    from plslam_amd import frontend, synth
    rec = {}

    def timed(bm, steps, warm=2, reps=1):
        """reps > 1: the median of `reps` timed stretches (short steps: a stretch of a few milliseconds sees the clocks settle)"""
        for k in range(warm):
            bm.run_overlapped(k)
        bm.synchronize_all()
        dts = []
        for r_ in range(reps):
            t0 = time.perf_counter()
            for k in range(steps):
                bm.run_overlapped(warm + r_ * steps + k)
            bm.synchronize_all()
            dts.append(time.perf_counter() - t0)
        dt = float(np.median(dts))
        timed.stretches = [steps / d for d in dts]            # steps per second of every stretch
        p0 = bm.plans[0]
        p0.set_profiling(True)
        p0.elapsed()
        for _ in range(3):
            p0.run(bm.streams[0].cuda_stream)
            bm.streams[0].synchronize()
        a_, b_, n_ = p0.elapsed()
        p0.set_profiling(False)
        return dt, a_ / max(n_, 1), b_ / max(n_, 1)

    def check_all(bm, st, n_orb, n_lbd, nnr_p, nnr_l, ref=None):
        """EVERY pair of the batch, both output buffers, against the oracle (all usable cores)."""
        if ref is None:
            ref, _ = oracle_tables(st, n_orb, n_lbd, nnr_p, nnr_l)
        for b_, tab in enumerate(bm.tables):
            got = tab.cpu().numpy()
            if not np.array_equal(got, ref):
                bad = np.argwhere(got != ref)
                raise SystemExit(f"secondary record: buffer {b_} differs from the oracle in {len(bad)} entries, first at pair {bad[0][0]}")
        return ref

    def pairs_run(tag, n_orb, n_lbd, pairs, steps, opts, workload, nnr_l=None, ref=None, with_gates=False, cpu_rate=False, warm=2,
                  reps=1):
        nnr_l = args.nnr_l if nnr_l is None else nnr_l
        for k, v in opts.items():
            ctx.set_option(k, v)
        try:
            st = synth.stereo_stream(pairs, n_orb, n_lbd, seed=synth.SEED0)
            geo_ = synth.stereo_geometry(st, first_pair=0) if with_gates else None
            bm = frontend.StereoBatchMatcher(ctx, st, nnr_p=args.nnr_p, nnr_l=nnr_l, mutual=True, device=dev, n_buffers=2,
                                             geometry=geo_, gates=dict(synth.KITTI_GATES) if with_gates else None, streams=streams)
            info = bm.plan.info()
            dt, scan_ms, post_ms = timed(bm, steps, warm, reps)
            stretches = [pairs * x for x in timed.stretches]
            t0 = time.perf_counter()
            check_all(bm, st, n_orb, n_lbd, args.nnr_p, nnr_l, ref)
            cpu_dt = time.perf_counter() - t0
            bm.close()
        finally:
            for k in opts:
                ctx.set_option(k, 0)
        gbs = info["algorithmic_bytes"] / (scan_ms / 1e3) / 1e9
        rec[tag] = {"metric": f"stereo pairs/sec ({n_orb} ORB + {n_lbd} LBD BF-match)", "value": pairs * steps / dt,
                    "unit": "stereo pairs/s", "workload": workload, "pairs_per_step": pairs, "steps": steps,
                    "scan_variant": info["scan_variant"], "scan_kernel_ms": scan_ms, "post_scan_ms": post_ms,
                    "hbm_roofline_frac": gbs / HBM_PEAK_GBS, "hbm_algorithmic_GBps": gbs,
                    "verified": f"all {pairs} pairs x 4 problems, both output buffers, bit-exact vs the oracle"}
        if reps > 1:
            rec[tag]["stretches_pairs_per_s"] = stretches
            rec[tag]["how"] = f"median of {reps} stretches of {steps} steps after {warm} warm-up steps"
        if cpu_rate and ref is None:
            rec[tag]["cpu_oracle_all_cores_pairs_per_s"] = pairs / cpu_dt     # (includes packing the inputs: a lower bound)
            rec[tag]["cpu_cores"] = usable_cpus()
        note(f"  {tag}: {rec[tag]['value']:.0f} pairs/s, scan {scan_ms:.3f} ms")

    def gather_1rank(tags_wires, pairs, steps, warm=2, reps=1):
        """Config 4's per-GPU shard as rank 0 of N > 1 runs it: a (forced) one-rank RCCL process group around
        shard_gather_record -- the very stepping of the N > 1 main loop, the plain step measured beside it in the same run."""
        import torch.distributed as dist
        if dist.is_initialized():
            return
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29537")
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        except Exception as e:                         # (no RCCL in this process: the record says so instead of failing the line)
            for tag, _, _ in tags_wires:
                rec[tag] = {"skipped": f"RCCL process group unavailable: {type(e).__name__}"}
            return
        try:
            for tag, wire, comm in tags_wires:
                # (streams of its own, created AFTER the process group -- the order of an N > 1 run, where the group comes up
                # before any matcher: with the headline's older stream pair the collective's internal stream can land on the
                # hardware queue of one of them, and the step then runs at 0.62-0.69 of the plain step instead of 0.98-0.99)
                rec[tag] = shard_gather_record(ctx, dev, args, 1, 0, pairs, wire, comm, steps, warm, reps, note, tag, streams=None)
                if "strong_512" in rec and "value" in rec[tag]:
                    rec[tag]["over_strong_512"] = rec[tag]["value"] / rec["strong_512"]["value"]
        finally:
            dist.destroy_process_group()

    n_orb, n_lbd, B = args.n_orb, args.n_lbd, args.pairs_per_gpu
    small = max(64, min(512, B))
    # (round 3's "tables_only" record -- the main workload without the gate stage, one repeated batch, six steps incl. the
    # pipeline fill -- is gone: its stepping was not the headline's, so the two numbers said nothing about the gates' cost;
    # kernel_ms.post_scan_stages of the main record is the gate-inclusive time of the stages behind the scan)
    # (round 3 timed 12 steps -- 5 ms -- behind 2 warm-up steps here: mostly the pipeline's fill and the clocks' ramp; a stream of
    # 512-pair steps settles after a few tens of them)
    pairs_run("strong_512", n_orb, n_lbd, 512, 150, {}, "the per-GPU shard of BASELINE config 4 (4096 pairs over 8 GPUs = 512 per GPU "
              "per step), gate stage included: the single-GPU rate at that step size", with_gates=True, warm=10, reps=3)
    gather_1rank((("strong_512_gather_1rank", "int16", "stage"), ("strong_512_gather_1rank_int32", "int32", "stage")),
                 512, 150, warm=10, reps=3)
    pairs_run("c1_substitute", 800, 100, min(B, 4096), 6, {}, "C1 substitute (SURVEY 8d): KITTI-00-shaped descriptor-level replay, "
              "800 ORB + 100 LBD per image (config_kitti.yaml:62,71), nnr_p 0.75, nnr_l 0.9, mutual; the reference's own "
              "plslam_dataset run cannot be built in this image", nnr_l=0.9, cpu_rate=True)
    pairs_run("popcount_u32_symmetric", n_orb, n_lbd, small, 4, {"scan_variant": plslam_amd.SCAN_SYMMETRIC},
              "XOR + popcount symmetric scan (K1b/K1b'), no matrix cores")
    pairs_run("popcount_u32_north_star_literal", n_orb, n_lbd, small, 3, {"scan_variant": plslam_amd.SCAN_WAVE_PER_QUERY},
              "north_star's literal kernel: train tile in LDS, wavefront-per-query popcount, wave best-2 reduce (K1d)")
    pairs_run("c5", 4000, 600, 128, 4, {"scan_variant": plslam_amd.SCAN_MFMA},
              "C5: 4000 ORB + 600 LBD per image (two 2048-column windows per ORB scan)")

    # ---- PCIe-inclusive: descriptors in (pinned) host memory, tables wanted in host memory -- never `value` ---------------
    Bp = max(16, min(256, B))
    stp = synth.stereo_stream(Bp, n_orb, n_lbd, seed=synth.SEED0)
    hp = frontend.HostStereoPipeline(ctx, Bp, n_orb, n_lbd, nnr_p=args.nnr_p, nnr_l=args.nnr_l, mutual=True, depth=3)
    for slot in range(hp.depth + 1):
        hp.fill(slot, stp)
    for k in range(4):
        hp.submit(k % (hp.depth + 1))
    hp.wait()
    nb = 16
    t0 = time.perf_counter()
    for k in range(nb):
        hp.submit(k % (hp.depth + 1))
    hp.wait()
    dt = time.perf_counter() - t0
    # the link itself: the same arena as a bare pinned -> device copy (what any encoding of these bytes is bounded by)
    pin = torch.empty(hp.arena_bytes, dtype=torch.uint8).pin_memory()
    dst = torch.empty(hp.arena_bytes, dtype=torch.uint8, device=dev)
    for _ in range(3):
        dst.copy_(pin, non_blocking=True)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(nb):
        dst.copy_(pin, non_blocking=True)
    torch.cuda.synchronize(dev)
    link = hp.arena_bytes * nb / (time.perf_counter() - t1) / 1e9
    del pin, dst
    tab = hp.tables[0].array
    slh = frontend.table_slices(n_orb, n_lbd)
    for name, d1, d2 in frontend.pair_problems(stp["orb_l"], stp["orb_r"], stp["lbd_l"], stp["lbd_r"], 0):
        if not np.array_equal(tab[0, slh[name]], O.match(d1, d2, args.nnr_p if name.startswith("orb") else args.nnr_l, True)[0]):
            raise SystemExit(f"secondary record pcie_inclusive: output differs from the oracle ({name})")
    rec["pcie_inclusive"] = {
        "value": Bp * nb / dt, "unit": "stereo pairs/s", "pairs_per_batch": Bp, "batches": nb, "in_flight": hp.depth,
        "h2d_GBps": hp.arena_bytes * nb / dt / 1e9, "d2h_GBps": Bp * hp.stride * 4 * nb / dt / 1e9,
        "h2d_link_GBps_bare_copy": link, "link_bound_pairs_per_s": link * 1e9 / (hp.arena_bytes / Bp),
        "bytes_up_per_pair": hp.arena_bytes / Bp, "bytes_down_per_pair": hp.stride * 4,
        "workload": "plslam_match_pipeline: arenas of descriptor rows in pinned host memory -> match tables in pinned host "
                    "memory; upload of batch k+1 and download of batch k-1 under the kernels of batch k (the host-side "
                    "fill of the arenas is not in the timed region: the descriptors are taken to be born there)",
        "verified": "pair 0 x 4 problems bit-exact vs the oracle"}
    hp.close()
    note(f"  pcie_inclusive: {rec['pcie_inclusive']['value']:.0f} pairs/s host-to-host, H2D {rec['pcie_inclusive']['h2d_GBps']:.1f} GB/s")

    # ---- C3: one map<->frame problem (10 000 x 1500 ORB + 2 000 x 200 LBD, mutual) and the LBA row pass --------------------
    st_ = torch.cuda.Stream(device=dev)
    s_ = st_.cuda_stream

    def ev_time(fn, iters, warm=5):
        with torch.cuda.stream(st_):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st_)
            for _ in range(iters):
                fn()
            e1.record(st_)
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    r = np.random.Generator(np.random.PCG64(31))
    frame_p = synth.random_desc(r, 1500)
    map_p = np.concatenate([synth.noisy_copy(r, frame_p)[0], synth.random_desc(r, 8500)])
    frame_l = synth.random_desc(r, 200)
    map_l = np.concatenate([synth.noisy_copy(r, frame_l)[0], synth.random_desc(r, 1800)])
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(mp=map_p, fp=frame_p, ml=map_l, fl=frame_l).items()}
    m_p = torch.empty(10000, dtype=torch.int32, device=dev)
    m_l = torch.empty(2000, dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    plan = ctx.plan([(t["mp"].data_ptr(), 10000, t["fp"].data_ptr(), 1500, 0.75, True, m_p.data_ptr(), cnt.data_ptr()),
                     (t["ml"].data_ptr(), 2000, t["fl"].data_ptr(), 200, 0.75, True, m_l.data_ptr(), cnt.data_ptr() + 4)])
    ms = ev_time(lambda: plan.run(s_), iters=200, warm=10)
    em, _ = O.match(map_p, frame_p, 0.75, True)
    el, _ = O.match(map_l, frame_l, 0.75, True)
    if not (np.array_equal(m_p.cpu().numpy(), em) and np.array_equal(m_l.cpu().numpy(), el)):
        raise SystemExit("secondary record c3: match tables differ from the oracle")
    pinfo = plan.info()
    plan.close()
    lm = synth.local_map()
    cam = plslam_amd.make_cam(**synth.EUROC)
    g = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in lm.items()}
    npt, nls = lm["pt_lm"].shape[0], lm["ls_lm"].shape[0]
    # many maps in one launch: the row kernels' streaming rate, at two footprints per kernel -- ~0.4 GB moved per launch (64
    # point maps / 256 line maps: partly absorbed by the 256 MB memory-side cache, whose write-back outlives the kernel) and
    # ~1.5 GB (256 / 1024 maps): the second one is the HBM figure, the first is reported beside it
    reps_pt, reps_ls = (64, 256), (256, 1024)

    n_pt_lm, n_ls_lm = int(lm["Xw"].shape[0]), int(lm["Lw"].shape[0])

    def rows(kind, n, nrep):
        """One launch over `nrep` maps.  Every replica has its OWN landmark array (indices offset per replica), so the batch
        streams from HBM like nrep different maps would (round 2 shared one 240 kB Xw among 64 replicas: an L2 test)."""
        key = ("obs_uv", "pt_lm", "pt_kf") if kind == "pt" else ("l_obs", "ls_lm", "ls_kf")
        lmk, nlm = ("pt_lm", n_pt_lm) if kind == "pt" else ("ls_lm", n_ls_lm)
        big = {k: torch.cat([g[k]] * nrep) for k in key if k != lmk}
        big[lmk] = torch.cat([g[lmk] + r_ * nlm for r_ in range(nrep)])
        X = torch.cat([g["Xw" if kind == "pt" else "Lw"]] * nrep)
        nb = n * nrep
        Jp = torch.empty((nb, 6), dtype=torch.float64, device=dev)
        Jl = torch.empty((nb, 3 if kind == "pt" else 6), dtype=torch.float64, device=dev)
        rr = torch.empty(nb, dtype=torch.float64, device=dev)
        ww = torch.empty(nb, dtype=torch.float64, device=dev)
        if kind == "pt":
            fn = lambda: ctx.lba_point_rows_dev(cam, 1e-7, g["T_kf_w"].data_ptr(), X.data_ptr(), big["obs_uv"].data_ptr(),  # noqa: E731
                                                big["pt_lm"].data_ptr(), big["pt_kf"].data_ptr(), nb, Jp.data_ptr(),
                                                Jl.data_ptr(), rr.data_ptr(), ww.data_ptr(), s_, n_pose_slots=int(g["T_kf_w"].shape[0]))
        else:
            fn = lambda: ctx.lba_line_rows_dev(cam, 1e-7, False, g["T_kf_w"].data_ptr(), X.data_ptr(),  # noqa: E731
                                               big["l_obs"].data_ptr(), big["ls_lm"].data_ptr(), big["ls_kf"].data_ptr(), nb,
                                               Jp.data_ptr(), Jl.data_ptr(), rr.data_ptr(), ww.data_ptr(), s_,
                                               n_pose_slots=int(g["T_kf_w"].shape[0]))
        return ev_time(fn, iters=30 if nrep > 1 else 200, warm=5)
    ms_p1, ms_l1 = rows("pt", npt, 1), rows("ls", nls, 1)
    ms_pb0, ms_lb0 = rows("pt", npt, reps_pt[0]), rows("ls", nls, reps_ls[0])
    ms_pb, ms_lb = rows("pt", npt, reps_pt[1]), rows("ls", nls, reps_ls[1])
    torch.cuda.empty_cache()
    # bytes the kernels really move per row (lba.hip): two int32 indices + the observation + the output row, plus every
    # landmark once (24 / 48 B shared by its observations); SURVEY 8(d)'s model prices the reference's 24-byte Vector6i and
    # one landmark read per ROW: 152 / 208 B -- reported beside it, labelled as the model
    moved_pt = 8 + 16 + 88 + 24.0 * n_pt_lm / npt
    moved_ls = 8 + 24 + 112 + 48.0 * n_ls_lm / nls

    def stream_rec(nrows, ms_b, moved, model, cache_resident=False):
        """frac_of_hbm_peak only for the footprint that defeats the 256 MB memory-side cache; SURVEY 8(d)'s per-row MODEL (the
        reference's 24-byte index records, one landmark read per row) is given as bytes only -- a rate computed from bytes
        the kernel does not move is not a rate, and could exceed the peak."""
        d = {"rows": nrows, "bytes_per_row_moved": moved, "GBps_moved": nrows * moved / (ms_b * 1e-3) / 1e9,
             "bytes_per_row_survey_model": model, "ms_per_launch": ms_b}
        if cache_resident:
            d["cache_resident"] = "~0.4 GB per launch: partly absorbed by the memory-side cache -- not an HBM rate, no roofline fraction"
        else:
            d["frac_of_hbm_peak"] = nrows * moved / (ms_b * 1e-3) / (HBM_PEAK_GBS * 1e9)
        return d
    rec["c3"] = {
        "workload": "C3: one local map against one frame -- 10 000 x 1500 ORB + 2 000 x 200 LBD mutual match "
                    "(mapHandler.cpp:532-752) and the LBA row pass over 50 000 point + 10 000 line observations (:1358-1540)",
        "match_us": 1e3 * ms, "match_scan_variant": pinfo["scan_variant"], "match_directed_evals": pinfo["directed_evals"],
        "match_verified": "both tables bit-exact vs the oracle",
        "lba_rows_pass_us": 1e3 * (ms_p1 + ms_l1), "lba_rows_pass_bytes": npt * 152 + nls * 208,
        "lba_point_rows_streaming": dict(stream_rec(npt * reps_pt[1], ms_pb, moved_pt, 152),
                                         at_0p4_GB_per_launch=stream_rec(npt * reps_pt[0], ms_pb0, moved_pt, 152, cache_resident=True)),
        "lba_line_rows_streaming": dict(stream_rec(nls * reps_ls[1], ms_lb, moved_ls, 208),
                                        at_0p4_GB_per_launch=stream_rec(nls * reps_ls[0], ms_lb0, moved_ls, 208, cache_resident=True)),
        "note": "one map = one launch of 9.7 MB: launch-bound (replicas only, SURVEY 8e); the streaming figures batch 256 point / "
                "1024 line maps (each with its own landmark array: ~1.5 GB moved) per launch to show the row kernels' HBM rate, "
                "and a quarter of that beside it (~0.4 GB: the memory-side cache flatters it); frac_of_hbm_peak is computed "
                "from the bytes the kernels move (indices 8 B, not the reference's 24-byte Vector6i; landmarks once), "
                "profiles/r4_*_lba_* hold the rocprofv3 kernel trace and FETCH_SIZE / WRITE_SIZE passes of the same launches"}
    note(f"  c3: match {1e3 * ms:.1f} us, rows {rec['c3']['lba_point_rows_streaming']['GBps_moved']:.0f} / "
         f"{rec['c3']['lba_line_rows_streaming']['GBps_moved']:.0f} GB/s moved")
    # ---- the other section-8 rows (bench_rows.py): each verified over everything it produced -----------------
    import bench_rows as R
    for tag, fn in (("grid", lambda: R.grid(ctx, dev, torch, O, st_)), ("drivers", lambda: R.drivers(ctx, O, dev)),
                    ("lba_plan_iterate_dev", lambda: R.lba_iterate(ctx, O)), ("lbd", lambda: R.lbd(ctx, dev, torch, O, st_)),
                    ("median_desc", lambda: R.median_desc(ctx, dev, torch, O, st_))):
        t0 = time.perf_counter()
        rec[tag] = fn()
        note(f"  {tag}: done in {time.perf_counter() - t0:.1f} s")
    return rec


if __name__ == "__main__":
    main()
