#!/usr/bin/env python3
"""bench.py -- stereo pairs/sec of the PL-SLAM matching front end on MI355X.

A "step" is one pass of the hot path over one device-resident batch of synthetic stereo pairs
(BASELINE.json config 2: 752x480-shaped stream, 1500 ORB + 200 LBD per image; per pair
ORB L<->R, ORB prev<->curr, LBD L<->R, LBD prev<->curr, each a mutual + ratio StVO::match).
With N > 1 ranks (one per GPU, torch.distributed 'nccl' == RCCL) every rank runs its own shard of
pairs (weak scaling) and the per-pair match tables are gathered to rank 0 inside the step.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

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


def cpu_baseline(stream, n_orb, n_lbd, nnr_p, nnr_l, budget_s=15.0):
    """The CPU restatement of the reference path (oracle, -O3 -march=native, popcnt) timed on this
    host's cores on a bounded sample of the SAME workload.  kind = "port"."""
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
        O.match_batched(a, oa, b, ob, nnr_p, True, nthreads=threads, L=L)
        O.match_batched(c, oc, d, od, nnr_l, True, nthreads=threads, L=L)
        return time.perf_counter() - t0

    B = stream["orb_l"].shape[0] - 1
    n_1 = min(B, 8)
    one = problems(list(range(n_1)))
    t_1, reps_1 = 0.0, 0
    while t_1 < 0.25 * budget_s:                        # single-thread leg, bounded by wall clock
        t_1 += run(one, 1)
        reps_1 += 1
    full = problems(list(range(B)))
    t_mt, reps = 0.0, 0
    while t_mt < 0.75 * budget_s:                       # all-cores leg: whole passes over the batch
        t_mt += run(full, cores)
        reps += 1
    return {"value": B * reps / t_mt, "unit": "stereo pairs/s", "cores": cores, "kind": "port",
            "host_logical_cpus": os.cpu_count(),
            "sample": f"{reps} pass(es) over the same {B}-pair batch on {cores} threads ({t_mt:.1f} s); "
                      f"1 thread: {reps_1} pass(es) over {n_1} pairs ({t_1:.1f} s)",
            "value_1thread": n_1 * reps_1 / t_1}


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
    ap.add_argument("--mfma-form", type=int, default=0, help="0 = auto (K1f), 1 = K1e (best-2 push per tile), 2 = K1f (group minima)")
    ap.add_argument("--fuse", type=int, default=0, help="K1f: 0 = auto, 1 = never, 2 = always one workgroup per problem incl. merge + finalize")
    ap.add_argument("--step-streams", type=int, default=2, help="output buffers / HIP streams the steps alternate over")
    ap.add_argument("--no-overlap", action="store_true",
                    help="single GPU: run the steps strictly one after another on one stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the NCCL(RCCL) process group and run the table gather even with one rank")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    args = ap.parse_args()

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
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 "
                             "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B = args.pairs_per_gpu
    # weak scaling: rank r owns pairs [r*B, (r+1)*B) of one global stream (with a one-pair halo)
    stream = synth.stereo_stream(B, args.n_orb, args.n_lbd, seed=synth.SEED0, first_pair=rank * B)

    note(f"synthetic stream of {B} pairs generated")
    ctx = plslam_amd.Context(local_rank)     # raises if libplslam_hip.so / a gfx950 device is missing
    if args.scan_variant:
        ctx.set_option("scan_variant", args.scan_variant)
    if args.scan_block:
        ctx.set_option("scan_block", args.scan_block)
    if args.sym_rows:
        ctx.set_option("sym_rows", args.sym_rows)
    if args.group_cap:
        ctx.set_option("group_cap", args.group_cap)
    if args.mfma_form:
        ctx.set_option("mfma_form", args.mfma_form)
    if args.fuse:
        ctx.set_option("fuse", args.fuse)
    overlap = not use_dist and not args.no_overlap
    bm = frontend.StereoBatchMatcher(ctx, stream, nnr_p=args.nnr_p, nnr_l=args.nnr_l, mutual=True, device=dev,
                                     n_buffers=max(2, args.step_streams) if (use_dist or overlap) else 1)
    info = bm.plan.info()
    devinfo = ctx.device_info()
    # N > 1: the table of step k is gathered to rank 0 over RCCL on a communication stream while
    # step k+1 computes into the other table buffer (steps are independent batches of a stream).
    pg = frontend.PipelinedGather(bm, world, rank, root=0) if use_dist else None
    step_no = [0]

    def step():
        if pg is not None:
            pg.step(step_no[0])
        elif overlap:
            # consecutive steps are independent batches: alternate two output buffers / HIP streams so
            # the next scan's ramp-up fills the CUs that idle in this step's drain, merge and finalize
            bm.run_overlapped(step_no[0])
        else:
            bm.run()
        step_no[0] += 1

    def sync():
        if pg is not None:
            pg.finish()
        if overlap:
            bm.synchronize_all()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    note(f"plan built: {info}")
    for _ in range(args.warmup):
        step()
    sync()
    note("warmup done")
    for p_ in bm.plans:
        p_.set_profiling(True)
        p_.elapsed()                          # reset the accumulators
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    scan_ms = fin_ms = 0.0
    runs = 0
    for p_ in bm.plans:
        a_, b_, n_ = p_.elapsed()
        scan_ms, fin_ms, runs = scan_ms + a_, fin_ms + b_, runs + n_
        p_.set_profiling(False)
    note(f"timed {args.steps} steps in {elapsed:.4f}s; scan {scan_ms / max(runs, 1):.3f} ms/launch")

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

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
        note(f"exclusive kernel times: scan {excl_scan_ms:.3f} ms, merge+finalize {excl_fin_ms:.3f} ms")

    # full-size determinism check: every output buffer was computed from the same inputs (under
    # overlap / contention), so whole tables and count arrays must be bit-identical
    for b_ in range(1, len(bm.tables)):
        if args.steps + args.warmup >= 2 and not (torch.equal(bm.tables[0], bm.tables[b_]) and
                                                  torch.equal(bm.count_bufs[0], bm.count_bufs[b_])):
            raise SystemExit(f"rank {rank}: output buffers 0 and {b_} differ (nondeterministic result)")
    # self-check of the timed output against the oracle (the checker, not the product): pair 0 of
    # EVERY rank's table as it arrived on rank 0 (N > 1: through the RCCL gather, both buffers)
    if rank == 0:
        from oracle import oracle as O
        sl = frontend.table_slices(args.n_orb, args.n_lbd)
        bufs = range(len(bm.tables)) if (pg is not None or overlap) and args.steps + args.warmup >= 2 else [0]
        for b_ in bufs:
            full = (pg.gathered(b_) if pg is not None else bm.tables[b_]).cpu().numpy()
            for r_ in range(world if pg is not None else 1):
                st_r = stream if r_ == 0 else synth.stereo_stream(1, args.n_orb, args.n_lbd, seed=synth.SEED0,
                                                                  first_pair=r_ * B)
                tab = full[r_ * B]
                for name, d1, d2 in frontend.pair_problems(st_r["orb_l"], st_r["orb_r"], st_r["lbd_l"],
                                                           st_r["lbd_r"], 0):
                    em, _ = O.match(d1, d2, args.nnr_p if name.startswith("orb") else args.nnr_l, True)
                    if not np.array_equal(tab[sl[name]], em):
                        raise SystemExit(f"bench output differs from the oracle: rank {r_} buffer {b_} pair 0 / {name}")
        note(f"output verified against the oracle for {world if pg is not None else 1} rank(s)")

    if rank == 0:
        pairs_total = B * world * args.steps
        # duration of the dominant kernel: exclusive (serial launches) for the roofline; the in-region
        # figure (overlapped with the neighbouring step) is reported next to it
        scan_s = excl_scan_ms / 1e3
        achieved_gbs = info["algorithmic_bytes"] / scan_s / 1e9
        valu_peak = devinfo["cu_count"] * VALU_LANES_PER_CLK_PER_CU * devinfo["clock_khz"] * 1e3
        # HBM bytes of the dominant kernel per launch: PMC counters cannot be read from inside this
        # process, so the figure comes from the committed rocprofv3 passes of this same command
        # (profiles/pmc_traffic.json) and is reported only for the configuration they were taken on.
        traffic = None
        executed = None
        mfma = info["scan_variant"] == 4
        wkey = (f"C2:{args.n_orb}+{args.n_lbd}:pairs{B}:v{info['scan_variant']}:sym{ctx.get_option('sym_rows')}"
                f":cap{ctx.get_option('group_cap')}")
        # (sym_rows 0 = auto: resolved per plan, reported in config.scan_block_threads: 64 => 4 rows/lane)
        pm = None
        try:
            with open(os.path.join(_ROOT, "profiles", "pmc_traffic.json")) as f:
                pm = json.load(f).get("entries", {}).get(wkey)
        except OSError:
            pass
        if pm:
            traffic = pm["traffic_bytes_per_launch"]
            valu_insts = pm["sq_insts_valu_per_launch"] - pm.get("sq_insts_mfma_per_launch", 0)
            lane_ops = valu_insts * 64 / (excl_scan_ms / 1e3)
            executed = {"lane_ops_per_s": lane_ops, "sq_insts_valu_per_launch": valu_insts,
                        "sq_insts_mfma_per_launch": pm.get("sq_insts_mfma_per_launch", 0),
                        "measured_ceiling_lane_ops_per_s": pm["measured_int_valu_ceiling_lane_ops_per_s"],
                        "frac_of_measured_ceiling": lane_ops / pm["measured_int_valu_ceiling_lane_ops_per_s"],
                        "note": "executed wave64 VALU instructions, MFMAs excluded (PMC, profiles/pmc_traffic.json) x 64 / "
                                "exclusive scan time, against the issue rate measured for this integer instruction mix "
                                "(16 lanes/clk/SIMD)"}
        kernel_name = {4: "k_scan_sym_mfma", 3: "k_scan_symmetric" + ("_r4" if info["scan_block_threads"] == 64 else ""),
                       2: "k_scan_wave_per_query", 1: "k_scan_lane_per_query"}.get(info["scan_variant"], "k_scan")
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
            # K1e: distances come from v_mfma_scale_f32_32x32x64_f8f6f4 over fp4 +-1 codes; 256 multiply-accumulates =
            # 512 ops per executed distance
            mx_peak = devinfo["cu_count"] * 4 * FP4_MFMA_OPS_PER_CLK_PER_SIMD * devinfo["clock_khz"] * 1e3
            mfma_ops = info["distance_evals"] * 512
            roofline = {
                "bound": "mfma", "achieved": mfma_ops / scan_s / 1e12, "peak": mx_peak / 1e12, "unit": "TFLOP/s",
                "frac": mfma_ops / scan_s / mx_peak, "traffic": traffic, "kernel": kernel_name,
                "kernel_ms": 1e3 * scan_s, "kernel_ms_in_timed_region": scan_ms / max(runs, 1), "timing": timing_note,
                "algorithmic_ops_per_launch": mfma_ops,
                "note": "block-scaled fp4 MFMA (operands are the e2m1 codes of +-1, fp32 accumulation of integers below "
                        "2^24: exact); dense MX-fp4 peak = CUs x 4 SIMDs x 4096 ops/clk x max clock (MI355X_MICROARCH.md "
                        "measures 9099 T for the 32x32x64 shape); 512 ops per executed 256-bit distance (each serves both "
                        "match directions).  The kernel is co-limited by the VALU best-2 bookkeeping that consumes the "
                        "accumulators: see valu_roofline.executed",
            }
        else:
            roofline = hbm_roofline
        out = {
            "metric": "stereo pairs/sec (1500 ORB + 200 LBD BF-match)",
            "value": pairs_total / elapsed,
            "unit": "stereo pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp4" if mfma else "u32",
            "data": "synthetic",
            "config": {
                "workload": f"C2: synthetic 752x480 stereo stream, {args.n_orb} ORB + {args.n_lbd} LBD per image, "
                            "ORB+LBD L<->R and prev<->curr, mutual + ratio (StVO::match), device-resident",
                "pairs_per_gpu_per_step": B, "nnr_p": args.nnr_p, "nnr_l": args.nnr_l, "mutual": True,
                "scan_variant": info["scan_variant"], "scan_block_threads": info["scan_block_threads"],
                "parallelism": f"pairs sharded over {world} rank(s); per-step RCCL gather of the match tables "
                               "to rank 0, overlapped with the next step" if use_dist else
                               ("single GPU; consecutive steps alternate two output buffers / HIP streams" if overlap
                                else "single GPU"),
            },
            "roofline": roofline,
            "hbm_roofline": hbm_roofline,
            "valu_roofline": {
                "bound": "valu-int", "achieved": info["directed_evals"] * 16 / scan_s / 1e12,
                "peak": valu_peak / 1e12, "unit": "T lane-ops/s",
                "frac": info["directed_evals"] * 16 / scan_s / valu_peak,
                "note": "16 algorithmic lane-ops (8 xor + 8 bcnt) per 256-bit distance x directed distances the "
                        "reference evaluates, priced as if done on the VALU; peak = CUs x 128 lanes/clk x max clock "
                        "(frac > 1 is possible when the distances come from the matrix cores)",
                "evals_per_launch": info["directed_evals"], "executed_evals_per_launch": info["distance_evals"],
                "executed": executed,
            },
            "kernel_ms": {"scan": excl_scan_ms, "merge+finalize": excl_fin_ms,
                          "scan_in_timed_region": scan_ms / max(runs, 1),
                          "merge+finalize_in_timed_region": fin_ms / max(runs, 1)},
            "device": devinfo["name"],
        }
        if world == 1 and not args.no_cpu_baseline:
            note("cpu baseline ...")
            out["cpu_baseline"] = cpu_baseline(stream, args.n_orb, args.n_lbd, args.nnr_p, args.nnr_l,
                                               args.cpu_budget_s)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    bm.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
