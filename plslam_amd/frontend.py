"""Batch stereo front end: the per-frame matching work of StVO::StereoFrame (L<->R association)
and StVO::StereoFrameHandler::f2fTracking (prev<->curr) [stvo-pl; driven from
app/plslam_dataset.cpp:127 of the reference] for a batch of independent stereo pairs, kept
device-resident and executed through one match plan of the C-ABI library.

Per pair four StVO::match() problems are solved:
    ORB  L   -> R      (nnr_p)      table columns [0, n_orb)
    ORB  prev-> curr   (nnr_p)                    [n_orb, 2 n_orb)
    LBD  L   -> R      (nnr_l)                    [2 n_orb, 2 n_orb + n_lbd)
    LBD  prev-> curr   (nnr_l)                    [2 n_orb + n_lbd, 2 n_orb + 2 n_lbd)
so a pair's match table is a fixed-stride int32 row (13.6 kB at 1500 ORB + 200 LBD).

This file is plumbing (torch owns the device memory and the process group); the arithmetic is in
plslam_amd/csrc.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_pairs: int, world: int, rank: int):
    """Contiguous blocks of ceil(n/world) pairs per rank."""
    per = -(-n_pairs // world)
    lo = min(rank * per, n_pairs)
    return lo, min(lo + per, n_pairs)


def table_stride(n_orb: int, n_lbd: int) -> int:
    return 2 * n_orb + 2 * n_lbd


def table_slices(n_orb: int, n_lbd: int):
    o = 0
    out = {}
    for name, n in (("orb_lr", n_orb), ("orb_pc", n_orb), ("lbd_lr", n_lbd), ("lbd_pc", n_lbd)):
        out[name] = slice(o, o + n)
        o += n
    return out


def pair_problems(orb_l, orb_r, lbd_l, lbd_r, i):
    """The four (d1, d2) descriptor pairs of stereo pair i (arrays carry the halo at index 0)."""
    return (("orb_lr", orb_l[i + 1], orb_r[i + 1]), ("orb_pc", orb_l[i], orb_l[i + 1]),
            ("lbd_lr", lbd_l[i + 1], lbd_r[i + 1]), ("lbd_pc", lbd_l[i], lbd_l[i + 1]))


class StereoBatchMatcher:
    """Device-resident batch of `B` stereo pairs -> (B, stride) int32 match table.

    `n_buffers` output tables (and plans) over the same descriptors let step k+1 compute while the
    table of step k is still being gathered (bench.py, N > 1)."""

    def __init__(self, ctx, stream_np: dict, nnr_p=0.75, nnr_l=0.75, mutual=True, device=None, n_buffers=1,
                 geometry: dict | None = None, gates: dict | None = None, streams=None, scan_streams: int = 1):
        """geometry (synth.stereo_geometry: kp_l, kp_r, seg_l, seg_r per frame) + gates (the thresholds max_dist_epip,
        min_disp, line_horiz_th, stereo_overlap_th, ls_min_disp_ratio) add the gate stage of StereoFrame to every plan:
        each run then also fills `stereo` (B, n_orb + n_lbd) int32 -- the L<->R associations that survive the epipolar /
        disparity / overlap gates -- `stereo_disp` (B, n_orb + 2 n_lbd) float64 and `stereo_counts` (B, 2)."""
        import torch
        self.torch = torch
        self.ctx = ctx
        dev = device if device is not None else torch.device("cuda", ctx.device)
        self.dev = dev
        self.B = stream_np["orb_l"].shape[0] - 1
        self.n_orb = stream_np["orb_l"].shape[1]
        self.n_lbd = stream_np["lbd_l"].shape[1]
        self.stride = table_stride(self.n_orb, self.n_lbd)
        self.d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in stream_np.items()
                  if k in ("orb_l", "orb_r", "lbd_l", "lbd_r")}
        self.g = None
        if geometry is not None:
            assert gates is not None, "thresholds required (e.g. synth.KITTI_GATES)"
            self.g = {k: torch.from_numpy(np.ascontiguousarray(geometry[k], np.float32)).to(dev)
                      for k in ("kp_l", "kp_r", "seg_l", "seg_r")}
        sl = table_slices(self.n_orb, self.n_lbd)
        row_o, row_l = self.n_orb * 32, self.n_lbd * 32
        ol, orr = self.d["orb_l"].data_ptr(), self.d["orb_r"].data_ptr()
        ll, lr = self.d["lbd_l"].data_ptr(), self.d["lbd_r"].data_ptr()
        self.tables, self.count_bufs, self.plans = [], [], []
        self.stereo_tabs, self.stereo_disps, self.stereo_cnts = [], [], []
        for _ in range(n_buffers):
            table = torch.full((self.B, self.stride), -2, dtype=torch.int32, device=dev)
            counts = torch.zeros((self.B, 4), dtype=torch.int32, device=dev)
            tb, cb = table.data_ptr(), counts.data_ptr()
            probs = []
            for i in range(self.B):
                t_i = tb + 4 * self.stride * i
                c_i = cb + 16 * i
                probs.append((ol + row_o * (i + 1), self.n_orb, orr + row_o * (i + 1), self.n_orb, nnr_p, mutual,
                              t_i + 4 * sl["orb_lr"].start, c_i))
                probs.append((ol + row_o * i, self.n_orb, ol + row_o * (i + 1), self.n_orb, nnr_p, mutual,
                              t_i + 4 * sl["orb_pc"].start, c_i + 4))
                probs.append((ll + row_l * (i + 1), self.n_lbd, lr + row_l * (i + 1), self.n_lbd, nnr_l, mutual,
                              t_i + 4 * sl["lbd_lr"].start, c_i + 8))
                probs.append((ll + row_l * i, self.n_lbd, ll + row_l * (i + 1), self.n_lbd, nnr_l, mutual,
                              t_i + 4 * sl["lbd_pc"].start, c_i + 12))
            self.tables.append(table)
            self.count_bufs.append(counts)
            plan = ctx.plan(probs)
            if self.g is not None:
                # the stereo gates over the L<->R tables of this buffer: points then lines of pair i
                st = torch.full((self.B, self.n_orb + self.n_lbd), -2, dtype=torch.int32, device=dev)
                sd = torch.zeros((self.B, self.n_orb + 2 * self.n_lbd), dtype=torch.float64, device=dev)
                sc = torch.zeros((self.B, 2), dtype=torch.int32, device=dev)
                kl, kr, gl, gr = (self.g[k].data_ptr() for k in ("kp_l", "kp_r", "seg_l", "seg_r"))
                glist = []
                for i in range(self.B):
                    t_i = tb + 4 * self.stride * i
                    s_i = st.data_ptr() + 4 * (self.n_orb + self.n_lbd) * i
                    d_i = sd.data_ptr() + 8 * (self.n_orb + 2 * self.n_lbd) * i
                    glist.append(dict(matches_12=t_i + 4 * sl["orb_lr"].start, f_l=kl + 8 * self.n_orb * (i + 1),
                                      f_r=kr + 8 * self.n_orb * (i + 1), n_l=self.n_orb, n_r=self.n_orb, lines=0,
                                      stereo_12=s_i, disp=d_i, n_stereo=sc.data_ptr() + 8 * i, **gates))
                    glist.append(dict(matches_12=t_i + 4 * sl["lbd_lr"].start, f_l=gl + 16 * self.n_lbd * (i + 1),
                                      f_r=gr + 16 * self.n_lbd * (i + 1), n_l=self.n_lbd, n_r=self.n_lbd, lines=1,
                                      stereo_12=s_i + 4 * self.n_orb, disp=d_i + 8 * self.n_orb,
                                      n_stereo=sc.data_ptr() + 8 * i + 4, **gates))
                plan.add_stereo_gates(glist)
                self.stereo_tabs.append(st)
                self.stereo_disps.append(sd)
                self.stereo_cnts.append(sc)
            self.plans.append(plan)
        self.table, self.counts, self.plan = self.tables[0], self.count_bufs[0], self.plans[0]
        if self.g is not None:
            self.stereo, self.stereo_disp, self.stereo_counts = self.stereo_tabs[0], self.stereo_disps[0], self.stereo_cnts[0]
        # A real (non-NULL) HIP stream: the C ABI reads a NULL stream as "the context's own stream",
        # and torch's legacy default stream has handle 0.
        # (`streams`: share another matcher's [scan stream, stage stream]: several batches stepping through one pipeline)
        # (PLSLAM_STREAM_PRIO, experiments: "scan_high" = the scan stream above the stage stream, "equal" = both normal)
        import os as _os
        prio = _os.environ.get("PLSLAM_STREAM_PRIO", "stage_high")
        scan_prio, stage_prio = {"stage_high": (0, -1), "scan_high": (-1, 0), "equal": (0, 0)}[prio]
        self.stream = torch.cuda.Stream(device=dev, priority=scan_prio) if streams is None else streams[0]
        # streams[1] carries the short HBM-bound stages behind a scan in run_overlapped(): high priority, so that they
        # take the workgroup slots the running scan frees instead of queueing behind its backlog
        self.streams = list(streams) if streams is not None else \
            [self.stream] + [torch.cuda.Stream(device=dev, priority=stage_prio if i == 0 else 0) for i in range(n_buffers - 1)]
        # `scan_streams` = 2: the scans of consecutive steps alternate between two streams, so the first workgroups of step
        # k+1 fill the slots the last wave of step k's scan leaves idle.  A 512-pair step is 9.3 rounds of the chip's 768
        # workgroup slots: its last round is a third full, and on one stream the next scan starts only when it is over.
        self.scan_streams = [self.streams[0]] + [torch.cuda.Stream(device=dev) for _ in range(max(1, scan_streams) - 1)]
        self.wire16 = None                 # enable_wire16(): per-buffer int16 mirrors of the tables, written by the finalize kernel

    def enable_wire16(self, on: bool = True):
        """int16 mirrors of the match tables (the N > 1 gather's wire format), stored by the finalize kernel beside the int32
        entries (plslam_match_plan_set_wire16): no narrowing pass between the step and its gather."""
        torch = self.torch
        if on and self.wire16 is None:
            from .capi import ENOTSUP, PlslamError
            w16 = [torch.full((self.B, self.stride), -2, dtype=torch.int16, device=self.dev) for _ in self.tables]
            try:
                for plan, t32, t16 in zip(self.plans, self.tables, w16):
                    plan.set_wire16(t32.data_ptr(), t16.data_ptr(), t32.numel())
            except PlslamError as e:
                # a plan whose tables another kernel than the finalize kernel writes (a column-split plan of a few large problems,
                # a fused plan): no mirror -- the caller narrows the table with a copy instead
                if e.code != ENOTSUP:
                    raise
                for plan, t32 in zip(self.plans, self.tables):
                    plan.set_wire16(t32.data_ptr(), 0, t32.numel())
                return False
            self.wire16 = w16
            return True
        elif not on and self.wire16 is not None:
            for plan, t32 in zip(self.plans, self.tables):
                plan.set_wire16(t32.data_ptr(), 0, t32.numel())
            self.wire16 = None
        return self.wire16 is not None

    def run_overlapped(self, k: int):
        """Step k of a stream of independent batches into buffer k % n_buffers: every scan on one HIP stream, the
        stages behind a scan (merge of the column partials, finalize, gates) on a second one -- they are HBM-bound and
        run under the NEXT step's scan, which is bound by instruction issue.  The plans order themselves (a plan's
        scan waits for the last stage of its previous run).  Call synchronize_all() before reading."""
        b = k % len(self.plans)
        self.plans[b].run_split(self.scan_stream_of(k).cuda_stream, self.streams[min(1, len(self.streams) - 1)].cuda_stream)
        return b

    def scan_stream_of(self, k: int):
        """The stream step k's scan runs on."""
        return self.scan_streams[k % len(self.scan_streams)]

    @property
    def stage_stream(self):
        """The stream the stages behind a scan run on in run_overlapped() (== the scan stream with one buffer)."""
        return self.streams[min(1, len(self.streams) - 1)]

    def synchronize_all(self):
        for s in list(self.streams) + list(self.scan_streams[1:]):
            s.synchronize()

    def run(self, buf: int = 0):
        """Enqueue one pass over the batch into table `buf`.  The kernels run on this object's stream,
        fenced on both sides against torch's current stream so that whatever the caller enqueues next
        is ordered after them."""
        cur = self.torch.cuda.current_stream(self.dev)
        self.stream.wait_stream(cur)
        self.plans[buf].run(self.stream.cuda_stream)
        cur.wait_stream(self.stream)
        return self.tables[buf]

    def run_async(self, buf: int = 0):
        """Enqueue one pass into table `buf` on that buffer's own stream; returns an event recorded
        after it (consecutive steps use different buffers/streams and may overlap)."""
        st = self.streams[buf]
        self.plans[buf].run(st.cuda_stream)
        ev = self.torch.cuda.Event()
        ev.record(st)
        return ev

    def close(self):
        for p in self.plans:
            p.close()


class HostStereoPipeline:
    """Host-to-host form of StereoBatchMatcher: descriptors born on the host (pinned memory), match tables wanted on the
    host, through plslam_match_pipeline (upload of batch k+1 and download of batch k-1 under the kernels of batch k).
    One batch = `B` stereo pairs; its arena holds every image's rows ONCE -- left rows of the B pairs and of the pair
    before the first (the halo), right rows of the B pairs: the prev<->curr and L<->R problems of a pair point into the
    same rows (109 kB per C2 pair)."""

    def __init__(self, ctx, B: int, n_orb: int, n_lbd: int, nnr_p=0.75, nnr_l=0.75, mutual=True, depth: int = 3):
        from .capi import MatchPipeline, PinnedArray
        self.B, self.n_orb, self.n_lbd, self.depth = B, n_orb, n_lbd, depth
        self.stride = table_stride(n_orb, n_lbd)
        ro, rl = n_orb * 32, n_lbd * 32
        self.off = {"orb_l": 0, "orb_r": (B + 1) * ro, "lbd_l": (2 * B + 1) * ro, "lbd_r": (2 * B + 1) * ro + (B + 1) * rl}
        self.arena_bytes = (2 * B + 1) * (ro + rl)
        sl = table_slices(n_orb, n_lbd)
        probs = []
        for i in range(B):
            t = self.stride * i
            probs.append((self.off["orb_l"] + ro * (i + 1), n_orb, self.off["orb_r"] + ro * i, n_orb, nnr_p, mutual, t + sl["orb_lr"].start))
            probs.append((self.off["orb_l"] + ro * i, n_orb, self.off["orb_l"] + ro * (i + 1), n_orb, nnr_p, mutual, t + sl["orb_pc"].start))
            probs.append((self.off["lbd_l"] + rl * (i + 1), n_lbd, self.off["lbd_r"] + rl * i, n_lbd, nnr_l, mutual, t + sl["lbd_lr"].start))
            probs.append((self.off["lbd_l"] + rl * i, n_lbd, self.off["lbd_l"] + rl * (i + 1), n_lbd, nnr_l, mutual, t + sl["lbd_pc"].start))
        self.pipe = MatchPipeline(ctx, self.arena_bytes, probs, B * self.stride, depth)
        # `depth` + 1 pinned arena / table pairs: the host fills one while `depth` are in flight
        self.arenas = [PinnedArray(ctx, (self.arena_bytes,), np.uint8) for _ in range(depth + 1)]
        self.tables = [PinnedArray(ctx, (B, self.stride), np.int32) for _ in range(depth + 1)]

    def fill(self, slot: int, stream_np: dict, first: int = 0):
        """Copy pairs [first, first + B) of a synth.stereo_stream (halo at index `first`) into arena `slot`."""
        a, B = self.arenas[slot].array, self.B
        for key, n, cnt, lo in (("orb_l", self.n_orb, B + 1, first), ("orb_r", self.n_orb, B, first + 1),
                                ("lbd_l", self.n_lbd, B + 1, first), ("lbd_r", self.n_lbd, B, first + 1)):
            o = self.off[key]
            a[o:o + cnt * n * 32] = stream_np[key][lo:lo + cnt].reshape(-1)

    def submit(self, slot: int):
        self.pipe.submit(self.arenas[slot].array, self.tables[slot].array)

    def wait(self):
        self.pipe.wait()

    def close(self):
        self.pipe.close()
        for x in self.arenas + self.tables:
            x.close()


class TableGatherPipeline:
    """The backend-agnostic half of the N > 1 path: wire format, receive buffers and the per-buffer ordering of
    "compute into table b" -> "narrow" -> "gather to root" -> "table b may be overwritten".  It knows nothing about the
    matcher; `submit()` is handed a finished (or, on a GPU, enqueued) table.  The same code runs over RCCL (CUDA
    tensors, a communication stream, events) and over gloo (CPU tensors, synchronous): tests/test_dist_cpu.py drives it
    at world size 2.

    Wire format: a table entry is a row index of the other image or -1 (or a negative filler), so with at most 32 767
    features per image it travels as int16 (27.8 MB instead of 55.7 MB per rank and step at 4096 pairs of 1500 + 200
    features): half the bytes on every xGMI link into the root and half the root's HBM writes; `gathered()` widens
    back to the int32 tables of the C ABI.  RCCL has no 16-bit integer type and a gather moves bytes, so both sides
    are viewed as uint8."""

    def __init__(self, rows: int, stride: int, max_index: int, world: int, rank: int, root: int = 0, nbuf: int = 2,
                 device=None, group=None, compact=None, comm_stream=None):
        import torch
        self.torch, self.world, self.rank, self.root, self.group, self.nbuf = torch, world, rank, root, group, nbuf
        self.rows, self.stride = rows, stride
        self.dev = device if device is not None else torch.device("cpu")
        self.cuda = self.dev.type == "cuda"
        self.compact = (max_index <= 32767) if compact is None else bool(compact)
        self.wire = torch.int16 if self.compact else torch.int32
        # int16 wire: the narrowed copy of table b (unless the producer writes one itself: submit(..., wire16=...)).  int32 wire:
        # the table itself travels -- no send buffer, no copy (round 4 copied 55.7 MB per 4096-pair step for nothing)
        self.send = [torch.empty((rows, stride), dtype=self.wire, device=self.dev) for _ in range(nbuf)] if self.compact else None
        self.recv = [torch.empty((world, rows, stride), dtype=self.wire, device=self.dev)
                     for _ in range(nbuf)] if rank == root else [None] * nbuf
        # root, compact wire format, GPU: the int32 tables of the C ABI are rebuilt on the communication stream right behind each
        # gather (world x 27.8 MB read + 55.7 MB written per step at C4 sizes) -- inside the step, where its cost is timed
        self.wide = [torch.empty((world, rows, stride), dtype=torch.int32, device=self.dev) for _ in range(nbuf)] \
            if (rank == root and self.compact and self.cuda) else None
        # high priority: the gather's kernels are short and must take the workgroup slots the running scan frees -- at normal
        # priority they queue behind the scan's whole backlog, and the step that recomputes this buffer waits for them
        # comm_stream: run the wait for the collective and the widening on a stream the caller already has (the matcher's stage
        # stream) instead of a third one of our own
        self.comm = (comm_stream if comm_stream is not None else torch.cuda.Stream(device=self.dev, priority=-1)) if self.cuda else None
        self.done_ev = [None] * nbuf         # CUDA: recorded on the comm stream after buffer b's gather
        self.works = [None] * nbuf

    def before_overwrite(self, b: int, compute_stream=None):
        """Call before table b is recomputed: its previous gather must have read the send buffer / table."""
        if self.cuda:
            if self.done_ev[b] is not None:
                (compute_stream or self.torch.cuda.current_stream(self.dev)).wait_event(self.done_ev[b])
        elif self.works[b] is not None:
            self.works[b].wait()
            self.works[b] = None

    def submit(self, b: int, table, compute_stream=None, wire16=None):
        """Start the gather of table (rows, stride) int32 of buffer b.  int32 wire: the table itself is the send buffer (it is
        not overwritten before before_overwrite(b) has seen this gather's end).  int16 wire: `wire16` -- a (rows, stride) int16
        table the producer wrote beside the int32 one (plan.set_wire16: the finalize kernel stores both) -- travels as it is;
        without it the table is narrowed into send buffer b first.  CUDA: `table` / `wire16` were produced on `compute_stream`;
        the narrowing (if any) is enqueued there, the gather on the communication stream behind an event."""
        import torch.distributed as dist
        torch = self.torch
        narrow = False
        if self.compact and wire16 is not None:
            assert wire16.dtype == torch.int16 and tuple(wire16.shape) == (self.rows, self.stride) and wire16.is_contiguous()
            src = wire16
        elif self.compact:
            src, narrow = self.send[b], True
        else:
            assert table.dtype == torch.int32 and table.is_contiguous()
            src = table
        if self.cuda:
            st = compute_stream or torch.cuda.current_stream(self.dev)
            with torch.cuda.stream(st):
                if narrow:
                    src.copy_(table)                       # int32 -> int16 right behind the producer
                ev = torch.cuda.Event()
                ev.record(st)
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                glist = list(self.recv[b].view(torch.uint8).unbind(0)) if self.rank == self.root else None
                w = dist.gather(src.view(torch.uint8), gather_list=glist, dst=self.root, group=self.group,
                                async_op=True)
                w.wait()                                   # stream-level wait (comm stream), not a host block
                if self.wide is not None:
                    self.wide[b].copy_(self.recv[b])       # int16 -> int32, still on the communication stream
                done = torch.cuda.Event()
                done.record(self.comm)
            self.works[b], self.done_ev[b] = w, done
        else:
            if narrow:
                src.copy_(table)
            glist = list(self.recv[b].view(torch.uint8).unbind(0)) if self.rank == self.root else None
            self.works[b] = dist.gather(src.view(torch.uint8), gather_list=glist, dst=self.root,
                                        group=self.group, async_op=True)
        return b

    def finish(self):
        if self.cuda:
            for ev in self.done_ev:
                if ev is not None:
                    ev.synchronize()
            self.torch.cuda.synchronize(self.dev)
        else:
            for b, w in enumerate(self.works):
                if w is not None:
                    w.wait()
                    self.works[b] = None

    def gathered(self, b: int):
        """Root: the (world * rows, stride) int32 table of buffer b in rank order; other ranks: None."""
        if self.rank != self.root:
            return None
        if self.wide is not None:                          # widened inside the step (submit)
            return self.wide[b].reshape(self.world * self.rows, self.stride)
        return self.recv[b].reshape(self.world * self.rows, self.stride).to(self.torch.int32)


class PipelinedGather:
    """Gathers the per-step match tables of all ranks to `root` on a communication stream while the
    next step computes: StereoBatchMatcher (compute) + TableGatherPipeline (wire format, buffers, ordering)."""

    def __init__(self, bm: StereoBatchMatcher, world: int, rank: int, root: int = 0, group=None, compact=None,
                 comm_on_stage_stream=None, wire16_from_kernel: bool = True, native=None):
        """comm_on_stage_stream: the wait for the collective and the root's widening go on the matcher's STAGE stream, behind the
        stages (and the narrowing copy, if any) of the step -- no third stream.  Measured on a ONE-rank RCCL group
        (tools/gather_step_probe.py, 512-pair steps): the step then costs what the plain step costs (0.344-0.349 against
        0.336-0.366 ms) whatever streams were created before, while a communication stream of its own -- from torch's
        high-priority pool, whose streams share a few hardware queues -- landed on 0.345-0.379 ms or, deterministically for
        some creation orders, on 0.45 ms.  A one-rank group only copies locally: with real links the collective's time would
        land on the stage stream's critical path (the next step's merge / finalize queue behind this step's gather), so the
        default (None) is the stage stream for world == 1 and a communication stream of its own for world > 1, where nothing has
        been measured; bench.py's probe times both.  The collective itself runs on the process group's internal stream either
        way; a HIGH-priority process-group stream made every step slower (0.44-0.59 ms): do not use it.
        wire16_from_kernel: with the int16 wire format the finalize kernel writes the int16 table itself
        (StereoBatchMatcher.enable_wire16) instead of a narrowing copy behind it.
        native (round 6; None = whenever possible): the whole step -- plan run, ncclGroup of sends / receives on the process
        group's own communicator, the root's widening -- is ONE call into the C ABI (plslam_match_plan_step_gather): no torch
        stream contexts, events or c10d work objects on the step's path (they were 0.06-0.2 ms of host time per 0.3 ms step)."""
        import time
        self._clock = time.perf_counter
        self.bm = bm
        if comm_on_stage_stream is None:
            comm_on_stage_stream = world == 1
        self.comm_on_stage_stream = bool(comm_on_stage_stream)
        self.pipe = TableGatherPipeline(bm.B, bm.stride, max(bm.n_orb, bm.n_lbd), world, rank, root,
                                        nbuf=len(bm.tables), device=bm.dev, group=group, compact=compact,
                                        comm_stream=bm.streams[min(1, len(bm.streams) - 1)] if comm_on_stage_stream else None)
        self.kernel_wire16 = bool(self.pipe.compact and wire16_from_kernel and self.pipe.cuda) and bm.enable_wire16(True)
        self.host_s, self.host_n = 0.0, 0          # host time spent inside step() (enqueueing only: nothing in it waits for the GPU)
        self.native, self._steps = False, {}
        if native is not False and self.pipe.cuda and (not self.pipe.compact or self.kernel_wire16):
            comm = self._native_comm(world, group)
            if comm is not None:
                self.native, self._comm, self._world, self._rank, self._root = True, comm, world, rank, root
            elif native:
                raise RuntimeError("PipelinedGather(native=True): the process group exposes no communicator (ProcessGroupNCCL._comm_ptr)")

    def _native_comm(self, world, group):
        """The ncclComm_t of this rank in `group` (0 for a one-rank job: the C entry point then needs none), with the C library
        told which loaded copy of librccl it belongs to; None if it cannot be had."""
        from . import capi
        try:
            import torch.distributed as dist
            if world == 1 and not (dist.is_available() and dist.is_initialized()):
                return 0                                    # a one-rank job without a process group: nothing to send
            pg = group if group is not None else dist.distributed_c10d._get_default_group()
            comm = int(pg._get_backend(self.bm.dev)._comm_ptr())
        except Exception:                                   # noqa: BLE001 -- any torch without the accessor: the torch path stays
            return 0 if world == 1 else None
        if not comm:
            return 0 if world == 1 else None
        try:
            path = next((l.split()[-1] for l in open("/proc/self/maps") if "librccl.so" in l), None)
        except OSError:
            path = None
        if path:
            capi.load().plslam_rccl_use(path.encode())     # (EINVAL once the library is loaded: the choice was made then)
        if world > 1 or comm:
            if not capi.load().plslam_rccl_available():     # (no librccl to be had: a set-up failure -- the torch path stays)
                return 0 if world == 1 and not comm else None
        return comm

    def _native_step(self, b: int, scan):
        key = (b, scan.cuda_stream)
        st = self._steps.get(key)
        if st is None:
            p, post = self.pipe, self.bm.streams[min(1, len(self.bm.streams) - 1)]
            send = self.bm.wire16[b] if p.compact else self.bm.tables[b]
            root = self._rank == self._root
            st = self.bm.plans[b].make_gather_step(
                self._comm, self._world, self._rank, self._root, 2 if p.compact else 4, send.data_ptr(), send.numel(),
                p.recv[b].data_ptr() if root else 0, p.wide[b].data_ptr() if (root and p.wide is not None) else 0,
                scan.cuda_stream, post.cuda_stream, p.comm.cuda_stream)
            self._steps[key] = st
        return st

    def step(self, k: int):
        """Compute step k into buffer k % nbuf and start gathering it."""
        t0 = self._clock()
        b = k % self.pipe.nbuf
        if self.native:
            self.bm.plans[b].step_gather(self._native_step(b, self.bm.scan_stream_of(k)))
            self.host_s += self._clock() - t0
            self.host_n += 1
            return b
        # as run_overlapped(): every scan on streams[0], the stages behind it -- which write table b -- on the
        # high-priority streams[1]; that stream therefore waits for the previous gather of buffer b, and the gather's event
        # (behind the narrowing copy, when there is one) goes behind the stages on it
        scan, post = self.bm.scan_stream_of(k), self.bm.streams[min(1, len(self.bm.streams) - 1)]
        self.pipe.before_overwrite(b, post)
        if scan is not post:                     # (column-split / fused plans write the table from the SCAN kernel: ADVICE r5)
            self.pipe.before_overwrite(b, scan)
        self.bm.plans[b].run_split(scan.cuda_stream, post.cuda_stream)
        r = self.pipe.submit(b, self.bm.tables[b], post, wire16=self.bm.wire16[b] if self.kernel_wire16 else None)
        self.host_s += self._clock() - t0
        self.host_n += 1
        return r

    def host_ms_per_step(self, reset: bool = True):
        v = 1e3 * self.host_s / max(self.host_n, 1)
        if reset:
            self.host_s, self.host_n = 0.0, 0
        return v

    def finish(self):
        if self.native:
            for p in self.bm.plans:
                p.gather_sync()
        self.pipe.finish()

    def gathered(self, b: int):
        return self.pipe.gathered(b)

    def close(self):
        """Detach from the matcher (the int16 mirrors go; the matcher can be wrapped again with another wire format)."""
        if self.native:
            self.finish()
            self.native, self._steps = False, {}
        if self.kernel_wire16:
            self.pipe.finish()
            self.bm.enable_wire16(False)
            self.kernel_wire16 = False


def spread_sample(B: int, n: int = 64):
    """>= min(n, B) pair indices of a B-pair shard for the root-side check: the shard's ends (0, 1, B - 2, B - 1: where the halo and
    the shard boundary are) and contiguous runs in between (a run costs one regeneration of its rank's inputs)."""
    if B <= n:
        return list(range(B))
    runs, per = 4, (n - 4) // 4
    idx = {0, 1, B - 2, B - 1}
    for q in range(runs):
        lo = min(max(2, (q * B) // runs + (B // runs - per) // 2), B - 2 - per)
        idx.update(range(lo, lo + per))
    i = 2
    while len(idx) < n:                     # (overlapping runs of a short shard)
        idx.add(i)
        i += 1
    return sorted(idx)


def verify_gathered_tables(full, world: int, B: int, n_orb: int, n_lbd: int, nnr_p: float, nnr_l: float, sample,
                           match_fn, seed=None, local_stream=None, first_pairs=None, workers=None):
    """Root-side check of a gathered (world * B, stride) table: pairs `sample` of EVERY rank against `match_fn(d1, d2,
    nnr) -> matches_12` (the caller's checker).  Rank r's inputs are regenerated from (seed, first_pairs[r]; default r * B:
    the weak-scaling shard rule of bench.py) -- except rank 0's, which may be passed in.  Contiguous runs of the sample are
    regenerated together (the synthetic chain restarts every 64 frames: a run costs at most 64 frames more than its length).
    Returns the list of mismatches (rank, pair, problem); empty = verified."""
    from . import synth
    sl = table_slices(n_orb, n_lbd)
    sample = sorted(set(int(i) for i in sample))
    runs = []
    for i_ in sample:
        if runs and i_ == runs[-1][1]:
            runs[-1][1] = i_ + 1
        else:
            runs.append([i_, i_ + 1])
    def check_run(job):
        r_, lo, hi = job
        first = r_ * B if first_pairs is None else first_pairs[r_]
        if r_ == 0 and local_stream is not None:
            st, off = local_stream, 0
        else:
            st = synth.stereo_stream(hi - lo, n_orb, n_lbd, seed=synth.SEED0 if seed is None else seed, first_pair=first + lo)
            off = lo
        out = []
        for i_ in range(lo, hi):
            tab = full[r_ * B + i_]
            for name, d1, d2 in pair_problems(st["orb_l"], st["orb_r"], st["lbd_l"], st["lbd_r"], i_ - off):
                em = match_fn(d1, d2, nnr_p if name.startswith("orb") else nnr_l)
                if not np.array_equal(np.asarray(tab[sl[name]]), em):
                    out.append((r_, i_, name))
        return out

    # (runs are independent; the checker's C code releases the GIL: at 8 ranks x 64 pairs the root would otherwise spend most
    # of a minute per table here)
    jobs = [(r_, lo, hi) for r_ in range(world) for lo, hi in runs]
    if workers is None:
        import os
        try:
            workers = max(1, min(16, len(os.sched_getaffinity(0))))
        except AttributeError:
            workers = 4
    if workers > 1 and len(jobs) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            res = list(ex.map(check_run, jobs))
    else:
        res = [check_run(j) for j in jobs]
    return sorted(x for out in res for x in out)


def gather_tables(local, world: int, rank: int, root: int = 0, group=None, force: bool = False):
    """Gather fixed-stride per-pair match tables to `root` (torch.distributed: backend 'nccl' is
    RCCL over xGMI on MI355X, 'gloo' on CPU).  Returns the (world*B, stride) tensor on root."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force:
        return local
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == root else None
    dist.gather(local, gather_list=bufs, dst=root, group=group)
    return torch.cat(bufs, dim=0) if rank == root else None
