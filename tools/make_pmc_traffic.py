#!/usr/bin/env python3
"""Turns the rocprofv3 counter summaries of one bench command (tools/pmc_passes.sh -> gpurun_out/<tag>_pmc_{a,fetch,write}.txt)
into the entry of profiles/pmc_traffic.json that bench.py reports as roofline.traffic / valu_executed -- keyed by workload and
kernel, and stamped with the hash of the kernel sources the passes were taken on (bench.py ignores an entry whose hash is not
the tree's).   usage: make_pmc_traffic.py <tag> <n_orb> <n_lbd> <pairs> <kernel substring, e.g. k_scan_sym_mfma_h> [fetch correction]

FETCH_SIZE correction: MI355X_MICROARCH.md measures that this rocprofv3 reports HALF the bytes of a wide coalesced streaming read
(16 B per lane) and calls other access widths uncalibrated ("calibrate on a known byte count in your own access pattern").  The
calibration for 4-byte-per-lane loads is in round 3's passes: k_merge_fix16 reads the column-partial table exactly once with
dword loads -- in set r3_z 1.24 GB at C2 / 4096 pairs (4096 x (2 x 24 x 1536 + 2 x 4 x 256) words; the scan's WRITE_SIZE agrees)
-- and FETCH_SIZE reported 599 317 KiB = 0.614 GB: the factor is 2.0 for dword loads as well.  (An earlier reading of the same
numbers took the table for 614 MB and keyed K1h's entry with 1.0: wrong, corrected here.)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402


def counters(path, kernel):
    out = {}
    for line in open(path):
        if kernel not in line:
            continue
        m = re.search(r"\s([A-Z][A-Z0-9_]+)\s+(\d+)\s+([0-9.]+)\s*$", line)
        if m:
            out[m.group(1)] = float(m.group(3))
    return out


def main():
    tag, n_orb, n_lbd, pairs, kernel = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    corr = float(sys.argv[6]) if len(sys.argv) > 6 else 2.0
    g = os.path.join(ROOT, "gpurun_out")
    a = counters(os.path.join(g, f"{tag}_pmc_a.txt"), kernel)
    f = counters(os.path.join(g, f"{tag}_pmc_fetch.txt"), kernel)
    w = counters(os.path.join(g, f"{tag}_pmc_write.txt"), kernel)
    fetch_kib, write_kib = f["FETCH_SIZE"], w["WRITE_SIZE"]
    # the same command's kernel trace (no counters: undisturbed durations): the scan kernel's average launch
    trace_us, trace_calls, trace_file = None, None, f"{tag}_kernel_trace_stats_serial.txt"
    try:
        for line in open(os.path.join(g, trace_file)):
            if kernel in line:
                m = re.search(r"\s(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line)
                if m:
                    trace_calls, trace_us = int(m.group(1)), float(m.group(3))
                    break
    except OSError:
        pass
    entry = {
        "kernel": kernel, "kernel_source_hash": kernel_source_hash(),
        "trace_avg_us": trace_us, "trace_calls": trace_calls, "trace_file": "profiles/" + trace_file,
        "fetch_size_kib_per_dispatch": fetch_kib, "write_size_kib_per_dispatch": write_kib, "fetch_correction": corr,
        "traffic_bytes_per_launch": int(fetch_kib * 1024 * corr + write_kib * 1024),
        "sq_insts_valu_per_launch": int(a["SQ_INSTS_VALU"]), "sq_insts_mfma_per_launch": int(a.get("SQ_INSTS_MFMA", 0)),
        "sq_valu_mfma_busy_cycles_per_launch": int(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)),
        "sq_active_inst_valu_per_launch": int(a.get("SQ_ACTIVE_INST_VALU", 0)),
        "sq_wave_cycles_per_launch": int(a.get("SQ_WAVE_CYCLES", 0)),
        "measured_int_valu_ceiling_lane_ops_per_s": 38500000000000.0,
        "measured_int_valu_ceiling_source": "profiles/r1_valu_microbench.txt, profiles/r2_valu_microbench2.txt: pk_min / perm / "
                                            "and_or class ops at 4.1-4.5 cycles per wave64 instruction per SIMD",
        "note": ("FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (the factor holds for dword-per-lane loads too: "
                 "k_merge_fix16 reads a table of known size once and reports half of it)" if corr == 2.0 else
                 f"FETCH_SIZE x {corr}") + "; WRITE_SIZE includes the kernel's register spills (scratch); "
                "SQ_INSTS_VALU counts the MFMAs too (bench.py subtracts SQ_INSTS_MFMA)",
        "source": f"profiles/{tag}_pmc_a.txt, profiles/{tag}_pmc_fetch.txt, profiles/{tag}_pmc_write.txt (rocprofv3 --pmc, "
                  "separate passes, python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-secondary)"}
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        doc = json.load(open(path))
    except OSError:
        doc = {"entries": {}}
    doc.setdefault("entries", {})[f"{n_orb}+{n_lbd}:pairs{pairs}:{kernel}"] = entry
    doc["key_format"] = "<n_orb>+<n_lbd>:pairs<pairs per GPU per step>:<scan kernel>; entries carry kernel_source_hash"
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
