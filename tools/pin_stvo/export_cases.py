#!/usr/bin/env python3
"""Exports the committed golden vectors of the matcher path (tests/golden/match_golden.npz, grid_golden.npz,
stereo_gates_golden.npz) and of the [RECALL] helpers of the LBA rows (se3_helpers_golden.npz) as flat binary arrays + a text manifest that tools/pin_stvo/pin_stvo.cpp reads -- the C++ side then
needs nothing but OpenCV and a stvo-pl checkout.   usage: export_cases.py <out_dir>

Array file: "PLSA" | dtype char (u = uint8, i = int32, f = float32, d = float64) | int32 ndim | int64 dims[ndim] | data.
Manifest: one case per line, whitespace-separated `key=value`; array values are file names."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
CODES = {np.dtype(np.uint8): b"u", np.dtype(np.int32): b"i", np.dtype(np.float32): b"f", np.dtype(np.float64): b"d"}


def write_array(path, a):
    a = np.ascontiguousarray(a)
    if a.dtype not in CODES:
        a = a.astype(np.int32) if np.issubdtype(a.dtype, np.integer) else a.astype(np.float64)
    with open(path, "wb") as f:
        f.write(b"PLSA" + CODES[a.dtype] + struct.pack("<i", a.ndim) + struct.pack(f"<{a.ndim}q", *a.shape))
        f.write(a.tobytes())


def read_array(path):
    inv = {v: k for k, v in CODES.items()}
    with open(path, "rb") as f:
        assert f.read(4) == b"PLSA"
        dt = inv[f.read(1)]
        nd, = struct.unpack("<i", f.read(4))
        shape = struct.unpack(f"<{nd}q", f.read(8 * nd))
        return np.frombuffer(f.read(), dtype=dt).reshape(shape)


def main(out):
    os.makedirs(out, exist_ok=True)
    lines = []

    def put(name, a):
        write_array(os.path.join(out, name + ".bin"), a)
        return name + ".bin"

    g = np.load(os.path.join(GOLD, "match_golden.npz"))
    for n in sorted({k.split("/")[0] for k in g.files}):
        q, t = put(f"match_{n}_q", g[f"{n}/q"]), put(f"match_{n}_t", g[f"{n}/t"])
        for nnr in ("0.6", "0.75", "0.9"):
            for mut in (0, 1):
                key = f"{n}/m12_nnr{nnr}_mut{mut}"
                if key in g.files:
                    lines.append(f"kind=match name={n} q={q} t={t} nnr={nnr} mutual={mut} "
                                 f"expect={put(f'match_{n}_m12_{nnr}_{mut}', g[key])}")
    g = np.load(os.path.join(GOLD, "grid_golden.npz"))
    for n in [str(x) for x in g["names"]]:
        cols, rows, w0, w1, w2, w3, is_lines = [int(x) for x in g[f"{n}_meta"]]
        base = (f"name={n} cols={cols} rows={rows} w0={w0} w1={w1} w2={w2} w3={w3} centres={put(f'grid_{n}_cen', g[f'{n}_centres'])} "
                f"d1={put(f'grid_{n}_d1', g[f'{n}_d1'])} d2={put(f'grid_{n}_d2', g[f'{n}_d2'])} "
                f"cell_start={put(f'grid_{n}_cs', g[f'{n}_cell_start'])} cell_items={put(f'grid_{n}_it', g[f'{n}_cell_items'])}")
        if is_lines:
            base += f" dir2={put(f'grid_{n}_dir2', g[f'{n}_dir2'])}"
        for mut in (0, 1):
            for r in (75, 90):
                lines.append(f"kind={'grid_lines' if is_lines else 'grid_points'} {base} nnr={r / 100} mutual={mut} "
                             f"expect={put(f'grid_{n}_m{mut}_r{r}', g[f'{n}_m{mut}_r{r}'])}")
    g = np.load(os.path.join(GOLD, "stereo_gates_golden.npz"))
    for c in range(3):
        for t, th in enumerate(g["point_thresholds"]):
            lines.append(f"kind=gate_points name=p{c}t{t} m12={put(f'gate_p{c}_m12', g[f'p{c}_m12'])} "
                         f"kp_l={put(f'gate_p{c}_kpl', g[f'p{c}_kp_l'])} kp_r={put(f'gate_p{c}_kpr', g[f'p{c}_kp_r'])} "
                         f"max_dist_epip={th[0]} min_disp={th[1]} expect={put(f'gate_p{c}_t{t}_s', g[f'p{c}_t{t}_stereo'])} "
                         f"expect_disp={put(f'gate_p{c}_t{t}_d', g[f'p{c}_t{t}_disp'])}")
        for t, th in enumerate(g["line_thresholds"]):
            lines.append(f"kind=gate_lines name=l{c}t{t} m12={put(f'gate_l{c}_m12', g[f'l{c}_m12'])} "
                         f"seg_l={put(f'gate_l{c}_segl', g[f'l{c}_seg_l'])} seg_r={put(f'gate_l{c}_segr', g[f'l{c}_seg_r'])} "
                         f"min_disp={th[0]} line_horiz_th={th[1]} stereo_overlap_th={th[2]} ls_min_disp_ratio={th[3]} "
                         f"expect={put(f'gate_l{c}_t{t}_s', g[f'l{c}_t{t}_stereo'])} expect_disp={put(f'gate_l{c}_t{t}_d', g[f'l{c}_t{t}_disp'])}")
    g = np.load(os.path.join(GOLD, "se3_helpers_golden.npz"))
    lines.append("kind=se3 name=helpers " + " ".join(f"{k}={put('se3_' + k, g[k])}" for k in sorted(g.files)))
    with open(os.path.join(out, "manifest.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"{len(lines)} cases -> {out}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "pin_cases")
