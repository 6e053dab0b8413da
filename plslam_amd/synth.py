"""Deterministic synthetic workloads: descriptor-level replays of stereo streams
and a C3-shaped local map for the LBA rows.  Pure numpy; no algorithm of the hot path lives here.
"""
from __future__ import annotations

import numpy as np

SEED0 = 20170530
EUROC = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, b=0.110, width=752, height=480)
KITTI00 = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, b=0.537165719, width=1241, height=376)


def random_desc(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def noisy_copy(rng, src, keep_frac=0.70, flip_p=0.08):
    """Rows of `src`: a random `keep_frac` get per-bit flips with prob. `flip_p` (true matches),
    the rest are replaced by fresh random rows; the result is row-shuffled.  Returns (dst, perm)
    where dst[k] descends from src[perm[k]] (or is fresh when fresh[k])."""
    n = src.shape[0]
    keep = rng.random(n) < keep_frac
    flips = np.packbits(rng.random((n, 256), dtype=np.float32) < flip_p, axis=1)
    dst = np.where(keep[:, None], src ^ flips, random_desc(rng, n))
    perm = rng.permutation(n)
    return np.ascontiguousarray(dst[perm]), perm, ~keep[perm]


def tie_stress_desc(rng, n, entropy_bits=5):
    """Rows drawn from only 2**entropy_bits distinct patterns => many exact ties/duplicates."""
    pats = random_desc(rng, 1 << entropy_bits)
    # make patterns near each other so that distances collide as well
    base = random_desc(rng, 1)[0]
    for k in range(pats.shape[0]):
        p = base.copy()
        bits = rng.integers(0, 256, size=3)
        for b in bits:
            p[b >> 3] ^= np.uint8(1 << (b & 7))
        pats[k] = p
    return np.ascontiguousarray(pats[rng.integers(0, pats.shape[0], size=n)])


def landmark_desc_lists(rng, n_lm, max_obs=8, flip=0.08, empty_frac=0.0, ties=False):
    """Observation descriptor lists of n_lm landmarks (MapPoint::desc_list), concatenated.
    Each list = noisy copies of one base row (flip = per-bit flip probability); list lengths are
    uniform in [1, max_obs]; a fraction may be empty.  ties=True draws the noise from few patterns so
    that equal medians between rows are common.  Returns (desc[total,32] u8, offsets[n_lm+1] i32)."""
    lens = rng.integers(1, max_obs + 1, size=n_lm)
    if empty_frac > 0:
        lens[rng.random(n_lm) < empty_frac] = 0
    off = np.zeros(n_lm + 1, np.int32)
    np.cumsum(lens, out=off[1:])
    total = int(off[-1])
    base = np.repeat(random_desc(rng, n_lm), lens, axis=0) if total else np.zeros((0, 32), np.uint8)
    if ties:
        pats = np.packbits(rng.random((4, 256)) < flip, axis=1)
        noise = pats[rng.integers(0, 4, size=total)]
    else:
        noise = np.packbits(rng.random((total, 256)) < flip, axis=1)
    return np.ascontiguousarray(base ^ noise), off


def lbd_float(rng, n, levels=0):
    """n x 72 float32 rows shaped like LBD output (binary_descriptor_custom.cpp computeLBD: per band 4
    means + 4 std-devs, unit-normalised, clipped): non-negative, <= 0.4.  levels > 0 quantises to that
    many distinct values so that f1[i] == f2[i] ties (strict '>' in binaryConversion) are frequent."""
    f = np.abs(rng.standard_normal((n, 72))).astype(np.float32)
    f /= np.maximum(np.linalg.norm(f, axis=1, keepdims=True), np.float32(1e-12))
    f = np.minimum(f, np.float32(0.4))
    if levels > 0:
        f = np.round(f * np.float32(levels / 0.4)).astype(np.float32) * np.float32(0.4 / levels)
    return np.ascontiguousarray(f, np.float32)


def stereo_stream(n_pairs, n_orb=1500, n_lbd=200, seed=SEED0, first_pair=0, tie_stress=False):
    """A stream of stereo pairs.  Returns dict of uint8 arrays:
         orb_l, orb_r : (n_pairs+1, n_orb, 32)   index 0 is the HALO = left image of the pair before
         lbd_l, lbd_r : (n_pairs+1, n_lbd, 32)   `first_pair` (needed for prev<->curr of pair 0)
    Frame f = first_pair - 1 + index.  Every frame's rows depend only on (seed, f) and on frame
    f-1's left rows, and the chain restarts every 64 frames, so any shard can be generated
    independently of the others (contiguous shards produce bit-identical data)."""
    out = {k: np.empty((n_pairs + 1, n, 32), np.uint8)
           for k, n in (("orb_l", n_orb), ("orb_r", n_orb), ("lbd_l", n_lbd), ("lbd_r", n_lbd))}
    # which left row a right row descends from (and whether it is a fresh random row instead): the ground truth
    # stereo_geometry() places the key points / segments by.  No effect on the descriptors or on the random sequence.
    for kind, n in (("orb", n_orb), ("lbd", n_lbd)):
        out[kind + "_rperm"] = np.zeros((n_pairs + 1, n), np.int32)
        out[kind + "_rfresh"] = np.ones((n_pairs + 1, n), bool)
    f0 = first_pair - 1
    start = (f0 // 64) * 64 if f0 >= 0 else f0  # chain restart boundary at or before f0
    prev = {}
    for f in range(start, f0 + n_pairs + 1):
        rng = np.random.Generator(np.random.PCG64(seed + 7919 * (f + 1)))
        cur = {}
        for kind, n in (("orb", n_orb), ("lbd", n_lbd)):
            rperm, rfresh = np.arange(n, dtype=np.int32), np.ones(n, bool)
            if tie_stress:
                left = tie_stress_desc(rng, n)
                right = tie_stress_desc(rng, n)
            else:
                if f == start or (f % 64) == 0 or kind not in prev:
                    left = random_desc(rng, n)
                else:
                    left, _, _ = noisy_copy(rng, prev[kind])
                right, rperm, rfresh = noisy_copy(rng, left)
            cur[kind] = left
            i = f - f0
            if i >= 0:
                out[kind + "_l"][i] = left
                out[kind + "_r"][i] = right
                out[kind + "_rperm"][i] = rperm
                out[kind + "_rfresh"][i] = rfresh
        prev = cur
    return out


def stereo_geometry(stream, seed=SEED0 + 1, width=752, height=480, first_pair=0):
    """Key points and line segments for the frames of a stereo_stream(): float32 arrays kp_l, kp_r (frames, n_orb, 2)
    and seg_l, seg_r (frames, n_lbd, 4) -- what StereoFrame's detectors would hand to matchStereoPoints / Lines.
    A right feature that descends from a left one (stream["*_rperm"], not "*_rfresh") sits at the left position minus
    a disparity, with sub-pixel noise across the epipolar line; fresh ones lie anywhere.  A share of the lines is
    horizontal, degenerate or only partly overlapping, so that every branch of the gates is taken.  Frame f's
    geometry depends only on (seed, first_pair + index)."""
    frames, n_orb = stream["orb_l"].shape[:2]
    n_lbd = stream["lbd_l"].shape[1]
    out = {"kp_l": np.empty((frames, n_orb, 2), np.float32), "kp_r": np.empty((frames, n_orb, 2), np.float32),
           "seg_l": np.empty((frames, n_lbd, 4), np.float32), "seg_r": np.empty((frames, n_lbd, 4), np.float32)}
    for i in range(frames):
        r = np.random.Generator(np.random.PCG64(seed + 104729 * (first_pair + i)))
        # FAST corners sit on integer pixels (level 0); the shipped configuration keeps only pairs on the SAME row
        # (max_dist_epip: 0.0), so the row noise is quantised: about 60 % of the true pairs stay on their row
        kp_l = np.floor(np.stack([r.uniform(0, width, n_orb), r.uniform(0, height, n_orb)], 1))
        src = stream["orb_rperm"][i]
        kp_r = kp_l[src] - np.stack([np.floor(r.uniform(-3, 60, n_orb)), np.rint(r.normal(0, 0.6, n_orb))], 1)
        fresh = stream["orb_rfresh"][i]
        kp_r[fresh] = np.stack([r.uniform(0, width, n_orb), r.uniform(0, height, n_orb)], 1)[fresh]
        out["kp_l"][i], out["kp_r"][i] = kp_l, kp_r
        a = np.stack([r.uniform(0, width, n_lbd), r.uniform(0, height, n_lbd)], 1)
        ang, ln = r.uniform(0, np.pi, n_lbd), r.uniform(5, 150, n_lbd)
        ln[r.random(n_lbd) < 0.03] = 0.0                                 # zero-length segments
        ang[r.random(n_lbd) < 0.08] = 0.0                                # horizontal: dy = 0 (division by zero)
        seg_l = np.concatenate([a, a + np.stack([np.cos(ang), np.sin(ang)], 1) * ln[:, None]], 1)
        src = stream["lbd_rperm"][i]
        d0, d1 = r.uniform(-2, 50, n_lbd), r.uniform(0.5, 1.5, n_lbd)
        seg_r = seg_l[src].copy()
        seg_r[:, 0] -= d0
        seg_r[:, 2] -= d0 * d1
        seg_r += r.normal(0, 0.6, seg_r.shape)
        cut = r.random(n_lbd) < 0.25                                     # partial vertical overlap
        seg_r[cut, 2:] = seg_r[cut, :2] + (seg_r[cut, 2:] - seg_r[cut, :2]) * r.uniform(0.1, 0.9, (int(cut.sum()), 1))
        fresh = stream["lbd_rfresh"][i]
        rnd = np.concatenate([np.stack([r.uniform(0, width, n_lbd), r.uniform(0, height, n_lbd)], 1),
                              np.stack([r.uniform(0, width, n_lbd), r.uniform(0, height, n_lbd)], 1)], 1)
        seg_r[fresh] = rnd[fresh]
        out["seg_l"][i], out["seg_r"][i] = seg_l, seg_r
    return out


# the stereo-gate thresholds of the reference's shipped configuration (config/config/config_kitti.yaml:25-36)
KITTI_GATES = dict(max_dist_epip=0.0, min_disp=1.0, line_horiz_th=0.1, stereo_overlap_th=0.75, ls_min_disp_ratio=0.7)


def se3_exp(x):
    t, w = np.asarray(x[:3], float), np.asarray(x[3:], float)
    th = np.linalg.norm(w)
    T = np.eye(4)
    if th < 1e-6:
        T[:3, 3] = t
        return T
    s = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
    R = np.eye(3) + s * np.sin(th) + s @ s * (1 - np.cos(th))
    V = np.eye(3) + s * (1 - np.cos(th)) / th + s @ s * (th - np.sin(th)) / th
    T[:3, :3] = R
    T[:3, 3] = V @ t
    return T


def local_map(n_kf=10, n_pt=10000, n_ls=2000, obs_per_lm=5, cam=EUROC, seed=7, noise_px=1.0):
    """C3-shaped local map: forward-moving KF trajectory, landmarks inside the first KF's frustum at
    2..30 m, observations = exact projection + N(0, noise_px).  Returns a dict of arrays laid out as
    the LBA row ABI expects."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fx, fy, cx, cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    W, H = cam["width"], cam["height"]
    T_kf_w = np.empty((n_kf, 4, 4))
    for k in range(n_kf):
        x = np.concatenate([[0.02 * rng.standard_normal(), 0.02 * rng.standard_normal(), 0.25 * k],
                            0.01 * rng.standard_normal(3)])
        T_kf_w[k] = se3_exp(x)

    def frustum(n):
        z = rng.uniform(4.0, 30.0, n)
        u = rng.uniform(0.15 * W, 0.85 * W, n)
        v = rng.uniform(0.15 * H, 0.85 * H, n)
        return np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], axis=1)

    def project(T, X):
        Ti = np.linalg.inv(T)
        Xc = X @ Ti[:3, :3].T + Ti[:3, 3]
        return np.stack([cx + fx * Xc[:, 0] / Xc[:, 2], cy + fy * Xc[:, 1] / Xc[:, 2]], axis=1)

    Xw = frustum(n_pt)
    kf_pt = np.stack([rng.permutation(n_kf)[:obs_per_lm] for _ in range(n_pt)]).astype(np.int32) \
        if n_pt else np.zeros((0, obs_per_lm), np.int32)
    lm_pt = np.repeat(np.arange(n_pt, dtype=np.int32), obs_per_lm)
    kf_pt = kf_pt.reshape(-1)
    uv = np.empty((lm_pt.shape[0], 2))
    for k in range(n_kf):
        sel = kf_pt == k
        uv[sel] = project(T_kf_w[k], Xw[lm_pt[sel]])
    uv += noise_px * rng.standard_normal(uv.shape)

    P = frustum(n_ls)
    Q = P + rng.uniform(-1.0, 1.0, (n_ls, 3)) * np.array([1.0, 1.0, 0.3])
    Lw = np.concatenate([P, Q], axis=1)
    kf_ls = np.stack([rng.permutation(n_kf)[:obs_per_lm] for _ in range(n_ls)]).astype(np.int32).reshape(-1) \
        if n_ls else np.zeros(0, np.int32)
    lm_ls = np.repeat(np.arange(n_ls, dtype=np.int32), obs_per_lm)
    l_obs = np.empty((lm_ls.shape[0], 3))
    for k in range(n_kf):
        sel = kf_ls == k
        p = project(T_kf_w[k], P[lm_ls[sel]]) + noise_px * rng.standard_normal((sel.sum(), 2))
        q = project(T_kf_w[k], Q[lm_ls[sel]]) + noise_px * rng.standard_normal((sel.sum(), 2))
        l = np.cross(np.concatenate([p, np.ones((p.shape[0], 1))], 1),
                     np.concatenate([q, np.ones((q.shape[0], 1))], 1))
        l /= np.sqrt(l[:, 0:1] ** 2 + l[:, 1:2] ** 2)  # normalised 2D line eq. (include/mapFeatures.h:93)
        l_obs[sel] = l
    return dict(T_kf_w=T_kf_w.reshape(n_kf, 16), Xw=Xw, obs_uv=uv, pt_lm=lm_pt, pt_kf=kf_pt,
                Lw=Lw, l_obs=l_obs, ls_lm=lm_ls, ls_kf=kf_ls)


def grid_frame_pair(rng, n1, n2, width=752, height=480, keep_frac=0.7, flip_p=0.08, shift_px=12.0, ties=False,
                    lines=False):
    """Two frames for the windowed matcher (StVO::matchGrid; src/mapHandler.cpp:252-271 / :381-418): n1 features of
    the previous keyframe projected into the current image (some fall outside it) and n2 features of the current
    one, `keep_frac` of which descend from previous ones (descriptor bit flips with prob. flip_p, position moved by
    N(0, shift_px)); the rest are fresh.  ties=True draws descriptors from few patterns.
    points: px1 (n1,2), px2 (n2,2) pixel positions; lines: seg1 (n1,4), seg2 (n2,4) pixel segments (x1,y1,x2,y2)."""
    mk = (lambda n: tie_stress_desc(rng, n)) if ties else (lambda n: random_desc(rng, n))
    d1 = mk(n1)
    p1 = np.stack([rng.uniform(-0.05 * width, 1.05 * width, n1), rng.uniform(-0.05 * height, 1.05 * height, n1)], 1)
    src = rng.integers(0, max(n1, 1), size=n2)
    keep = (rng.random(n2) < keep_frac) & (n1 > 0)
    flips = np.packbits(rng.random((n2, 256), dtype=np.float32) < flip_p, axis=1)
    d2 = np.where(keep[:, None], (d1[src] if n1 else mk(n2)) ^ flips, mk(n2))
    fresh = np.stack([rng.uniform(0, width, n2), rng.uniform(0, height, n2)], 1)
    p2 = np.where(keep[:, None], (p1[src] if n1 else fresh) + rng.normal(0, shift_px, (n2, 2)), fresh)
    out = dict(d1=np.ascontiguousarray(d1), d2=np.ascontiguousarray(d2), width=width, height=height)
    if not lines:
        out.update(px1=p1, px2=p2)
        return out
    def seg(p, n):
        ang = rng.uniform(0, np.pi, n)
        ln = rng.uniform(0, 0.25 * width, n) * (rng.random(n) > 0.05)      # a few zero-length segments
        v = np.stack([np.cos(ang), np.sin(ang)], 1) * ln[:, None]
        return np.concatenate([p, p + v], 1)
    s1 = seg(p1, n1)
    s2 = seg(p2, n2)
    if n1:
        s2 = np.where(keep[:, None], s1[src] + np.tile(p2 - p1[src], 2) + rng.normal(0, 1.0, (n2, 4)), s2)
    out.update(seg1=s1, seg2=s2)
    return out


def gradient_images(rng, width=752, height=480, smooth=3):
    """Sobel-like int16 gradient images of a random smoothed image (what EDLineDetector / the octave pyramid hand to
    computeLBD as dxImg / dyImg)."""
    img = rng.integers(0, 256, size=(height, width)).astype(np.float64)
    for _ in range(smooth):
        img = (img + np.roll(img, 1, 0) + np.roll(img, -1, 0) + np.roll(img, 1, 1) + np.roll(img, -1, 1)) / 5.0
    p = np.pad(img, 1, mode="edge")
    dx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    return np.rint(dx).astype(np.int16), np.rint(dy).astype(np.int16)


def lbd_lines(rng, n, width=752, height=480, min_len=10.0, max_len=250.0, dtype=None):
    """Random segments as the line detector reports them (OctaveSingleLine fields read by computeLBD); some reach past
    the image border so that the clamping is exercised."""
    sx, sy = rng.uniform(-5, width + 5, n), rng.uniform(-5, height + 5, n)
    ang, ln = rng.uniform(-np.pi, np.pi, n), rng.uniform(min_len, max_len, n)
    ex, ey = sx + ln * np.cos(ang), sy + ln * np.sin(ang)
    out = np.zeros(n, dtype=dtype)
    out["num_pixels"] = np.rint(ln).astype(np.int32)
    out["sx"], out["sy"], out["ex"], out["ey"] = sx, sy, ex, ey
    out["direction"] = np.arctan2(ey - sy, ex - sx)
    return out
