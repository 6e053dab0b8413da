"""CPU check of the arithmetic behind K1e (plslam_amd/csrc/hamming_mfma.hip): the Hamming distance as a
contraction over fp4 (e2m1) codes of +-1 with fp32 accumulation started at 2^23 + 16384 + tag.  The test
emulates the operand expansion, the block scale and the accumulation in float32 -- in several summation
orders, since the instruction's internal order is not specified -- and checks that the low 16 bits of the
resulting float are exactly (d << 7 | tag).  (The kernel itself is tested bit-exact on the GPU.)"""
import numpy as np
import pytest

from oracle import oracle as O
from plslam_amd import synth

E2M1 = {0x2: 1.0, 0xA: -1.0}          # the only two codes the kernel produces


def expand_byte_fp4(byte):
    """hamming_mfma.hip::expand_byte_fp4: bit k -> nibble k = 0x2 | (bit << 3)."""
    x = (byte | (byte << 12)) & 0x000F000F
    x = (x | (x << 6)) & 0x03030303
    x = (x | (x << 3)) & 0x11111111
    return ((x << 3) | 0x22222222) & 0xFFFFFFFF


def values_of_row(row_u8, a_side):
    out = []
    for byte in row_u8:
        w = expand_byte_fp4(int(byte))
        if a_side:
            w ^= 0x88888888           # -s(a)
        out += [E2M1[(w >> (4 * k)) & 0xF] for k in range(8)]
    return np.array(out, np.float32)


def test_expansion_table():
    for byte in range(256):
        w = expand_byte_fp4(byte)
        for k in range(8):
            assert (w >> (4 * k)) & 0xF == (0xA if (byte >> k) & 1 else 0x2)


@pytest.mark.parametrize("order", ["forward", "reverse", "pairwise", "random"])
def test_accumulation_is_exact_in_fp32(order):
    r = np.random.Generator(np.random.PCG64(11))
    a = synth.random_desc(r, 10)
    b = np.concatenate([synth.random_desc(r, 8), a[:2], ~a[2:4]])       # incl. distance 0 and 256
    va = [values_of_row(x, True) * np.float32(64.0) for x in a]          # block scale 2^6 on the a side
    vb = [values_of_row(x, False) for x in b]
    for tag in (0, 27, 90, 127):
        for i in range(a.shape[0]):
            for j in range(b.shape[0]):
                prod = (va[i] * vb[j]).astype(np.float32)                # +-64 exactly
                acc = np.float32(8388608.0 + 16384.0 + tag)
                if order == "pairwise":                                  # tree sum of each K = 64 block, chained
                    for blk in range(4):
                        p = prod[64 * blk:64 * blk + 64].copy()
                        while p.size > 1:
                            p = (p[0::2] + p[1::2]).astype(np.float32)
                        acc = np.float32(acc + p[0])
                else:
                    idx = {"forward": np.arange(256), "reverse": np.arange(255, -1, -1),
                           "random": r.permutation(256)}[order]
                    for k in idx:
                        acc = np.float32(acc + prod[k])
                        assert abs(float(acc)) < 2 ** 24
                d = O.hamming256(a[i], b[j])
                bits = int(np.array([acc], np.float32).view(np.uint32)[0])
                assert bits == 0x4B000000 + 128 * d + tag                # 2^23 + 128 d + tag
                assert ((bits & 0xFFFF) >> 7) == d and (bits & 0x7F) == tag


def test_key_orders_like_distance_then_index():
    """(d << 7 | tile + LOC) within one row state: LOC is constant, so keys order like (d, tile)."""
    loc = 27
    keys = [((d << 7) | (t + loc), d, t) for d in (0, 1, 128, 256) for t in (0, 1, 63)]
    assert [k[1:] for k in sorted(keys)] == sorted(k[1:] for k in keys)
    assert max(k[0] for k in keys) <= 0x807F


@pytest.mark.parametrize("fname", ["hamming_mfma.hip", "hamming_mfma_g.hip", "hamming_mfma_h.hip"])
def test_no_inline_asm_reads_mfma_results(fname):
    """Guard for the K1e determinism bug: the wait states between a v_mfma and a VALU access to
    its destination registers are inserted by the compiler, which does not look inside asm statements.  pack_acc -- the
    one consumer of accumulator registers -- must therefore stay a builtin, and no asm statement may take an accumulator
    element as an operand."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plslam_amd", "csrc",
                            fname)).read()
    src = "\n".join(l.split("//")[0] for l in src.split("\n"))          # code only
    assert "__builtin_amdgcn_perm(" in src
    assert 'asm("v_perm_b32' not in src and "asm volatile(\"v_perm_b32" not in src
    for m in re.finditer(r"asm(?:\s+volatile)?\s*\(([^;]*);", src):
        body = m.group(1)
        if re.match(r'\s*""', body):                      # an empty template issues no instruction
            continue
        assert not re.search(r"\bacc[01]\b|\bm[01]\b\s*[\[\)]|\bA[01]\b|\bB[01]\b", body), body[:120]
    # the accumulators reach the bookkeeping through pack_acc only (K1e: acc0[R] / acc1[R] inside the call; K1f: the
    # elements are copied to scalars f0 / f1 first -- a toolchain defect with bit casts of vector elements -- and those
    # scalars go nowhere but into pack_acc)
    if fname == "hamming_mfma.hip":
        uses = [l for l in src.split("\n") if re.search(r"\bacc[01]\[", l)]
        assert uses and all("pack_acc(" in l for l in uses), uses
    else:
        uses = [l for l in src.split("\n") if re.search(r"\bm[01]\[r\]", l)]
        assert uses and all(re.search(r"const float f0 = m0\[r\], f1 = m1\[r\];", l) for l in uses), uses
        assert len(re.findall(r"\bf[01]\b", src)) == len(re.findall(r"pack_acc\(f0, f1", src)) * 2 + 2 * len(uses)


@pytest.mark.parametrize("fname", ["hamming_mfma.hip", "hamming_mfma_g.hip", "hamming_mfma_h.hip", "hamming_mfma_d.hip", "hamming_mfma_i.hip"])
def test_final_isa_has_no_mfma_destination_hazard(tmp_path, fname):
    """Compiles hamming_mfma.hip to gfx950 assembly (as build.py does, -S instead of -shared) and runs
    tools/check_mfma_hazards.py over the FINAL listing, inline-asm bodies included: no non-MFMA instruction may touch a
    destination register of a v_mfma within 12 wait states on any path (fall-through and taken branches).  The listing
    of the kernel as it was before the K1e determinism fix has 94 such places; the shipped one must have none."""
    import importlib.util
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("hipcc not available")
    out = str(tmp_path / (fname + ".s"))
    # (-DPLSLAM_BUILD_LEGACY_SCANS=1: K1h's scan kernel is compiled only on request -- plslam_amd/build.py -- and is checked here
    # all the same, like K1e and K1g, whose files the product build leaves out)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-inline-asm",
                    "-DPLSLAM_BUILD_LEGACY_SCANS=1", "-I" + os.path.join(root, "include"),
                    "-S", "--cuda-device-only", "-o", out, os.path.join(root, "plslam_amd", "csrc", fname)],
                   check=True, capture_output=True)
    spec = importlib.util.spec_from_file_location("check_mfma_hazards", os.path.join(root, "tools", "check_mfma_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = open(out).read()
    # all instantiations are there (K1h / K1g: one set of MFMAs per unrolled step)
    # (K1i issues the UNSCALED instruction -- also through one inline-asm statement per tile, which is why the checker reads
    # the final listing, asm bodies included)
    if fname == "hamming_mfma_i.hip":
        assert text.count("v_mfma_f32_32x32x64_f8f6f4") >= 128 and "v_mfma_scale" not in text
    else:
        assert text.count("v_mfma_scale_f32_32x32x64_f8f6f4") >= (24 if fname == "hamming_mfma_d.hip" else 128)
    # K1i (round 6) reads its accumulators from inline asm (v_pk_minimum3_f16 with op_sel / v_min3_f32 on the unpacked
    # registers): the compiler counts no wait states for those operands, so the listing must keep the distance the compiler
    # keeps for its own instructions -- 14 by this checker's counting (round 5's listing: `s_nop 5` in front of the v_perm
    # that read a chain's result 8 instructions behind its last MFMA; no finding at 14, five at 15)
    findings = mod.check(out, 14 if fname == "hamming_mfma_i.hip" else 12)
    assert not findings, findings[:5]


def test_k1h_workgroup_column_combine_model():
    """K1h's column partial (hamming_mfma_h.hip::combine_columns), modelled in numpy: the 16 group minima of a column --
    4 waves x 2 lane halves x 2 M-tiles, 16-bit keys (d << 7 | 32 w + 16 g + r), M-tile 0 in the low halves of the 8 parked
    words and M-tile 1 (the block's upper 128 rows) in the high halves -- go through the packed min / max network, are
    widened to (d << 8 | row in the block) and merged: K0 must be the column's best row over the workgroup's 256 rows
    (ties: the lowest row), K1 the best row outside K0's group of 16 -- incl. missing rows (0xFFFF) and penalised columns."""
    r = np.random.Generator(np.random.PCG64(5))

    def pk(op, a, b):                                      # v_pk_min_u16 / v_pk_max_u16
        lo, hi = op(a & 0xFFFF, b & 0xFFFF), op(a >> 16, b >> 16)
        return (hi << 16) | lo

    def pk_merge(a0, a1, c0, c1):
        m = pk(max, a0, c0)
        return pk(min, a0, c0), pk(min, m, pk(min, a1, c1))

    def merge2(a0, a1, c0, c1):
        return min(a0, c0), min(max(a0, c0), min(a1, c1))

    for it in range(4000):
        ties = it % 3 == 0
        words, groups = [], []                             # groups: (key widened by hand, group id)
        for w in range(4):
            for g in range(2):
                halves = []
                for mt in range(2):
                    kind = r.integers(0, 12)
                    rr = int(r.integers(0, 16))
                    if kind == 0:
                        key = 0xFFFF                       # no row of this group exists
                    elif kind == 1:
                        key = 0xBF80 + 32 * w + 16 * g + rr  # a column that does not exist (zero codes + penalty)
                    else:
                        d = int(r.integers(0, 4)) if ties else int(r.integers(0, 257))
                        key = (d << 7) | (32 * w + 16 * g + rr)
                    halves.append(key)
                    if key != 0xFFFF:
                        groups.append((((key >> 7) << 8) | (128 * mt + (key & 0x7F)), (w, g, mt)))
                words.append((halves[1] << 16) | halves[0])
        lo = [pk(min, words[2 * q], words[2 * q + 1]) for q in range(4)]
        hi = [pk(max, words[2 * q], words[2 * q + 1]) for q in range(4)]
        lo[0], hi[0] = pk_merge(lo[0], hi[0], lo[1], hi[1])
        lo[2], hi[2] = pk_merge(lo[2], hi[2], lo[3], hi[3])
        lo[0], hi[0] = pk_merge(lo[0], hi[0], lo[2], hi[2])
        e0, e1, u0, u1 = lo[0] & 0xFFFF, hi[0] & 0xFFFF, lo[0] >> 16, hi[0] >> 16
        k0, k1 = merge2(e0 + (e0 & 0xFF80), e1 + (e1 & 0xFF80), u0 + (u0 & 0xFF80) + 128, u1 + (u1 & 0xFF80) + 128)
        assert k0 <= 0x1FFFF and (k1 >> 8) <= 511           # the one-word partial: K0 << 9 | K1's distance
        groups.sort()
        if not groups:
            assert (k0 >> 8) == 511 and (k1 >> 8) == 511
            continue
        assert k0 == groups[0][0], (it, hex(k0), hex(groups[0][0]))
        assert (k0 & 0xFF) >> 4 == (groups[0][0] & 0xFF) >> 4
        others = [k for k, gid in groups if gid != groups[0][1]]
        if others:
            assert k1 == others[0], (it, hex(k1), hex(others[0]))
        else:
            assert (k1 >> 8) == 511


@pytest.mark.parametrize("fname,kernel,max_spill", [("hamming_mfma_i.hip", "k_scan_sym_mfma_i", 8), ("hamming_mfma_h.hip", "k_scan_sym_mfma_h", 16)])
def test_the_scan_stays_at_three_workgroups_per_cu(tmp_path, fname, kernel, max_spill):
    """What the scan's time depends on and a careless edit loses silently: 3 workgroups per CU need <= 168 VGPRs and <= 53 248 B
    of LDS per workgroup -- LDS is handed out in granules of 2 KB, so 54 272 B are 27 granules and TWO workgroups per CU (K1h:
    3.29 ms; K1i with a fourth slot in its LDS-DMA ring, round 5: 2.84 against 2.52 ms) --, and the spilled registers stay
    few (they sit outside the tile loops).  K1i is the default scan; K1h's kernel is compiled on request only."""
    import os
    import re
    import subprocess
    from plslam_amd import build as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "plslam_amd", "csrc", fname)
    r = subprocess.run([B.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-inline-asm",
                        "-DPLSLAM_BUILD_LEGACY_SCANS=1", "-I" + os.path.join(root, "include"), "-S", "--cuda-device-only", src, "-o", "-"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    seen = 0
    for blk in r.stdout.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if kernel not in name:
            continue
        seen += 1
        lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        assert lds <= 53248 and vgpr <= 168 and spill <= max_spill, (name, lds, vgpr, spill)
    assert seen == 2                                       # the symmetric and the directed instantiation


def test_k1i_inline_asm_reads_accumulators_only_in_the_bookkeeping_macro():
    """K1i's asm statements and the accumulators.  (1) The first MFMA of the loop-carried chain is an asm statement with an
    early-clobber DESTINATION (the compiler's tied form would copy 16 seed registers per tile) -- never an input.  (2) Round 6:
    the bookkeeping macro PLSLAM_MI_EPI2 reads accumulator ELEMENTS from asm (v_pk_minimum3_f16 with op_sel, v_min3_f32 /
    v_min_f32 on the unpacked registers: no v_perm pack).  The compiler counts no MFMA -> VALU wait states for asm operands:
    every such read must sit in that macro (whose uses are separated from the chain's last MFMA by the other chain's MFMAs,
    or by the hand-counted s_nop in front of phase 1 and of the epilogue), and the final listing is checked at the compiler's
    own distance by test_final_isa_has_no_mfma_destination_hazard."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plslam_amd", "csrc", "hamming_mfma_i.hip")).read()
    src = "\n".join(l.split("//")[0] for l in src.split("\n"))
    stmts = [m.group(1) for m in re.finditer(r"asm(?:\s+volatile)?\s*\(([^;]*);", src) if not re.match(r'\s*""', m.group(1))]
    acc = [b for b in stmts if re.search(r"\(m[01]\)", b)]
    assert len(acc) == 1 and acc[0].lstrip().startswith('"v_mfma_f32_32x32x64_f8f6f4') and '"=&v"(m1)' in acc[0]
    assert not re.search(r':\s*"v"\(m[01]\)|,\s*"v"\(m[01]\)', acc[0])            # not an input
    # accumulator elements: pack_acc (a builtin) or the macro's asm statements, nowhere else
    uses = [l for l in src.split("\n") if re.search(r"\bACC\[", l) and "ACC[KS]" not in l and "ACC[0]" not in l]
    ok = re.compile(r'pack_acc\(|asm\("v_pk_minimum3_f16 %0, %0, %1, %2 op_sel:\[0,0,1\] op_sel_hi:\[1,1,0\]"|asm\("v_min3_f32 %0, %0, %1, %2"|asm\("v_min_f32 %0, %1, %2"')
    assert uses and all(ok.search(l) for l in uses), [l for l in uses if not ok.search(l)]
    # the hand-counted wait states are there: in front of phase 1 of a step and in front of the epilogue
    assert 'if (DIRECTED) asm volatile("s_nop 7"); else asm volatile("s_nop 4");' in src and 'asm volatile("s_nop 7\\n\\ts_nop 7");' in src


def test_k1i_stays_at_three_workgroups_per_cu_and_owns_m0(tmp_path):
    """Three workgroups per CU (<= 168 VGPRs, <= 53 248 B of LDS), no spill traffic inside the tile loops (a reload there waits
    for vmcnt(0), i.e. for the whole LDS-DMA prefetch), and M0 -- declared clobbered, not saved -- written by nothing but the
    kernel's own LDS-DMA statements."""
    import os
    import re
    import subprocess
    from plslam_amd import build as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "plslam_amd", "csrc", "hamming_mfma_i.hip")
    r = subprocess.run([B.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-inline-asm",
                        "-I" + os.path.join(root, "include"), "-S", "--cuda-device-only", src, "-o", "-"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    seen = 0
    for blk in r.stdout.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if "k_scan_sym_mfma_i" not in name:
            continue
        seen += 1
        lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        assert lds <= 53248 and vgpr <= 168 and spill <= 8, (name, lds, vgpr, spill)
    assert seen == 2
    # per kernel: between the first and the last MFMA lie the tile loops with their rare paths (column combine, group push,
    # ragged-group penalties): at most a couple of reloads there, none in a stretch between two barriers that is a plain tile
    # step (8 MFMAs, no store of column results, no parked-pair traffic); and every write of m0 is followed (s_nop apart) by
    # the LDS-DMA load
    for body in re.split(r"\n_ZN6plslam17k_scan_sym_mfma_i", r.stdout)[1:]:
        body = body.split("s_endpgm")[0]
        first, last = body.index("v_mfma_f32_32x32x64_f8f6f4"), body.rindex("v_mfma_f32_32x32x64_f8f6f4")
        assert body[first:last].count("scratch_") <= 4, body[first:last].count("scratch_")
        plain = [seg for seg in body[first:last].split("s_barrier")
                 if seg.count("v_mfma_f32_32x32x64_f8f6f4") == 8 and "global_store" not in seg and "ds_write2st64_b64" not in seg]
        assert len(plain) >= 4 and all("scratch_" not in seg for seg in plain), len(plain)
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        for i, l in enumerate(lines):
            if re.match(r"s_mov_b32 m0,", l):
                nxt = [x for x in lines[i + 1:i + 4] if not x.startswith("s_nop")]
                assert nxt and nxt[0].startswith("global_load_lds_dword"), lines[i:i + 4]
            assert not re.search(r"\bm0\b", l) or l.startswith("s_mov_b32 m0,"), l


def test_k1i_workgroup_column_combine_model():
    """K1i's column combine (hamming_mfma_i.hip::combine_columns, the one-word form), modelled in numpy.  The parked keys are
    (d << 5 | 16 g + r) -- the unscaled MFMA leaves five bits under the distance --, so keys of DIFFERENT waves do not order
    like rows: the best row is the minimum of the waves' best keys re-encoded with the wave between distance and tag
    (x + 3 (x & ~31) + 32 v = d << 7 | 32 v + 16 g + r, saturating), the second entry the second smallest parked VALUE (only its
    distance is kept).  Must equal the definition: K0 = the column's best row over the workgroup's 256 rows (ties: the lowest
    row), K1's distance = that of the best row outside K0's group of 16 -- incl. struck-out groups ("none" = 0x7BFF, the
    largest finite half float), penalised columns and distance ties across waves (the case a value-only network gets wrong)."""
    r = np.random.Generator(np.random.PCG64(15))
    NONE = 0x7BFF

    def pk(op, a, b):
        lo, hi = op(a & 0xFFFF, b & 0xFFFF), op(a >> 16, b >> 16)
        return (hi << 16) | lo

    def pk_merge(a0, a1, c0, c1):
        m = pk(max, a0, c0)
        return pk(min, a0, c0), pk(min, m, pk(min, a1, c1))

    def merge2(a0, a1, c0, c1):
        return min(a0, c0), min(max(a0, c0), min(a1, c1))

    def enc(x, v):                                         # v_pk_add_u16 + v_pk_mad_u16 ... clamp on one half
        return min((x & 0xFFE0) * 3 + ((x + 32 * v) & 0xFFFF), 0xFFFF)

    for it in range(6000):
        ties = it % 3 == 0
        words, groups = [], []
        for w in range(4):
            for g in range(2):
                halves = []
                for mt in range(2):
                    kind = r.integers(0, 12)
                    rr = int(r.integers(0, 16))
                    if kind == 0:
                        key = NONE
                    elif kind == 1:
                        key = 0x5000 + 16 * g + rr          # zero codes ("distance 128") + the column penalty
                    else:
                        d = int(r.integers(0, 3)) if ties else int(r.integers(0, 257))
                        key = (d << 5) | (16 * g + rr)
                    halves.append(key)
                    if key < 0x4000:
                        groups.append((((key >> 5) << 8) | (128 * mt + 32 * w + (key & 31)), (w, g, mt)))
                words.append((halves[1] << 16) | halves[0])
        lo = [pk(min, words[2 * q], words[2 * q + 1]) for q in range(4)]
        hi = [pk(max, words[2 * q], words[2 * q + 1]) for q in range(4)]
        e = [(enc(lo[q] >> 16, q) << 16) | enc(lo[q] & 0xFFFF, q) for q in range(4)]
        best = pk(min, pk(min, e[0], e[1]), pk(min, e[2], e[3]))
        lo[0], hi[0] = pk_merge(lo[0], hi[0], lo[1], hi[1])
        lo[2], hi[2] = pk_merge(lo[2], hi[2], lo[3], hi[3])
        lo[0], hi[0] = pk_merge(lo[0], hi[0], lo[2], hi[2])
        e0, u0, e1, u1 = best & 0xFFFF, best >> 16, hi[0] & 0xFFE0, (hi[0] >> 16) & 0xFFE0
        k0, k1 = merge2(e0 + (e0 & 0xFF80), e1 << 3, u0 + (u0 & 0xFF80) + 128, u1 << 3)
        k0c, d1 = min(k0, 0x1FFFF), min(k1 >> 8, 511)     # the word's fields
        groups.sort()
        if not groups:
            assert (k0c >> 8) > 256 and d1 > 256
            continue
        assert k0c == groups[0][0], (it, hex(k0c), hex(groups[0][0]))
        others = [k for k, gid in groups if gid != groups[0][1]]
        if others:
            assert d1 == others[0] >> 8, (it, d1, others[0] >> 8)
        else:
            assert d1 > 256


def test_column_side_decision_equals_the_row_side_one():
    """k_split_post (the kernel behind a column-split scan, two launches per run) decides the matches from the COLUMN side: column
    j's ratio test names the one row i* it can match, row i*'s ratio test names its column m, and (i*, j) is a match iff m = j.
    stvo-pl's match() reads the same set from the row side (row i's candidate m survives iff column m's candidate is i).  Model of
    both on the oracle's two kNN-2 tables: identical tables and counts on random, tie-heavy, tiny and one-column inputs."""
    from oracle import oracle as O
    from plslam_amd import synth
    r = np.random.Generator(np.random.PCG64(20260925))

    def ratio_pick(idx, dist, nnr):
        ok = (idx[:, 1] >= 0) & (dist[:, 0].astype(np.float32) < dist[:, 1].astype(np.float32) * np.float32(nnr))
        return np.where(ok, idx[:, 0], -1)

    for n1, n2, gen in ((700, 300, synth.random_desc), (257, 513, synth.tie_stress_desc), (5, 2, synth.random_desc),
                        (64, 1, synth.random_desc), (300, 300, synth.tie_stress_desc)):
        d2 = gen(r, n2)
        d1 = np.concatenate([synth.noisy_copy(r, d2)[0], gen(r, n1)])[:n1]
        d1 = np.ascontiguousarray(d1[r.permutation(n1)])
        for nnr in (0.6, 0.75, 0.9):
            i12, t12 = O.knn2(d1, d2)
            i21, t21 = O.knn2(d2, d1)
            m12, m21 = ratio_pick(i12, t12, nnr), ratio_pick(i21, t21, nnr)
            rows = np.where((m12 >= 0) & (m21[np.clip(m12, 0, None)] == np.arange(n1)), m12, -1)        # match(): row side
            cols = np.full(n1, -1, np.int64)                                                            # k_split_post: column side
            for j in range(n2):
                istar = m21[j]
                if istar >= 0 and m12[istar] == j:
                    cols[istar] = j
            ref, nref = O.match(d1, d2, nnr, True)
            assert np.array_equal(rows, ref) and np.array_equal(cols, ref), (n1, n2, nnr)
            assert int((cols >= 0).sum()) == nref


def test_dealt_finalize_table_covers_every_entry_once():
    """plan_build (option post_xcd 2) deals the finalize kernel's block table to the 8 XCDs problem by problem -- rows of equal
    length, padding entries behind the short ones -- and workgroup b (which the hardware places on XCD b % 8) takes entry
    (b % 8) * row + b / 8; a capped grid (a multiple of 8 workgroups) walks b += grid.  Model of both: every real entry is taken
    exactly once, by a workgroup of the XCD whose row holds it, a problem's entries all lie in one row, rows differ by at most one
    problem's blocks, and with the contiguous-eighth mapping (post_xcd 1) the same holds for the chunk a workgroup's XCD owns."""
    r = np.random.Generator(np.random.PCG64(77))
    for trial in range(20):
        nprob = int(r.integers(11, 400))
        nblk = r.choice([1, 1, 6, 6, 6, 2, 16], size=nprob)              # row blocks per problem (LBD 1, ORB 6, ...)
        table = [(p, 256 * k) for p in range(nprob) for k in range(int(nblk[p]))]
        # --- the dealing of plan_build
        rows, x = [[] for _ in range(8)], 0
        i = 0
        while i < len(table):
            j = i
            while j < len(table) and table[j][0] == table[i][0]:
                j += 1
            best = x
            for t in range(8):
                c = (x + t) & 7
                if len(rows[c]) < len(rows[best]):
                    best = c
            rows[best] += table[i:j]
            x = (best + 1) & 7
            i = j
        L = max(len(q) for q in rows)
        dealt = [(-1, 0)] * (8 * L)
        for c in range(8):
            dealt[c * L:c * L + len(rows[c])] = rows[c]
        assert max(len(q) for q in rows) - min(len(q) for q in rows) <= int(nblk.max())
        row_of = {}
        for c in range(8):
            for p, _ in rows[c]:
                assert row_of.setdefault(p, c) == c                      # a problem's blocks share one XCD
        # --- the kernel's walk, uncapped and capped
        for grid in (8 * L, 8, 8 * max(1, L // 3), 16):
            grid = min(grid, 8 * L)
            taken = []
            for wg in range(grid):
                b = wg
                while b < 8 * L:
                    blk = (b & 7) * L + (b >> 3)
                    assert blk // L == wg % 8                            # the entry lies in the row of the workgroup's XCD
                    if dealt[blk][0] >= 0:
                        taken.append(dealt[blk])
                    b += grid
            assert sorted(taken) == sorted(table)
        # --- post_xcd 1: a contiguous eighth per XCD over the table in problem order
        n = len(table)
        per = (n + 7) // 8
        taken = [table[(b & 7) * per + (b >> 3)] for b in range(8 * per) if (b & 7) * per + (b >> 3) < n]
        assert sorted(taken) == sorted(table)
