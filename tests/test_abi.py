"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/plslam_hip.h declares; there is no CPU fallback (ctx creation fails without a device)."""
import ctypes
import os
import re

import pytest

import plslam_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "plslam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(plslam_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    syms = _header_symbols()
    assert len(syms) >= 20
    lib = ctypes.CDLL(plslam_amd.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(plslam_amd.ABI_SYMBOLS) == syms  # the binding declares exactly the header's surface


def test_abi_version_and_strerror():
    L = plslam_amd.load()
    assert L.plslam_abi_version() == plslam_amd.capi.ABI_VERSION == 5
    assert L.plslam_strerror(0) == b"ok"
    assert b"device" in L.plslam_strerror(-2)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the ENODEV path is exercised on CPU-only hosts")
    with pytest.raises(plslam_amd.PlslamError) as e:
        plslam_amd.Context(0)
    assert e.value.code == -2


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU implementation)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "plslam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), os.path.join(dirpath, f)
    assert "oracle" not in open(os.path.join(ROOT, "include", "plslam_hip.h")).read()


def test_every_locked_entry_point_selects_the_context_device():
    """A host thread starts on device 0 whatever device its context lives on (StVO::setDevice(1), Context(rank > 0) from the
    local-mapping thread): every entry point that takes the context lock must switch to ctx->device before it allocates,
    copies or launches.  Structural check of the sources (the run-time check needs two GPUs: test_gpu_match)."""
    import re
    n = 0
    for f in sorted(os.listdir(os.path.join(ROOT, "plslam_amd", "csrc"))):
        if not f.endswith(".hip"):
            continue
        lines = open(os.path.join(ROOT, "plslam_amd", "csrc", f)).read().split("\n")
        for i, ln in enumerate(lines):
            if "std::lock_guard<std::mutex>" in ln and "->mu" in ln:
                n += 1
                assert re.search(r"\bDeviceGuard\s+\w+\(", lines[i + 1]), f"{f}:{i + 1}: lock without DeviceGuard"
    assert n >= 20


def test_every_symbol_has_a_declared_signature():
    """ctypes guesses int for anything undeclared (floats are refused, pointers truncated): every entry point of the
    binding carries explicit argtypes."""
    L = plslam_amd.load()
    missing = [s for s in plslam_amd.ABI_SYMBOLS if getattr(L, s).argtypes is None]
    assert not missing, missing


def test_no_device_scope_fence_and_no_flat_access_in_any_kernel():
    """A device-scope fence is an L2 write-back + invalidate on gfx950 (buffer_wbl2 / buffer_inv) whose cost grows with what the
    rest of the chip has written (it doubled the windowed matcher's time per problem under load): every exchange through
    memory inside these kernels is between lanes of one workgroup.  Checked in the ISA of every kernel source.
    Also: no FLAT memory instruction.  A pointer read from a launch table is generic to the compiler, and a generic access
    counts on lgkmcnt as well as vmcnt (every LDS wait then also waits for it) and cannot take a scalar base; the kernels
    spell the global address space out where they dereference (common.hpp: g_())."""
    import re
    import subprocess
    from plslam_amd import build as B
    csrc = os.path.join(ROOT, "plslam_amd", "csrc")
    for src in B.SOURCES:
        r = subprocess.run([B.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                            "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", os.path.join(csrc, src), "-o", "-"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        assert "buffer_wbl2" not in r.stdout and "buffer_inv" not in r.stdout, src
        flat = re.findall(r"^\s+(flat_(?:load|store|atomic)\w*)", r.stdout, flags=re.M)
        assert not flat, (src, len(flat), flat[:3])


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/plslam_hip.h compiles on its own as C99 (and as C++11), pedantic, no warnings."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "plslam_hip.h"\nint main(void) { return 0; }\n')
    for cmd in (["gcc", "-std=c99"], ["g++", "-std=c++11", "-x", "c++"]):
        r = subprocess.run(cmd + ["-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), "-fsyntax-only",
                                  str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]


def test_the_library_refuses_other_device_targets(tmp_path):
    """K1h / K1i rely on gfx9's in-order vmcnt and on gfx950 instructions through inline asm.  build.py force-includes
    gfx950_only.hpp into every translation unit: the same flags with another --offload-arch must fail at the #error, not compile."""
    import shutil
    import subprocess
    from plslam_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    assert "-include" in B.FLAGS and B.FLAGS[B.FLAGS.index("-include") + 1].endswith("gfx950_only.hpp")
    src = tmp_path / "t.hip"
    src.write_text("#include <hip/hip_runtime.h>\n__global__ void k(int* p) { *p = 1; }\n")
    flags = [f for f in B.FLAGS if f not in ("-shared", "--offload-arch=gfx950")]
    ok = subprocess.run([hipcc, "--offload-arch=gfx950"] + flags + ["-c", "--cuda-device-only", str(src), "-o", str(tmp_path / "a.o")],
                        capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    bad = subprocess.run([hipcc, "--offload-arch=gfx942"] + flags + ["-c", "--cuda-device-only", str(src), "-o", str(tmp_path / "b.o")],
                         capture_output=True, text=True)
    assert bad.returncode != 0 and "gfx950" in bad.stderr
