"""Builds and runs the C++ host-shim test (plslam_amd/host/stvo_match.hpp = StVO::match drop-in,
lba_rows.hpp = the LBA row builder) against the C-ABI library on the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp):
    exe = os.path.join(tmp, "test_host_shim")
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-pthread",
           os.path.join(ROOT, "tests", "cpp", "test_host_shim.cpp"),
           "-I" + os.path.join(ROOT, "include"),
           "-L" + os.path.join(ROOT, "plslam_amd", "lib"), "-lplslam_hip",
           "-L" + os.path.join(ROOT, "oracle"), "-lplslam_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "plslam_amd", "lib"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def test_host_shim_compiles_against_the_abi(tmp_path):
    """CPU: the shim headers compile and link against the C-ABI library (no device needed)."""
    _compile(str(tmp_path))


@pytest.mark.gpu
def test_host_shim_runs_and_matches_oracle(tmp_path):
    exe = _compile(str(tmp_path))
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(res.stdout[-2000:], res.stderr[-2000:])
    assert res.returncode == 0, res.stdout[-2000:]
    assert "all checks passed" in res.stdout
