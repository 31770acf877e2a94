"""CPU: the library's host code under AddressSanitizer and UndefinedBehaviorSanitizer.

`make -C nerf_amd/csrc asan` compiles every translation unit for the host only (a few seconds) with both
sanitizers.  Under the sanitizer runtime this test then (i) sweeps the sizing functions and the weight-gradient
work division over sample counts 1 .. 10^6, both architectures, one and two networks (tests/asan_sweep.py), and
(ii) runs the argument-validation tests of tests/test_cabi.py against that library."""
import os
import subprocess
import sys

import pytest

from util import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
CLANG = "/opt/rocm/lib/llvm/bin/clang"


@pytest.fixture(scope="module")
def asan_env():
    if not (os.path.exists(HIPCC) and os.path.exists(CLANG)):
        pytest.skip("no ROCm compiler here")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "nerf_amd", "csrc"), "asan", "-j", "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lib = os.path.join(ROOT, "build", "asan", "libnerf_hip_asan.so")
    rt = subprocess.run([CLANG, "--print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(lib) and os.path.exists(rt), (lib, rt)
    return dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
                NERF_TEST_LIB=lib)


def test_sizing_and_work_division_sweep_under_sanitizers(asan_env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_sweep.py")], env=asan_env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "asan sweep ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_cabi_argument_checks_under_sanitizers(asan_env):
    """tests/test_cabi.py's own library-level tests, with NERF_TEST_LIB pointing at the sanitizer build"""
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(ROOT, "tests", "test_cabi.py"), "-k",
                        "exports or argument_errors or fp32_program or fp32_backward", "-p", "no:cacheprovider"],
                       env=asan_env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
