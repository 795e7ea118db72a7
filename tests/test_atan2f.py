"""cmx_atan2f.h (the histogram kernels' atan2) against this machine's libm, on the host: the
header is plain f32 C++, so the device computes the same bits (checked on the GPU through the
histogram itself, tests/test_gpu_r2_paths.py::test_compute_histogram_equals_the_oracle)."""
import os
import platform
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# cmx_atan2f.h restates glibc's fdlibm-derived atanf / atan2f as shipped up to 2.35 (this image).
# Newer glibc releases replace the float functions by correctly rounded ones (CORE-MATH): there
# the header no longer equals the machine's libm, and neither does a reference built on that
# machine -- the device histogram is then within 1 ulp of it instead of equal.
_LIBC = platform.libc_ver()
_PINNED = _LIBC[0] == "glibc" and tuple(int(x) for x in _LIBC[1].split(".")[:2]) <= (2, 35)


@pytest.mark.skipif(not _PINNED, reason=f"cmx_atan2f.h is pinned on glibc <= 2.35's atan2f, this is {_LIBC}")
def test_header_equals_libm_bit_for_bit(tmp_path):
    exe = tmp_path / "atan2f_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off",
                    "-I", os.path.join(ROOT, "cartographer_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "atan2f_check.cc"), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe), "30000000", "61"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "atanf_mismatches 0" in out.stdout and "atan2f_mismatches 0" in out.stdout
