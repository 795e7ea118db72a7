"""cmx_atan2f.h (the histogram kernels' atan2) against this machine's libm, on the host: the
header is plain f32 C++, so the device computes the same bits (checked on the GPU through the
histogram itself, tests/test_gpu_r2_paths.py::test_compute_histogram_equals_the_oracle)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_equals_libm_bit_for_bit(tmp_path):
    exe = tmp_path / "atan2f_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off",
                    "-I", os.path.join(ROOT, "cartographer_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "atan2f_check.cc"), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe), "30000000", "61"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "atanf_mismatches 0" in out.stdout and "atan2f_mismatches 0" in out.stdout
