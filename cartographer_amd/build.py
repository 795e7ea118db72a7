"""Builds the in-tree native libraries.

  cartographer_amd/lib/libcartographer_mi355x.so   HIP kernels + C ABI (hipcc, gfx950)
  cartographer_amd/lib/libcmx_synth.so             host-only fixture tooling (g++)
  tools/bin/{gather_ceiling,row_gather_ceiling}    stand-alone micro-benchmarks (hipcc, gfx950)

No JIT cache: the .so files live in the tree so they travel to the GPU box.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: bit-exact parity with the reference's non-FMA arithmetic.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
        [os.path.join(HERE, "..", "include", "cartographer_mi355x.h")]
    objs = []
    procs = []
    for src in srcs:
        obj = os.path.join(LIB, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            cmd = [HIPCC] + HIP_FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    out = os.path.join(LIB, "libcartographer_mi355x.so")
    if force or procs or _newer(out, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build_synth(force=False):
    os.makedirs(LIB, exist_ok=True)
    srcs = [os.path.join(CSRC, "host", f) for f in ("probability_grid_builder.cc", "hybrid_grid_builder.cc", "synth.cc",
                                                  "thread_driver.cc")]
    hdrs = [os.path.join(CSRC, "host", h) for h in ("probability_grid_builder.h",
                                                  "hybrid_grid_builder.h")]
    out = os.path.join(LIB, "libcmx_synth.so")
    if force or _newer(out, srcs + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-pthread", "-o", out] + srcs)
    return out


def build_tools(force=False):
    """The stand-alone micro-benchmarks under tools/ (gather ceilings the rooflines are priced on):
    tools/bin/*, cross-compiled like the library so that they travel to the GPU box."""
    root = os.path.dirname(HERE)
    out_dir = os.path.join(root, "tools", "bin")
    os.makedirs(out_dir, exist_ok=True)
    built = []
    for name in ("gather_ceiling", "row_gather_ceiling", os.path.join("probes", "lds_unaligned"),
                 os.path.join("probes", "launch_rate"), os.path.join("probes", "dispatch_rate")):
        src = os.path.join(root, "tools", name + ".hip")
        name = os.path.basename(name)
        out = os.path.join(out_dir, name)
        if os.path.exists(src) and (force or _newer(out, [src])):
            subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-pthread", src, "-o", out])
        built.append(out)
    return built


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_synth(force), build_tools(force)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
