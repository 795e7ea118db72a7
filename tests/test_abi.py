"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports
every symbol include/cartographer_mi355x.h declares, and refuses to compute
without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cartographer_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cmx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_loader_agree():
    from cartographer_amd import _lib
    assert _declared_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from cartographer_amd import _lib
    L = _lib.lib()
    for name in _declared_symbols():
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    assert b"gfx950" in L.cmx_version()


def test_no_cpu_fallback_without_device():
    from cartographer_amd import _lib, scan_matching as sm
    L = _lib.lib()
    if L.cmx_device_count() > 0:
        pytest.skip("a HIP device is present")
    grid = sm.Grid2D(np.zeros((4, 4), np.uint16), 0.05, 0.1, 0.1)
    with pytest.raises(_lib.CmxError) as e:
        sm.FastCorrelativeScanMatcher2D(grid, 3)
    assert e.value.status == _lib.DEVICE_ERROR
    assert "no CPU fallback" in str(e.value)
    m = sm.RealTimeCorrelativeScanMatcher2D(0.1, 0.1, 0.0, 0.0)
    with pytest.raises(_lib.CmxError) as e:
        m.match(sm.Rigid2d(), np.zeros((3, 3), np.float32), grid)
    assert e.value.status == _lib.DEVICE_ERROR


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "cartographer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cc")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "liboracle" not in text and \
                    "oracle_2d" not in text and "oracle_3d" not in text, os.path.join(dirpath, f)


def test_header_is_plain_c_and_links(tmp_path):
    """include/cartographer_mi355x.h compiles as C (gcc -std=c99 -pedantic) and a C
    program links against the shared library and runs (no GPU: DEVICE_ERROR, exit 0)."""
    import subprocess
    from cartographer_amd import _lib
    exe = str(tmp_path / "c_abi_demo")
    lib_dir = os.path.dirname(_lib.SO_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror",
                           "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-o", exe,
                           "-L", lib_dir, "-lcartographer_mi355x",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cartographer_mi355x" in out.stdout


def test_struct_layouts_agree_with_the_header(tmp_path):
    """Every struct the ctypes mirror declares has the size and member offsets a C compiler
    gives the header's struct of the same role (a field added on one side only -- as happened to
    cmx_match_stats this round -- would otherwise corrupt the caller's stack silently)."""
    import subprocess
    from cartographer_amd import _lib
    pairs = {                      # header struct -> ctypes mirror
        "cmx_pose2d": _lib.Pose2d, "cmx_pose3d": _lib.Pose3d,
        "cmx_grid2d_limits": _lib.Grid2DLimits, "cmx_rt_options": _lib.RtOptions,
        "cmx_fast2d_options": _lib.Fast2DOptions, "cmx_fast3d_options": _lib.Fast3DOptions,
        "cmx_match_stats": _lib.MatchStats, "cmx_ceres2d_options": _lib.Ceres2DOptions,
        "cmx_ceres_summary": _lib.CeresSummary, "cmx_ceres3d_options": _lib.Ceres3DOptions,
        "cmx_ceres3d_pair": _lib.Ceres3DPair, "cmx_voxel": _lib.Voxel,
        "cmx_intensity_voxel": _lib.IntensityVoxel, "cmx_candidate2d": _lib.Candidate2D,
        "cmx_node_data3d": _lib.NodeData3D, "cmx_result3d": _lib.Result3D,
    }
    header = open(os.path.join(ROOT, "include", "cartographer_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "cartographer_mi355x.h"',
             'int main(void) {']
    members = {}
    for name in pairs:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S)
        assert body, f"{name} not found in the header"
        fields = []
        for decl in body.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):            # "double x, y, theta" / "double t[3]"
                ident = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[[^\]]*\])?\s*$", part.strip())
                fields.append(ident[0])
        members[name] = fields
        lines.append(f'  printf("{name} %zu", sizeof({name}));')
        for fld in fields:
            lines.append(f'  printf(" %zu", offsetof({name}, {fld}));')
        lines.append('  printf("\\n");')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                           exe])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        words = line.split()
        name, size, offsets = words[0], int(words[1]), [int(w) for w in words[2:]]
        mirror = pairs[name]
        assert C.sizeof(mirror) == size, (name, C.sizeof(mirror), size)
        got = [getattr(mirror, f[0]).offset for f in mirror._fields_]
        assert len(got) == len(offsets), (name, members[name], [f[0] for f in mirror._fields_])
        assert got == offsets, (name, got, offsets)


def test_debug_switches_the_bench_and_the_tools_rely_on_exist():
    """cmx_debug_set works without a device (it only stores the value): `timing` brackets the calls
    with HIP events for cmx_match_stats' *_ms fields (bench.py switches it on for its instrumented
    passes only), the RT-2D batch knobs drive tools/c1_probe.py; an unknown name is an error, not
    a silent no-op."""
    from cartographer_amd import _lib
    try:
        _lib.debug_set(timing=1, rt2d_parts=2, rt2d_grid_share=2, no_copy_kernels=1,
                       no_direct_results=1)
        with pytest.raises(_lib.CmxError):
            _lib.debug_set(no_such_switch=1)
    finally:
        _lib.debug_reset()
