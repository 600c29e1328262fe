"""CPU: the C-ABI libraries load and export every symbol include/flipv.h and host/flipv_host.h declare
(no compute calls here -- there is no GPU in the build container and no CPU fallback in the product)."""
import ctypes
import os
import re

from helpers import ROOT


def declared_functions(header):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(flipvh?_[a-z0-9_]+)\s*\(", src)))


def test_flipv_header_symbols_exported():
    from flipviscosity3d_amd import capi
    names = declared_functions(os.path.join(ROOT, "include", "flipv.h"))
    assert len(names) >= 35
    lib = ctypes.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libflipv.so does not export %s" % n
    # the ctypes binding covers the whole header, nothing more
    assert sorted(capi.SYMBOLS) == names


def test_host_header_symbols_exported():
    from flipviscosity3d_amd import hostapi
    names = declared_functions(os.path.join(ROOT, "flipviscosity3d_amd", "host", "flipv_host.h"))
    lib = ctypes.CDLL(hostapi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libflipv_host.so does not export %s" % n
    assert sorted(hostapi.SYMBOLS) == names


def test_struct_layouts_match_header():
    """ctypes mirrors of the ABI structs: sizes as the C compiler lays them out"""
    import subprocess
    import tempfile
    from flipviscosity3d_amd import capi
    code = '#include <stdio.h>\n#include "flipv.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(flipv_params), ' \
           'sizeof(flipv_solve_info), sizeof(flipv_stats), sizeof(flipv_kernel_stats), sizeof(flipv_debug_params));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write(code)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(capi.Params), ctypes.sizeof(capi.SolveInfo), ctypes.sizeof(capi.Stats),
                     ctypes.sizeof(capi.KernelStats), ctypes.sizeof(capi.DebugParams)]


def test_abi_version_is_the_headers():
    """flipv_abi_version(): what a binding checks before it passes a struct (the structs' layout changes with FLIPV_VERSION)"""
    import re
    from flipviscosity3d_amd import capi
    L = capi.load()
    hdr = open(os.path.join(ROOT, "include", "flipv.h")).read()
    assert L.flipv_abi_version() == int(re.search(r"#define FLIPV_VERSION (\d+)", hdr).group(1)) == 6


def test_default_params_are_the_reference_constants():
    from flipviscosity3d_amd import capi
    L = capi.load()
    p = capi.Params()
    assert L.flipv_default_params(ctypes.byref(p)) == 0
    assert p.cfl_number == 5.0 and abs(p.min_frac - 0.01) < 1e-9 and abs(p.pic_ratio - 0.05) < 1e-9  # fluidsimulation.h:128-130
    assert p.pressure_tolerance == 1e-9                                                               # pressuresolver.h:224
    assert p.viscosity_tolerance == 1e-6 and p.viscosity_max_iterations == 700                        # viscositysolver.h:200-202
    assert p.viscosity_accept_tolerance == 10.0


def test_create_without_gpu_fails_loudly():
    """no silent CPU fallback: on a box without a HIP device flipv_create must return an error"""
    import torch
    from flipviscosity3d_amd import capi
    if torch.cuda.is_available():
        return
    try:
        capi.Context(8, 8, 8, 0.125)
    except capi.FlipvError as e:
        assert "no HIP device" in str(e) or "-2" in str(e) or "failed" in str(e)
    else:
        raise AssertionError("flipv_create succeeded without a GPU")
