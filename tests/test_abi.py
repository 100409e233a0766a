"""The C ABI: every symbol include/msfm.h declares is exported by libmsfm.so and bound by the
ctypes host; struct layouts agree with a C compiler's; the product never touches the oracle; and
without a GPU the library refuses to work instead of falling back."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from metricsfm_amd import _abi as A
from metricsfm_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "msfm.h")


def declared_functions():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(msfm_[a-z0-9_]+)\s*\(", txt)) - {"msfm_allreduce_fn"})


def test_every_declared_symbol_is_exported_and_bound():
    fns = declared_functions()
    assert len(fns) >= 30
    L = capi.lib()
    for f in fns:
        assert hasattr(L, f), "libmsfm.so does not export %s" % f
    assert sorted(capi.SYMBOLS) == fns
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (msfm_\w+)", out))
    assert exported == set(fns), exported ^ set(fns)  # nothing undeclared leaks out either
    assert L.msfm_version() == 100


def test_struct_layouts_match_the_c_compiler(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "msfm.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(msfm_ba_problem),sizeof(msfm_ba_options),sizeof(msfm_ba_iteration),sizeof(msfm_ba_summary),'
                   'sizeof(msfm_tracks),sizeof(msfm_kernel_stat),offsetof(msfm_ba_problem,gps_weight),'
                   'offsetof(msfm_ba_summary,solve_ms),offsetof(msfm_ba_options,jacobi_scaling));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)], text=True).split()]
    want = [C.sizeof(A.BaProblem), C.sizeof(A.BaOptions), C.sizeof(A.BaIteration), C.sizeof(A.BaSummary), C.sizeof(A.Tracks),
            C.sizeof(A.KernelStat), A.BaProblem.gps_weight.offset, A.BaSummary.solve_ms.offset, A.BaOptions.jacobi_scaling.offset]
    assert got == want


def test_defaults_are_the_ceres_defaults():
    o = capi.default_options()
    assert (o.max_num_iterations, o.huber_delta, o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (200, 1.0, 1e-6, 1e-10, 1e-8)
    assert (o.initial_trust_region_radius, o.max_trust_region_radius, o.min_relative_decrease) == (1e4, 1e16, 1e-3)
    assert (o.min_lm_diagonal, o.max_lm_diagonal, o.max_num_consecutive_invalid_steps, o.jacobi_scaling) == (1e-6, 1e32, 5, 1)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(capi.MsfmError) as e:
        capi.Context(0)
    assert e.value.code == A.MSFM_E_DEVICE


def test_product_never_uses_the_oracle():
    pkg = os.path.join(ROOT, "metricsfm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".cc")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert "msfm_oracle" not in txt and "liboracle" not in txt and "orc_" not in txt, f
    out = subprocess.check_output(["ldd", capi.LIB_PATH], text=True)
    assert "oracle" not in out


def test_rccl_interface_matches_rccl_h():
    """libmsfm declares the few RCCL types and values it uses by hand (metricsfm_amd/csrc/rccl_iface.h; librccl is opened with
    dlopen).  tests/rccl_iface_check.cpp includes the real rccl.h beside them and static_asserts the id size, ncclFloat64,
    ncclSum, ncclMax and the parameter lists of every entry point the library resolves - compiling it is the test."""
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no rccl.h in this image")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "rccl_iface_check.cpp")])
