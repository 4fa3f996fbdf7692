"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/vcg.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vcg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vcg_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol(pkg):
    path = pkg._native.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    syms = header_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in vcg.h but not exported: {missing}"


def test_python_binding_covers_the_header(pkg):
    assert sorted(pkg._native.SIGNATURES) == header_symbols()
    lib = pkg._native.lib()
    assert lib.vcg_abi_version() == 1


def test_code_object_targets_gfx950(pkg):
    path = pkg._native.build()
    blob = open(path, "rb").read()
    assert b"gfx950" in blob
    # the conv kernels must really be on the fp32 matrix core
    tool = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if os.path.exists(tool):
        out = subprocess.run(["bash", "-c", f"cd /tmp && /opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input={path} 2>/dev/null | head -5"],
                             capture_output=True, text=True).stdout
        assert "gfx950" in out or out == ""


def test_bad_arguments_are_rejected_without_touching_the_gpu(pkg):
    lib = pkg._native.lib()
    cd = (ctypes.c_int32 * 16)()
    cd[0], cd[1], cd[2], cd[3], cd[4] = 1, 8, 8, 3, 8          # Cin pitch 3 is illegal
    cd[5] = cd[6] = 3
    cd[7], cd[8], cd[9], cd[10], cd[12], cd[13] = 1, 1, 1, 1, 3, 8
    assert lib.vcg_conv_wgrad_workspace(cd) == 0
    rc = lib.vcg_conv_fwd(None, None, None, None, cd, None, 0, None)
    assert rc != 0
    assert b"multiple of 4" in lib.vcg_last_error()
    cd[3] = 4
    cd[1] = cd[2] = 1                                           # reflect pad 1 on a 1x1 map is illegal
    assert lib.vcg_conv_fwd(None, None, None, None, cd, None, 0, None) != 0
    assert b"reflect" in lib.vcg_last_error()


def test_product_path_refuses_cpu_tensors(pkg):
    m = pkg.Networks.S(4, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 4, 4, 4))
    with pytest.raises(RuntimeError):
        pkg.optim.FusedAdam(m.parameters(), lr=1e-3)
