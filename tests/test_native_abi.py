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


def test_no_c_symbol_is_exported_behind_the_headers_back(pkg):
    """Every `vcg_*` function with C linkage in the library is declared in include/vcg.h (internal helpers keep C++ linkage;
    diagnostic hooks are named here)."""
    import subprocess
    path = pkg._native.build()
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("vcg_")})
    diagnostic = {"vcg_debug_set_stamp"}                                   # stamp hook of the -DVCG_STAMP diagnostic build (a no-op here)
    extra = [s for s in exported if s not in header_symbols() and s not in diagnostic]
    assert not extra, f"exported with C linkage but not declared in vcg.h: {extra}"


def test_python_binding_covers_the_header(pkg):
    assert sorted(pkg._native.SIGNATURES) == header_symbols()
    lib = pkg._native.lib()
    assert lib.vcg_abi_version() == 6


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


def _desc(n, h, w, cin, cout, k, stride=1, pad=1, reflect=1, ups=1, act=0, cin_log=None, cout_log=None):
    cd = (ctypes.c_int32 * 16)()
    cd[0:14] = [n, h, w, cin, cout, k, k, stride, pad, reflect, ups, act, cin_log or cin, cout_log or cout]
    return cd


def test_launch_plans_are_host_side_and_consistent(pkg):
    """Workspace / packed-weight sizes come from the same planners the launches use (no GPU needed): the Winograd
    path is taken exactly for 3x3 stride-1 layers with Kc >= 128, Cout >= 64 on even maps, the kw-folded path for the
    3-channel head, everything else stays on the direct kernels."""
    lib = pkg._native.lib()
    f4 = 4
    # R-block conv: Winograd in all three directions
    r = _desc(8, 16, 16, 1024, 1024, 3)
    T, Tp, kc, co = 8 * 8 * 8, 8 * 9 * 9, 1024, 1024
    # Wf (fp32) + U + Ud: the Winograd-transformed kernels as pre-split fp16 planes (2 pieces x 2 bytes = 1 float per value)
    # + the 16-float header that keeps the kernel's largest magnitude (the planes hold w / s)
    assert lib.vcg_pack_weight_floats(r) == 9 * kc * co + 2 * 16 * kc * co + 16
    assert lib.vcg_conv_fwd_workspace(r) == 16 * T * (kc + co) * f4 + 512
    assert lib.vcg_conv_dgrad_workspace(r) == 16 * Tp * (kc + co) * f4 + 1024       # V and M over the padded domain's 9 x 9 tiles
    assert lib.vcg_conv_wgrad_workspace(r) > 16 * T * (kc + co) * f4                  # transforms + stream-K slabs
    # same channels on an odd map: direct kernels, K-sliced forward (slabs) but no Winograd buffers
    odd = _desc(8, 15, 15, 1024, 1024, 3)
    assert lib.vcg_pack_weight_floats(odd) == lib.vcg_pack_weight_floats(r)           # packing sees no image size
    assert lib.vcg_conv_fwd_workspace(odd) < 16 * 8 * 7 * 7 * (kc + co) * f4
    # D block (folded PixelUnshuffle): Kc = 4 * 128
    d2 = _desc(8, 128, 128, 128, 256, 3, ups=2)
    assert lib.vcg_pack_weight_floats(d2) == 9 * 512 * 256 + 2 * 16 * 512 * 256 + 16
    assert lib.vcg_conv_fwd_workspace(d2) == 16 * (8 * 32 * 32) * (512 + 256) * f4 + 512
    # small channel counts: no transformed copies
    u4 = _desc(8, 256, 256, 32, 64, 3)
    # Wf + the pre-split weight planes of the direct split-operand kernels: WFT [Cout][K/32][2][32] (forward), WFD
    # [(tap, c)][Cout/32][2][32] (data gradient): 32 floats per (row, 32-wide block); + the header
    assert lib.vcg_pack_weight_floats(u4) == 9 * 32 * 64 + 64 * 9 * 32 + 9 * 32 * 2 * 32 + 16
    assert lib.vcg_conv_fwd_workspace(u4) == 0
    # ... and it runs on the LDS-slab kernels: the data gradient over the padded domain (H + 2) x (W + 2), folded afterwards
    assert lib.vcg_conv_dgrad_workspace(u4) == 8 * 258 * 258 * 32 * f4 + 256
    # ... and its weight gradient on the row-ring kernel: one [288][64] partial per workgroup + 16 group sums (conv_ring.hip)
    def ring_ws(n, yd, xd, nr, sub=1):
        nseg = (xd + 31) // 32
        nrs = max(768 // (n * nseg * sub), 1)
        rows = max((yd + nrs - 1) // nrs, 8)
        nwg = n * nseg * ((yd + rows - 1) // rows)
        per = (nwg + 15) // 16
        return (sub * nwg + sub * ((nwg + per - 1) // per)) * nr * 64 * f4 + 256
    assert 0 <= lib.vcg_conv_wgrad_workspace(u4) - (ring_ws(8, 256, 256, 288) + 255) // 256 * 256 <= 1 << 20     # + bias column sums
    # the latent convs 1024 -> 64: Kc Cout / (Kc + Cout) = 60 is under every Winograd gate (forward 64, data gradient 80):
    # no transformed copies, the planes of the direct split-operand kernels instead
    mu = _desc(8, 16, 16, 1024, 64, 3)
    assert lib.vcg_pack_weight_floats(mu) == 9 * 1024 * 64 + 64 * (9216 // 32) * 32 + 9 * 1024 * 2 * 32 + 16
    # decoder head 64 -> 3 (pitch 4), 7x7: Wf + the kw-folded copy [(kh, c)][32]; P buffer over the padded columns
    head = _desc(8, 256, 256, 64, 4, 7, pad=3, cout_log=3)
    # (+ that copy's pre-split planes [32][7 * 64 / 32][2][32] for the LDS-slab column kernel, + the header)
    assert lib.vcg_pack_weight_floats(head) == ((49 * 64 * 4 + 63) // 64) * 64 + 7 * 64 * 32 + 7 * 64 * 32 + 16
    assert lib.vcg_conv_fwd_workspace(head) == 8 * 256 * (256 + 6) * 32 * f4 + 256
    # its weight gradient: the ring kernel over the PADDED input pixels (262 x 262), 7 tap rows x 32 columns per partial
    assert 0 <= lib.vcg_conv_wgrad_workspace(head) - (ring_ws(8, 262, 262, 224) + 255) // 256 * 256 <= 1 << 20
    # encoder stem 3 -> 64: the data gradient takes the folded path (padded-domain dxp + P), the forward does not
    stem = _desc(8, 256, 256, 4, 64, 7, pad=3, cin_log=3)
    assert lib.vcg_pack_weight_floats(stem) == ((49 * 4 * 64 + 63) // 64) * 64 + 7 * 64 * 32 + 7 * 64 * 32 + 64 * 7 * 32 + 16     # K = 196 -> 7 blocks
    assert lib.vcg_conv_dgrad_workspace(stem) >= 8 * 262 * 262 * 4 * f4 + 8 * 262 * (256 + 12) * 32 * f4
    # stride-2 discriminator conv: direct everywhere; forward and data gradient may slice K (whole output- / input-sized
    # slabs; the data gradient's blockIdx.z enumerates parity class + 4 x slice)
    disc = _desc(8, 128, 128, 64, 128, 4, stride=2, pad=1)
    assert lib.vcg_pack_weight_floats(disc) == 16 * 64 * 128 + 128 * 32 * 32 + 16 * 64 * 4 * 32 + 16
    assert (lib.vcg_conv_fwd_workspace(disc) - 256) % (8 * 64 * 64 * 128 * f4) in (0, (8 * 64 * 64 * 128 * f4) - 256)
    assert lib.vcg_conv_dgrad_workspace(disc) % (8 * 128 * 128 * 64 * f4) in (0, 256)


def test_product_path_refuses_cpu_tensors(pkg):
    m = pkg.Networks.S(4, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 4, 4, 4))
    with pytest.raises(RuntimeError):
        pkg.optim.FusedAdam(m.parameters(), lr=1e-3)
