"""ctypes binding of libvcg.so (the C ABI declared in include/vcg.h).

There is no CPU or eager-PyTorch fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.  `build()` compiles the HIP sources for gfx950 in-tree.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# VCG_LIBVCG: load another build of the same sources (A/B runs of kernel variants, tools/); the default is the in-tree library
LIB_PATH = os.environ.get("VCG_LIBVCG") or os.path.join(_HERE, "libvcg.so")
SOURCES = ["conv_igemm.hip", "conv_thin.hip", "conv_thinin.hip", "conv_wino.hip", "conv_slab.hip", "conv_ring.hip", "gemm_split.hip", "norm.hip", "misc.hip", "input.hip"]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "vcg.h")

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_Z = _c.c_size_t
_F = _c.c_float
_U64 = _c.c_uint64
_I32P = _c.POINTER(_c.c_int32)
_U64P = _c.POINTER(_c.c_uint64)

# name -> (restype, argtypes); mirrors include/vcg.h one to one
SIGNATURES = {
    "vcg_abi_version": (_I, []),
    "vcg_last_error": (_c.c_char_p, []),
    "vcg_amax_hint": (None, [_U64, _U64]),
    "vcg_amax_last": (_U64, []),
    "vcg_amax_measure": (_U64, [_P, _Z, _P]),
    "vcg_amax_valid": (_I, [_U64]),
    "vcg_conv_fwd_in_h": (_I, [_P, _P, _P, _P, _P, _P, _F, _P, _I32P, _P, _Z, _U64, _P]),
    "vcg_conv_dgrad_h": (_I, [_P, _P, _P, _I32P, _P, _Z, _U64, _P]),
    "vcg_conv_wgrad_saved_h": (_I, [_P, _P, _P, _P, _P, _I32P, _P, _Z, _U64, _U64, _P]),
    "vcg_in_apply_h": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _U64P, _P]),
    "vcg_in_bwd_h": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _U64P, _P]),
    "vcg_in_bwd_bias": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _Z, _U64P, _P]),
    "vcg_act_bwd_h": (_I, [_P, _P, _P, _Z, _I, _U64P, _P]),
    "vcg_profile_enable": (_I, [_I]),
    "vcg_profile_read": (_c.c_long, [_c.c_char_p, _Z]),
    "vcg_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "vcg_nhwc_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "vcg_fill": (_I, [_P, _F, _Z, _P]),
    "vcg_pack_weight_floats": (_Z, [_I32P]),
    "vcg_pack_weight": (_I, [_P, _P, _I32P, _P]),
    "vcg_conv_reads_wf": (_I, [_I32P]),
    "vcg_conv_fwd_workspace": (_Z, [_I32P]),
    "vcg_conv_fwd": (_I, [_P, _P, _P, _P, _I32P, _P, _Z, _P]),
    "vcg_conv_fwd_in_workspace": (_Z, [_I32P]),
    "vcg_conv_saved_floats": (_Z, [_I32P]),
    "vcg_conv_fwd_in": (_I, [_P, _P, _P, _P, _P, _P, _F, _P, _I32P, _P, _Z, _P]),
    "vcg_conv_pre_ok": (_I, [_I32P]),
    "vcg_conv_fwd_in_pre": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _F, _P, _I32P, _P, _Z, _P]),
    "vcg_conv_wgrad_saved": (_I, [_P, _P, _P, _P, _P, _I32P, _P, _Z, _P]),
    "vcg_conv_dgrad_workspace": (_Z, [_I32P]),
    "vcg_conv_dgrad": (_I, [_P, _P, _P, _I32P, _P, _Z, _P]),
    "vcg_conv_wgrad_workspace": (_Z, [_I32P]),
    "vcg_conv_wgrad": (_I, [_P, _P, _P, _P, _I32P, _P, _Z, _P]),
    "vcg_in_workspace": (_Z, [_I, _I, _I]),
    "vcg_in_stats": (_I, [_P, _P, _P, _I, _I, _I, _F, _P, _Z, _P]),
    "vcg_in_apply": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vcg_in_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "vcg_act_bwd": (_I, [_P, _P, _P, _Z, _I, _P]),
    "vcg_pixel_shuffle": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "vcg_chan_split": (_I, [_P, _P, _P, _Z, _I, _I, _P]),
    "vcg_chan_cat": (_I, [_P, _P, _P, _Z, _I, _I, _P]),
    "vcg_add_into": (_I, [_P, _P, _Z, _P]),
    "vcg_reparam_fwd": (_I, [_P, _P, _P, _P, _P, _P, _Z, _U64, _U64, _P]),
    "vcg_reparam_bwd": (_I, [_P, _P, _P, _P, _P, _P, _Z, _P]),
    "vcg_randn": (_I, [_P, _Z, _U64, _U64, _P]),
    "vcg_rand_uniform": (_I, [_P, _Z, _U64, _U64, _P]),
    "vcg_reduce_workspace": (_Z, [_Z]),
    "vcg_l1_fwd": (_I, [_P, _P, _P, _Z, _Z, _P, _Z, _P]),
    "vcg_l1_bwd": (_I, [_P, _P, _P, _P, _P, _Z, _Z, _P]),
    "vcg_mse_const_fwd": (_I, [_P, _F, _P, _Z, _P]),
    "vcg_mse_const_bwd": (_I, [_P, _F, _P, _P, _Z, _P]),
    "vcg_kl_fwd": (_I, [_P, _P, _P, _Z, _P, _Z, _P]),
    "vcg_kl_bwd": (_I, [_P, _P, _P, _P, _P, _Z, _P]),
    "vcg_lincomb_fwd": (_I, [_c.POINTER(_P), _c.POINTER(_F), _I, _P, _P]),
    "vcg_sn_prepare": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "vcg_fullmap_fwd": (_I, [_P, _P, _P, _P, _I, _Z, _P]),
    "vcg_fullmap_dgrad": (_I, [_P, _P, _P, _I, _Z, _P]),
    "vcg_fullmap_wgrad": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "vcg_input_resample": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "vcg_input_prejitter": (_I, [_P, _P, _P, _P, _P, _I, _P]),
    "vcg_input_color_jitter": (_I, [_P, _P, _I, _I, _P]),
    "vcg_adam_step": (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _F, _F, _F, _P]),
}

_lib = None
ABI_VERSION = 6


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> libvcg.so next to this file (cross-compiles without a GPU)."""
    if os.environ.get("VCG_LIBVCG"):
        return LIB_PATH                              # a variant build made by hand: never rebuilt here
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "vcg_common.h"), HEADER]
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
            return LIB_PATH
    # -fno-slp-vectorize: hipcc's SLP pass turns neighbouring fp32 adds / subtractions (the operand split) into v_pk_add_f32,
    # and packed-fp32 VALU issues badly beside another wave's MFMA stream (MI355X_MICROARCH.md, "price of one filler beside
    # MFMAs"): measured on the ping-pong GEMM, the staging phase of the waves that run beside their partners' matrix phase
    # took 3500 shader clocks per K-step with the packed ops and 1900 without (profiles/r02_gemm_pp_stamps.txt)
    # one hipcc per source, in parallel (no -fgpu-rdc: the translation units share no device symbol), objects under csrc/_obj
    # (git-ignored), then one link into a temporary name that is renamed over libvcg.so: a process that is loading the old
    # library, or another rank that finds the file while it is being linked, never maps a half-written one
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC"]
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, "vcg_common.h")), os.path.getmtime(HEADER))

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time):
            return obj, None
        cmd = [hipcc_path()] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        return obj, (res.stdout + res.stderr if res.returncode != 0 else None)

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as pool:
        results = list(pool.map(compile_one, srcs))
    bad = [err for _, err in results if err]
    if bad:
        raise RuntimeError("hipcc failed:\n" + "\n".join(bad))
    tmp = LIB_PATH + f".tmp{os.getpid()}"
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + [o for o, _ in results]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


def lib():
    """The loaded library, with argtypes set.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "This package has no fallback path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.vcg_abi_version() != ABI_VERSION:
            raise RuntimeError("libvcg.so ABI version mismatch; rebuild it")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().vcg_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
