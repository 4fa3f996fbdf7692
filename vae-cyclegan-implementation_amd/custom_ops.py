"""The conv block as `torch.library` custom ops: `torch.ops.vcg.conv_block` / `torch.ops.vcg.conv_block_backward`.

SURVEY.md §7 step 3 / §8b (north_star: "PyTorch-ROCm custom ops"; VERDICT r3 missing #5).  `ops.conv_block` — what the model
classes of `Networks.py` call — is a `torch.autograd.Function` with side channels that a training step wants and a functional op
cannot have: weight gradients accumulated straight into the optimizer's flat buffer (autograd sees None), the forward's
Winograd-transformed input kept for the weight gradient, magnitude handles travelling on tensor objects, a second stream.  This
module is the same block WITHOUT them, for every other caller:

  * a functional forward op that returns what its backward needs as ordinary outputs, and a functional backward op that
    RETURNS dx / dweight / dbias — any optimizer, `torch.autograd.grad`, gradient checkers work;
  * registered with the dispatcher (`torch.library.custom_op`), with fake (shape-only) implementations and
    `register_autograd`, so `torch.compile` / `make_fx` trace through a model that uses it as one opaque node per block and
    `torch.library.opcheck` can test the registration (tests/test_custom_ops.py);
  * the same C-ABI entry points underneath (include/vcg.h: vcg_conv_fwd_in_h, vcg_in_apply_h, vcg_in_bwd_h, vcg_act_bwd_h,
    vcg_conv_wgrad_saved_h, vcg_conv_dgrad_h) — no second implementation, and no CPU path: a CPU tensor raises.

The block is the reference's `Conv2d(reflect) [+ ReLU / Sigmoid] [+ InstanceNorm2d [+ ReLU] [+ residual] [+ PixelShuffle]]`
(/root/reference/Networks.py:83-96 CaSb, :98-116 D, :118-131 U, :60-81 R); arguments as `ops.ConvSpec`.
"""
import ctypes
from collections import OrderedDict
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _native, ops

_SPECS = OrderedDict()          # geometry -> ConvSpec (descriptor builder only: nothing about a weight is cached here)
_SPECS_MAX = 512


def _spec_for(weight, stride, pad, reflect, ups, epi_act, norm, post_act, shuffle):
    cout, cin, k, k2 = weight.shape
    if k != k2:
        raise RuntimeError(f"vcg::conv_block: square kernels only, got {k} x {k2}")
    key = (cin, cout, k, stride, pad, reflect, ups, epi_act, norm, post_act, shuffle)
    sp = _SPECS.get(key)
    if sp is None:
        sp = ops.ConvSpec(cin, cout, k, stride, pad, reflect, ups, epi_act, norm, post_act, shuffle)
        _SPECS[key] = sp
        if len(_SPECS) > _SPECS_MAX:
            _SPECS.popitem(last=False)
    else:
        _SPECS.move_to_end(key)
    return sp


def _fresh_pack(spec, weight, geom):
    """The packed form of `weight` (include/vcg.h, vcg_pack_weight), made for THIS call into a buffer of its own.  The training
    path caches packs per parameter and optimizer step (`ConvSpec.packed`, keyed on torch's version counter); a functional op has
    no such contract with its caller — a weight written through `.data`, or a new tensor that landed on a freed one's address,
    looks unchanged to any key short of the contents — so the op pays one pack launch per call instead."""
    spec._packed = spec._packed_key = None
    wf = spec.packed(weight, geom)
    spec._packed = spec._packed_key = None
    return wf


def _geometry(x, weight, stride, pad, ups, norm, shuffle):
    """Shapes of (out, t, mean, rstd) — shared by the real and the fake implementations."""
    n, cphys, h, w = x.shape
    cout, cin, k, _ = weight.shape
    cin_phys = cin // 4 if ups == 2 else cin
    if cphys != cin_phys:
        raise RuntimeError(f"vcg::conv_block: the weight takes {cin} channels ({cin_phys} before the fused un-shuffle), x has {cphys}")
    hl, wl = h // ups, w // ups
    ho, wo = (hl + 2 * pad - k) // stride + 1, (wl + 2 * pad - k) // stride + 1
    cp = ops.pitch(cout)
    if norm and shuffle:
        out = (n, cout // 4, 2 * ho, 2 * wo)
    else:
        out = (n, cout, ho, wo)
    return out, (n, ho, wo, cp), (n, cp)


def _empty_logical(shape, device):
    n, c, h, w = shape
    return ops.logical_of(torch.empty((n, h, w, ops.pitch(c)), dtype=torch.float32, device=device), c)


@torch.library.custom_op("vcg::conv_block", mutates_args=(), device_types="cuda")
def conv_block_op(x: Tensor, weight: Tensor, bias: Optional[Tensor], residual: Optional[Tensor], stride: int, pad: int,
                  reflect: bool, ups: int, epi_act: int, norm: bool, post_act: int,
                  shuffle: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """-> (out, t, mean, rstd).  `out`: the block's result, a logical (N, C, H, W) tensor in the package's NHWC storage
    (`ops.to_nchw_contiguous` converts).  `t` (the conv output the InstanceNorm saw), `mean`, `rstd`: what the backward needs
    — empty when `norm` is false (then `out` itself is what the activation's backward reads)."""
    lib = _native.lib()
    spec = _spec_for(weight, stride, pad, reflect, ups, epi_act, norm, post_act, shuffle)
    _geometry(x, weight, stride, pad, ups, norm, shuffle)
    xp = ops.as_phys(x)
    n, h, w, _ = xp.shape
    ho, wo = spec.out_hw(h, w)
    cd = spec.desc(n, h, w)
    dev = x.device
    wf = _fresh_pack(spec, weight, (n, h, w))
    c = spec.cout_pitch
    t = torch.empty((n, ho, wo, c), dtype=torch.float32, device=dev)
    none = ctypes.c_void_p(0)
    if norm:
        mean = torch.empty((n, c), dtype=torch.float32, device=dev)
        rstd = torch.empty((n, c), dtype=torch.float32, device=dev)
        ws = ops.workspace(lib.vcg_conv_fwd_in_workspace(cd), dev)
        _native.check(lib.vcg_conv_fwd_in_h(ops._ptr(xp), ops._ptr(wf), ops._ptr(bias), ops._ptr(t), ops._ptr(mean), ops._ptr(rstd),
                                            ops.IN_EPS, none, cd, ops._ptr(ws), ws.numel() * 4, 0, ops._stream()), "vcg_conv_fwd_in")
        resp = ops.as_phys(residual) if residual is not None else None
        if shuffle:
            outp = torch.empty((n, 2 * ho, 2 * wo, c // 4), dtype=torch.float32, device=dev)
            cout_log = spec.cout // 4
        else:
            outp = torch.empty((n, ho, wo, c), dtype=torch.float32, device=dev)
            cout_log = spec.cout
        amax = ctypes.c_uint64(0)
        _native.check(lib.vcg_in_apply_h(ops._ptr(t), ops._ptr(mean), ops._ptr(rstd), ops._ptr(resp), ops._ptr(outp), n, ho, wo, c,
                                         post_act, int(shuffle), ctypes.byref(amax), ops._stream()), "vcg_in_apply")
        return ops.logical_of(outp, cout_log), t, mean, rstd
    if residual is not None or shuffle or post_act:
        raise RuntimeError("vcg::conv_block: residual / shuffle / post_act need norm=True")
    ws = ops.workspace(lib.vcg_conv_fwd_workspace(cd), dev)
    _native.check(lib.vcg_conv_fwd_in_h(ops._ptr(xp), ops._ptr(wf), ops._ptr(bias), ops._ptr(t), none, none, ops.IN_EPS, none, cd,
                                        ops._ptr(ws), ws.numel() * 4, 0, ops._stream()), "vcg_conv_fwd_in")
    e = torch.empty(0, dtype=torch.float32, device=dev)
    return ops.logical_of(t, spec.cout), e, torch.empty_like(e), torch.empty_like(e)


@conv_block_op.register_fake
def _(x, weight, bias, residual, stride, pad, reflect, ups, epi_act, norm, post_act, shuffle):
    out, tshape, sshape = _geometry(x, weight, stride, pad, ups, norm, shuffle)
    o = _empty_logical(out, x.device)
    if norm:
        return (o, torch.empty(tshape, dtype=torch.float32, device=x.device), torch.empty(sshape, dtype=torch.float32, device=x.device),
                torch.empty(sshape, dtype=torch.float32, device=x.device))
    e = torch.empty(0, dtype=torch.float32, device=x.device)
    return o, e, torch.empty_like(e), torch.empty_like(e)


@torch.library.custom_op("vcg::conv_block_backward", mutates_args=(), device_types="cuda")
def conv_block_backward_op(g: Tensor, x: Tensor, weight: Tensor, out: Tensor, t: Tensor, mean: Tensor, rstd: Tensor, stride: int,
                           pad: int, reflect: bool, ups: int, epi_act: int, norm: bool, post_act: int, shuffle: bool,
                           need_dx: bool, need_dw: bool, need_db: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (dx, dweight, dbias) of `conv_block` for the output gradient `g` (the residual's gradient is `g` itself); tensors not
    asked for come back empty.  dbias of a block whose InstanceNorm follows the convolution directly (epi_act 0) is exactly
    zero — the mean subtraction cancels the bias — and is returned as zeros."""
    lib = _native.lib()
    spec = _spec_for(weight, stride, pad, reflect, ups, epi_act, norm, post_act, shuffle)
    xp = ops.as_phys(x)
    n, h, w, _ = xp.shape
    ho, wo = spec.out_hw(h, w)
    cd = spec.desc(n, h, w)
    dev = x.device
    c = spec.cout_pitch
    gp = ops.as_phys(g)
    hnd = ctypes.c_uint64(0)
    if norm:
        dt = torch.empty_like(t)
        ws = ops.workspace(lib.vcg_in_workspace(n, ho * wo, c), dev)
        _native.check(lib.vcg_in_bwd_h(ops._ptr(gp), ops._ptr(t), ops._ptr(mean), ops._ptr(rstd), ops._ptr(dt), n, ho, wo, c, epi_act,
                                       post_act, int(shuffle), ops._ptr(ws), ws.numel() * 4, ctypes.byref(hnd), ops._stream()), "vcg_in_bwd")
    elif epi_act != ops.ACT_NONE:
        tp = ops.as_phys(out)
        dt = torch.empty_like(tp)
        _native.check(lib.vcg_act_bwd_h(ops._ptr(gp), ops._ptr(tp), ops._ptr(dt), tp.numel(), epi_act, ctypes.byref(hnd), ops._stream()),
                      "vcg_act_bwd")
    else:
        dt = gp
    e = torch.empty(0, dtype=torch.float32, device=dev)
    dw, db, dx = e, torch.empty_like(e), torch.empty_like(e)
    if need_dw or need_db:
        gw = torch.zeros(weight.shape, dtype=torch.float32, device=dev)
        gb = torch.zeros(spec.cout, dtype=torch.float32, device=dev)
        cancels = norm and epi_act == ops.ACT_NONE
        ws = ops.workspace(lib.vcg_conv_wgrad_workspace(cd), dev)
        _native.check(lib.vcg_conv_wgrad_saved_h(ops._ptr(xp), ops._ptr(dt), ops._ptr(gw), None if cancels else ops._ptr(gb),
                                                 ctypes.c_void_p(0), cd, ops._ptr(ws), ws.numel() * 4, 0, 0, ops._stream()), "vcg_conv_wgrad")
        if need_dw:
            dw = gw
        if need_db:
            db = gb
    if need_dx:
        wf = _fresh_pack(spec, weight, (n, h, w))
        dxp = torch.empty_like(xp)
        ws = ops.workspace(lib.vcg_conv_dgrad_workspace(cd), dev)
        _native.check(lib.vcg_conv_dgrad_h(ops._ptr(dt), ops._ptr(wf), ops._ptr(dxp), cd, ops._ptr(ws), ws.numel() * 4, 0, ops._stream()),
                      "vcg_conv_dgrad")
        dx = ops.logical_of(dxp, spec.cin_phys_log)
    return dx, dw, db


@conv_block_backward_op.register_fake
def _(g, x, weight, out, t, mean, rstd, stride, pad, reflect, ups, epi_act, norm, post_act, shuffle, need_dx, need_dw, need_db):
    def e():
        return torch.empty(0, dtype=torch.float32, device=x.device)
    dx = _empty_logical(tuple(x.shape), x.device) if need_dx else e()
    dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device) if need_dw else e()
    db = torch.empty(weight.shape[0], dtype=torch.float32, device=x.device) if need_db else e()
    return dx, dw, db


def _setup_context(ctx, inputs, output):
    x, weight, bias, residual, stride, pad, reflect, ups, epi_act, norm, post_act, shuffle = inputs
    out, t, mean, rstd = output
    ctx.save_for_backward(x, weight, out, t, mean, rstd)
    ctx.cfg = (stride, pad, reflect, ups, epi_act, norm, post_act, shuffle)
    ctx.has_bias, ctx.has_res = bias is not None, residual is not None


def _backward(ctx, g_out, g_t, g_mean, g_rstd):
    # t / mean / rstd are outputs only so that the backward can be a function of tensors; nothing differentiates through them
    x, weight, out, t, mean, rstd = ctx.saved_tensors
    need = ctx.needs_input_grad
    dx, dw, db = torch.ops.vcg.conv_block_backward(g_out, x, weight, out, t, mean, rstd, *ctx.cfg, need[0], need[1],
                                                   ctx.has_bias and need[2])
    return (dx if need[0] else None, dw if need[1] else None, db if (ctx.has_bias and need[2]) else None,
            g_out if (ctx.has_res and need[3]) else None, None, None, None, None, None, None, None, None)


conv_block_op.register_autograd(_backward, setup_context=_setup_context)


def conv_block(x, weight, bias=None, residual=None, *, stride=1, pad=1, reflect=True, ups=1, epi_act=ops.ACT_NONE, norm=False,
               post_act=ops.ACT_NONE, shuffle=False):
    """The block's result alone (see `torch.ops.vcg.conv_block` for the auxiliary outputs)."""
    return torch.ops.vcg.conv_block(x, weight, bias, residual, stride, pad, reflect, ups, epi_act, norm, post_act, shuffle)[0]
