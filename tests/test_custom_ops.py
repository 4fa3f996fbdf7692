"""The `torch.library` surface (vae-cyclegan-implementation_amd/custom_ops.py): torch.ops.vcg.conv_block / conv_block_backward.

CPU part: registration, schemas, shape inference on meta tensors, no CPU path.  GPU part: every block kind of the reference
(/root/reference/Networks.py:60-140: CaSb, D, R, U, S and the discriminator's strided conv) through the custom op against
 (a) plain torch ops in float64 — conv2d over a reflect-padded input, InstanceNorm, activation, residual, PixelShuffle — for the
     result and for dx / dweight / dbias taken by `torch.autograd.grad` (fp32 tolerance, written below), and
 (b) `ops.conv_block`, the training path's autograd.Function over the same kernels: same result, same gradients;
`torch.library.opcheck` on the registration; and a two-block function under `torch.compile(backend="aot_eager")`."""
import importlib

import pytest
import torch
import torch.nn.functional as F

PKG = importlib.import_module("vae-cyclegan-implementation_amd")
ops, cops = PKG.ops, PKG.custom_ops

#        name        cin_phys cout k  stride pad ups epi              norm  post           shuffle residual bias
BLOCKS = [
    ("CaSb",         3,   32, 7, 1, 3, 1, ops.ACT_NONE,    True,  ops.ACT_RELU,    False, False, True),
    ("D",            16,  32, 3, 1, 1, 2, ops.ACT_RELU,    True,  ops.ACT_NONE,    False, False, True),
    ("R.conv2",      32,  32, 3, 1, 1, 1, ops.ACT_NONE,    True,  ops.ACT_NONE,    False, True,  True),
    ("U shuffled",   16,  32, 3, 1, 1, 1, ops.ACT_RELU,    True,  ops.ACT_NONE,    True,  False, True),
    ("S + Tanh",     16,  3,  3, 1, 1, 1, ops.ACT_TANH,    False, ops.ACT_NONE,    False, False, True),
    ("disc k4 s2",   8,   16, 4, 2, 1, 1, ops.ACT_LEAKY,   False, ops.ACT_NONE,    False, False, False),
]


def _act(t, a):
    return {ops.ACT_NONE: lambda v: v, ops.ACT_RELU: F.relu, ops.ACT_LEAKY: lambda v: F.leaky_relu(v, 0.2), ops.ACT_TANH: torch.tanh,
            ops.ACT_SIGMOID: torch.sigmoid}[a](t)


def torch_block(x, w, b, res, stride, pad, ups, epi, norm, post, shuffle):
    if ups == 2:
        x = F.pixel_unshuffle(x, 2)
    t = F.conv2d(F.pad(x, (pad,) * 4, mode="reflect"), w, b, stride=stride)
    t = _act(t, epi)
    if norm:
        t = _act(F.instance_norm(t, eps=1e-5), post)
        if res is not None:
            t = t + res
        if shuffle:
            t = F.pixel_shuffle(t, 2)
    return t


def _inputs(block, device, dtype=torch.float32, seed=0):
    name, cphys, cout, k, stride, pad, ups, epi, norm, post, shuffle, has_res, has_bias = block
    g = torch.Generator().manual_seed(seed)
    cin = cphys * 4 if ups == 2 else cphys
    x = torch.randn(2, cphys, 24, 24, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1 if has_bias else None
    res = torch.randn(2, cout, 24, 24, generator=g) if has_res else None
    mk = lambda t: None if t is None else t.to(device=device, dtype=dtype).requires_grad_(True)   # noqa: E731
    return mk(x), mk(w), mk(b), mk(res)


# ------------------------------------------------------------------ CPU: registration
def test_ops_are_registered_with_schemas():
    s = str(torch.ops.vcg.conv_block.default._schema)
    assert s.startswith("vcg::conv_block(Tensor x, Tensor weight, Tensor? bias, Tensor? residual,") and s.endswith("-> (Tensor, Tensor, Tensor, Tensor)")
    s = str(torch.ops.vcg.conv_block_backward.default._schema)
    assert "bool need_dx, bool need_dw, bool need_db) -> (Tensor, Tensor, Tensor)" in s


@pytest.mark.parametrize("block", BLOCKS, ids=[b[0] for b in BLOCKS])
def test_shape_inference_on_meta_tensors(block):
    name, cphys, cout, k, stride, pad, ups, epi, norm, post, shuffle, has_res, has_bias = block
    x, w, b, res = _inputs(block, "meta")
    out, t, mean, rstd = torch.ops.vcg.conv_block(x, w, b, res, stride, pad, True, ups, epi, norm, post, shuffle)
    want = torch_block(*(None if v is None else torch.empty(v.shape, device="meta") for v in (x, w, b, res)), stride, pad, ups, epi, norm, post, shuffle)
    assert tuple(out.shape) == tuple(want.shape)
    n, c, h, ww = out.shape
    assert out.stride() == ops.nhwc_strides(n, c, h, ww)          # the package's NHWC storage behind a logical NCHW shape
    if norm:
        assert tuple(mean.shape) == tuple(rstd.shape) == (2, ops.pitch(cout)) and t.dim() == 4
    else:
        assert t.numel() == mean.numel() == rstd.numel() == 0
    dx, dw, db = torch.ops.vcg.conv_block_backward(out, x, w, out, t, mean, rstd, stride, pad, True, ups, epi, norm, post, shuffle,
                                                   True, True, has_bias)
    assert tuple(dx.shape) == tuple(x.shape) and tuple(dw.shape) == tuple(w.shape)
    assert tuple(db.shape) == ((cout,) if has_bias else (0,))


def test_there_is_no_cpu_path():
    with pytest.raises(NotImplementedError):
        torch.ops.vcg.conv_block(torch.zeros(1, 4, 8, 8), torch.zeros(4, 4, 3, 3), None, None, 1, 1, True, 1, 0, False, 0, False)


# ------------------------------------------------------------------ GPU
def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.gpu
@pytest.mark.parametrize("block", BLOCKS, ids=[b[0] for b in BLOCKS])
def test_custom_op_matches_torch_float64_and_the_training_path(block, device):
    name, cphys, cout, k, stride, pad, ups, epi, norm, post, shuffle, has_res, has_bias = block
    x, w, b, res = _inputs(block, device)
    out = cops.conv_block(x, w, b, res, stride=stride, pad=pad, reflect=True, ups=ups, epi_act=epi, norm=norm, post_act=post, shuffle=shuffle)
    gy = torch.randn(tuple(out.shape), generator=torch.Generator().manual_seed(7)).to(device)
    wanted = [v for v in (x, w, b, res) if v is not None]
    grads = torch.autograd.grad(out, wanted, gy)
    # (a) plain torch in float64 on the CPU
    x64, w64, b64, r64 = _inputs(block, "cpu", torch.float64)
    ref = torch_block(x64, w64, b64, r64, stride, pad, ups, epi, norm, post, shuffle)
    ref_grads = torch.autograd.grad(ref, [v for v in (x64, w64, b64, r64) if v is not None], gy.cpu().double())
    # fp32 arithmetic against float64: 2e-5 of the tensor's norm on results, 2e-4 on gradients (one ReLU mask flip at a value
    # that is rounding-level zero in fp32 moves a gradient by about its share of one element)
    assert _rel(ops.to_nchw_contiguous(out.detach()), ref) <= 2e-5, name
    names = [n for n, v in zip(("dx", "dweight", "dbias", "dresidual"), (x, w, b, res)) if v is not None]
    for nm, got, want in zip(names, grads, ref_grads):
        got = ops.to_nchw_contiguous(got) if got.dim() == 4 and ops.is_nhwc_view(got) else got
        if nm == "dbias" and norm and epi == ops.ACT_NONE:
            # InstanceNorm directly on conv + bias: the mean subtraction cancels the bias; float64 autograd leaves rounding noise
            assert float(got.abs().max()) == 0.0 and float(want.abs().max()) <= 1e-9 * float(gy.abs().sum())
            continue
        assert _rel(got, want) <= 2e-4, f"{name} {nm}: {_rel(got, want):.2e}"
    # (b) the training path's autograd.Function (gradients of the parameters accumulate into .grad there)
    x2, w2, b2, res2 = _inputs(block, device)
    spec = ops.ConvSpec(w2.shape[1], cout, k, stride, pad, True, ups, epi, norm, post, shuffle)
    x2n = ops.to_nhwc(x2)
    out2 = ops.conv_block(x2n, w2, b2, spec, residual=None if res2 is None else ops.to_nhwc(res2))
    w2.grad = torch.zeros_like(w2)
    if b2 is not None:
        b2.grad = torch.zeros_like(b2)
    out2.backward(gy)
    assert _rel(out.detach(), out2.detach()) <= 1e-6
    assert _rel(grads[0], x2.grad) <= 1e-5 and _rel(grads[1], w2.grad) <= 1e-5
    if b is not None and not (norm and epi == ops.ACT_NONE):
        assert _rel(grads[2], b2.grad) <= 1e-5


@pytest.mark.gpu
def test_opcheck_on_the_registration(device):
    block = BLOCKS[1]
    name, cphys, cout, k, stride, pad, ups, epi, norm, post, shuffle, has_res, has_bias = block
    x, w, b, res = _inputs(block, device)
    args = (x, w, b, res, stride, pad, True, ups, epi, norm, post, shuffle)
    torch.library.opcheck(torch.ops.vcg.conv_block.default, args, test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    out, t, mean, rstd = torch.ops.vcg.conv_block(*[a.detach() if isinstance(a, torch.Tensor) else a for a in args])
    bargs = (torch.randn_like(out), x.detach(), w.detach(), out, t, mean, rstd, stride, pad, True, ups, epi, norm, post, shuffle, True, True, True)
    torch.library.opcheck(torch.ops.vcg.conv_block_backward.default, bargs, test_utils=("test_schema", "test_faketensor"))


@pytest.mark.gpu
def test_two_blocks_under_torch_compile_aot_eager(device):
    """The ops are opaque nodes a tracing compiler can carry: forward and backward of two chained blocks through
    torch.compile(backend="aot_eager") (no code generation — this package has no use for one) equal eager."""
    b1, b2 = BLOCKS[1], BLOCKS[2]
    x, w1, bi1, _ = _inputs(b1, device, seed=1)
    g = torch.Generator().manual_seed(3)
    w2 = (torch.randn(32, 32, 3, 3, generator=g) * 0.06).to(device).requires_grad_(True)

    def f(x, w1, bi1, w2):
        h = cops.conv_block(x, w1, bi1, None, stride=1, pad=1, ups=2, epi_act=ops.ACT_RELU, norm=True)
        return cops.conv_block(h, w2, None, h, stride=1, pad=1, epi_act=ops.ACT_NONE, norm=True)

    eager = f(x, w1, bi1, w2)
    ge = torch.autograd.grad(eager.square().sum(), (x, w1, bi1, w2))
    comp = torch.compile(f, backend="aot_eager", fullgraph=True)(x, w1, bi1, w2)
    gc = torch.autograd.grad(comp.square().sum(), (x, w1, bi1, w2))
    assert torch.equal(eager, comp)
    for a, b in zip(ge, gc):
        assert _rel(a, b) <= 1e-6


@pytest.mark.gpu
def test_the_op_is_a_function_of_the_weight_values(device):
    """No pack is cached across calls: a weight rewritten through `.data` (no version bump — what a hand-rolled optimizer or a
    broadcast does) is seen by the next call, forward and backward."""
    block = BLOCKS[4]
    x, w, b, _ = _inputs(block, device)
    kw = dict(stride=1, pad=1, epi_act=ops.ACT_NONE, norm=False)
    o1 = cops.conv_block(x, w, None, None, **kw)
    (dx1,) = torch.autograd.grad(o1, (x,), torch.ones_like(o1), retain_graph=True)
    ver = w._version
    w.data.mul_(2.0)
    assert w._version == ver
    o2 = cops.conv_block(x, w, None, None, **kw)
    (dx2,) = torch.autograd.grad(o2, (x,), torch.ones_like(o2))
    assert _rel(o2.detach(), 2.0 * o1.detach()) <= 1e-6 and _rel(dx2, 2.0 * dx1) <= 1e-6
    # ... and the FIRST graph's backward, run now, sees the weight as it is now too (it is saved by reference, like any torch op's)
    (dx1_late,) = torch.autograd.grad(o1, (x,), torch.ones_like(o1))
    assert _rel(dx1_late, dx2) <= 1e-6
