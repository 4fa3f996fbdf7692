"""Every conv layer of the BASELINE.json configurations that run on one MI355X — configs[3] (CycleVAEGAN, batch 8, 3x256x256,
latent 64: the headline), configs[1] (`ae`, batch 16) and configs[2] (`vae`, latent 1024, batch 16) — at its FULL size,
through the C ABI.

The CPU oracle cannot run these sizes in test time, so the three directions of each layer are pinned by properties
that do not depend on the size (the shapes below take the Winograd, split-operand 128x128 / 128x64, stream-K,
kw-folded thin and swapped-role paths that the small oracle cases in test_gpu_parity.py do not reach):

  forward          y at 512 random output positions against the convolution sum written out in float64
                   (reflect padding Networks.py:60, PixelUnshuffle channel order Networks.py:86)
  data gradient    the adjoint identity  <conv(x) - bias, g> = <x, dx>   (conv is linear in x)
  weight gradient  64 random elements of dw against their float64 sums over all N*Ho*Wo positions, the bilinear
                   identity <conv(x) - bias, g> = <w, dw>, and db against the float64 column sums of g

torch on the device is the checker here (index gathers and float64 sums), never the thing under test.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

S = 256
SEED = 20261003
B = 8            # the D-block and loss-reduction tests below run at the headline batch

# (name, logical conv cin, cout, k, stride, pad, ups, physical input channels, input H = W)
LAYERS = [
    ("encoder stem CaSb(3,64,k7)", 3, 64, 7, 1, 3, 1, 3, S),
    ("D1 64->128", 256, 128, 3, 1, 1, 2, 64, S),
    ("D2 128->256", 512, 256, 3, 1, 1, 2, 128, S // 2),
    ("D3 256->512", 1024, 512, 3, 1, 1, 2, 256, S // 4),
    ("D4 512->1024", 2048, 1024, 3, 1, 1, 2, 512, S // 8),
    ("R 1024", 1024, 1024, 3, 1, 1, 1, 1024, S // 16),
    ("mu / logvar conv 1024->64", 1024, 64, 3, 1, 1, 1, 1024, S // 16),
    ("logvar conv 64->64", 64, 64, 3, 1, 1, 1, 64, S // 16),
    ("latent 64->1024", 64, 1024, 3, 1, 1, 1, 64, S // 16),
    ("U1 conv 256->512", 256, 512, 3, 1, 1, 1, 256, S // 8),
    ("U2 conv 128->256", 128, 256, 3, 1, 1, 1, 128, S // 4),
    ("U3 conv 64->128", 64, 128, 3, 1, 1, 1, 64, S // 2),
    ("U4 conv 32->64", 32, 64, 3, 1, 1, 1, 32, S),
    ("decoder head CaSb(64,3,k7)", 64, 3, 7, 1, 3, 1, 64, S),
    ("discriminator 3->64 k4 s2", 3, 64, 4, 2, 1, 1, 3, S),
    ("discriminator 64->128 k4 s2", 64, 128, 4, 2, 1, 1, 64, S // 2),
    ("discriminator 128->256 k4 s2", 128, 256, 4, 2, 1, 1, 128, S // 4),
    ("discriminator 256->512 k4 s2", 256, 512, 4, 2, 1, 1, 256, S // 8),
]

# BASELINE.json configs[1] (`ae`, batch 16) and configs[2] (`vae` latent 1024, batch 16) run the same layers at twice the
# pixel count — M doubles, so gemm_plan / wgrad_plan pick other tiles, split factors and stream-K runs — and latent 1024
# adds the bare 1024 -> 1024 bottleneck convs (mu, logvar x2, latent -> 1024; Networks.py:214-237).  No discriminator there.
AE_LAYERS = [l for l in LAYERS if not l[0].startswith(("discriminator", "mu /", "logvar", "latent"))]
VAE1024_ONLY = [("bare 1024->1024 (latent 1024: mu / logvar / latent->1024)", 1024, 1024, 3, 1, 1, 1, 1024, S // 16)]
CASES = ([(8, l) for l in LAYERS] + [(16, l) for l in AE_LAYERS] + [(16, l) for l in VAE1024_ONLY])

Y_TOL = 2e-6        # rel. L2 over the sampled outputs: fp32 rounding of a K <= 18432 sum is ~3e-7
DW_TOL = 4e-6       # sampled dw elements: sums over up to 524288 positions
DOT_TOL = 2e-5      # inner-product identities, relative to ||a|| ||b|| / sqrt(n) (the size of a random inner product)


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


def _scale(a, b):
    return (a.double().norm() * b.double().norm()).item() / a.numel() ** 0.5


@pytest.mark.parametrize("B,layer", CASES, ids=[f"B{b} {l[0]}" for b, l in CASES])
def test_conv_layer_at_headline_size(B, layer, pkg, device):
    name, cin, cout, k, stride, pad, ups, cphys, h = layer
    ops = pkg.ops
    gen = torch.Generator(device="cpu").manual_seed(SEED)
    lid = (LAYERS + VAE1024_ONLY).index(layer) + (100 if B != 8 else 0)
    x = ops.randn((B, cphys, h, h), device, seed=SEED, offset=lid << 32)               # NCHW, the reference's layout
    w = ops.randn((cout, cin, k, k), device, seed=SEED + 1, offset=lid << 32) * (2.0 / (k * k * cout)) ** 0.5
    b = ops.randn((cout,), device, seed=SEED + 2, offset=lid << 32) * 0.1
    spec = ops.ConvSpec(cin, cout, k, stride, pad, True, ups)
    xd = ops.to_nhwc(x).requires_grad_(True)
    wd, bd = torch.nn.Parameter(w.clone()), torch.nn.Parameter(b.clone())
    y = ops.conv_block(xd, wd, bd, spec)
    ho, wo = y.shape[2], y.shape[3]
    g = ops.randn((B, cout, ho, wo), device, seed=SEED + 3, offset=lid << 32)
    y.backward(ops.to_nhwc(g))
    torch.cuda.synchronize()
    yn = ops.to_nchw_contiguous(y.detach())
    dx = ops.to_nchw_contiguous(xd.grad)
    assert yn.shape == (B, cout, ho, wo) and dx.shape == x.shape
    assert torch.isfinite(yn).all() and torch.isfinite(dx).all() and torch.isfinite(wd.grad).all()

    # the logical, padded input of the torch conv the layer stands for
    xl = F.pixel_unshuffle(x, 2) if ups == 2 else x
    xpad = F.pad(xl, (pad, pad, pad, pad), mode="reflect")
    assert (xpad.shape[2] - k) // stride + 1 == ho

    # forward: sampled outputs in float64
    ns = 512
    sn = torch.randint(0, B, (ns,), generator=gen).to(device)
    sco = torch.randint(0, cout, (ns,), generator=gen).to(device)
    soh = torch.randint(0, ho, (ns,), generator=gen).to(device)
    sow = torch.randint(0, wo, (ns,), generator=gen).to(device)
    # always include the four corners and an edge (reflect padding) of the first image
    for i, (a, c) in enumerate([(0, 0), (0, wo - 1), (ho - 1, 0), (ho - 1, wo - 1), (0, wo // 2), (ho // 2, 0)]):
        sn[i], soh[i], sow[i] = 0, a, c
    ar = torch.arange(k, device=device)
    ih = (soh * stride)[:, None, None, None] + ar[None, None, :, None]
    iw = (sow * stride)[:, None, None, None] + ar[None, None, None, :]
    patch = xpad[sn[:, None, None, None], torch.arange(cin, device=device)[None, :, None, None], ih, iw].double()
    ref = (patch * w[sco].double()).sum((1, 2, 3)) + b[sco].double()
    got = yn[sn, sco, soh, sow].double()
    err = ((got - ref).norm() / ref.norm()).item()
    assert err <= Y_TOL, f"{name}: forward differs from the float64 convolution sum by {err:.2e} (rel. L2 over {ns} outputs)"
    worst = ((got - ref).abs().max() / ref.abs().max()).item()
    assert worst <= 10 * Y_TOL, f"{name}: one sampled output is off by {worst:.2e} of the largest"

    # data gradient: adjoint identity
    y0 = yn - b[None, :, None, None]
    lhs, rhs = _dot(y0, g), _dot(x, dx)
    assert abs(lhs - rhs) <= DOT_TOL * _scale(y0, g), \
        f"{name}: <conv x, g> = {lhs:.9e} but <x, dx> = {rhs:.9e} (scale {_scale(y0, g):.3e})"

    # weight gradient: bilinear identity, sampled elements, bias column sums
    dw = wd.grad
    rhs = _dot(w, dw)
    assert abs(lhs - rhs) <= DOT_TOL * _scale(y0, g), \
        f"{name}: <conv x, g> = {lhs:.9e} but <w, dw> = {rhs:.9e} (scale {_scale(y0, g):.3e})"
    nw = 64
    wco = torch.randint(0, cout, (nw,), generator=gen).tolist()
    wc = torch.randint(0, cin, (nw,), generator=gen).tolist()
    wkh = torch.randint(0, k, (nw,), generator=gen).tolist()
    wkw = torch.randint(0, k, (nw,), generator=gen).tolist()
    wref = torch.empty(nw, dtype=torch.float64, device=device)
    for i in range(nw):
        win = xpad[:, wc[i], wkh[i]:wkh[i] + (ho - 1) * stride + 1:stride, wkw[i]:wkw[i] + (wo - 1) * stride + 1:stride]
        wref[i] = (win.double() * g[:, wco[i]].double()).sum()
    wgot = dw[wco, wc, wkh, wkw].double()
    err = ((wgot - wref).norm() / wref.norm()).item()
    assert err <= DW_TOL, f"{name}: weight gradient differs from the float64 sums by {err:.2e} (rel. L2 over {nw} elements)"
    dbref = g.double().sum((0, 2, 3))
    err = ((bd.grad.double() - dbref).norm() / dbref.norm()).item()
    assert err <= DW_TOL, f"{name}: bias gradient differs from the float64 column sums by {err:.2e}"


# layers whose output goes into an InstanceNorm (Networks.py:93-95, 110-115, 128-130, 244-247): epilogue activation as in the model
# (ReLU before the norm in D / U / R.conv1, none in CaSb — the stem and the discriminators' normalised layers)
NORM_LAYERS = [(l, act) for l, act in [
    (LAYERS[0], 0), (LAYERS[1], 1), (LAYERS[2], 1), (LAYERS[3], 1), (LAYERS[4], 1), (LAYERS[5], 1), (LAYERS[5], 0),
    (LAYERS[9], 1), (LAYERS[10], 1), (LAYERS[11], 1), (LAYERS[12], 1), (LAYERS[15], 0), (LAYERS[16], 0), (LAYERS[17], 0)]]
NORM_CASES = [(8, l, a) for l, a in NORM_LAYERS] + [(16, l, a) for l, a in NORM_LAYERS if not l[0].startswith("discriminator")]


@pytest.mark.parametrize("B,layer,act", NORM_CASES, ids=[f"B{b} {l[0]} act{a}" for b, l, a in NORM_CASES])
def test_conv_fwd_in_statistics_at_headline_size(B, layer, act, pkg, device):
    """vcg_conv_fwd_in at the full size of every normalised layer (VERDICT r2 weak #1c): whichever kernel leaves the InstanceNorm
    partials — the Winograd output transform, the 128-row tiles of the direct split-operand kernel, the LDS-slab pixel blocks,
    the separate pass — mean and rstd must be those of the y it wrote, to a float64 reduction's accuracy.  (Round 3: the direct
    and slab tile epilogues summed 64 squares in fp32 before going to double; on channels whose mean dwarfs their spread that
    left rstd ~1e-5 off — this test is the one that would have said so.)"""
    import ctypes
    name, cin, cout, k, stride, pad, ups, cphys, h = layer
    ops, lib, nat = pkg.ops, pkg._native.lib(), pkg._native
    lid = LAYERS.index(layer) + (200 if B != 8 else 0) + 1000 * act
    x = ops.randn((B, cphys, h, h), device, seed=SEED + 7, offset=lid << 32)
    x = x + 0.75                                                                     # post-ReLU-like inputs: a mean next to the spread
    w = torch.nn.Parameter(ops.randn((cout, cin, k, k), device, seed=SEED + 8, offset=lid << 32) * (2.0 / (k * k * cin)) ** 0.5)
    b = ops.randn((cout,), device, seed=SEED + 9, offset=lid << 32) * 2.0            # channels whose mean dwarfs their spread
    spec = ops.ConvSpec(cin, cout, k, stride, pad, True, ups, act)
    xp = ops.as_phys(ops.to_nhwc(x))
    n, hh, ww = B, h, h
    cd = spec.desc(n, hh, ww)
    ho, wo = spec.out_hw(hh, ww)
    c = spec.cout_pitch
    wf = spec.packed(w)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    y = torch.full((n, ho, wo, c), float("nan"), dtype=torch.float32, device=device)
    mean = torch.full((n, c), float("nan"), dtype=torch.float32, device=device)
    rstd = torch.full((n, c), float("nan"), dtype=torch.float32, device=device)
    ws = ops.workspace(lib.vcg_conv_fwd_in_workspace(cd), device)
    nat.check(lib.vcg_conv_fwd_in(P(xp), P(wf), P(b), P(y), P(mean), P(rstd), ops.IN_EPS, None, cd, P(ws), ws.numel() * 4, st), "vcg_conv_fwd_in")
    torch.cuda.synchronize()
    m64 = torch.empty((n, c), dtype=torch.float64, device=device)
    v64 = torch.empty((n, c), dtype=torch.float64, device=device)
    for i in range(n):                                                               # one image at a time: float64 copies of 134 MB maps
        yd = y[i].double().reshape(ho * wo, c)
        m64[i] = yd.mean(0)
        v64[i] = yd.var(0, unbiased=False)
    r64 = 1.0 / torch.sqrt(v64 + ops.IN_EPS)
    scale = v64.sqrt() + m64.abs() + 1e-6
    assert torch.isfinite(mean).all() and torch.isfinite(rstd).all()
    em = ((mean.double() - m64).abs() / scale).max().item()
    er = ((rstd.double() - r64).abs() / r64).max().item()
    assert em <= 1e-6, f"{name}: mean off by {em:.2e} of (std + |mean|)"
    assert er <= 3e-6, f"{name}: rstd off by {er:.2e}"


def test_instance_norm_block_at_headline_size(pkg, device):
    """D(64,128) on an 8x64x256x256 activation: the block's output is ReLU'd THEN normalised (Networks.py:83-96), so every
    (image, channel) plane has mean 0 and biased variance var/(var + eps); and the block matches the same three torch
    calls run by torch on the device in fp32."""
    torch.manual_seed(0)
    mod = pkg.Networks.D(64, 128).to(device)
    x = pkg.ops.randn((B, 64, S, S), device, seed=SEED + 9)
    y = pkg.ops.to_nchw_contiguous(mod(x))
    m = y.double().mean((2, 3))
    v = y.double().var((2, 3), unbiased=False)
    assert m.abs().max().item() < 1e-5
    assert (v - 1).abs().max().item() < 1e-3                     # eps / var is ~1e-5 here
    t = F.relu(F.conv2d(F.pad(F.pixel_unshuffle(x, 2), (1, 1, 1, 1), mode="reflect"), mod.conv.weight, mod.conv.bias))
    ref = F.instance_norm(t, eps=1e-5)
    err = ((y.double() - ref.double()).norm() / ref.double().norm()).item()
    assert err < 1e-4, f"D block at full size differs from torch's conv + relu + instance_norm by {err:.2e}"


def test_l1_and_kl_reductions_at_headline_size(pkg, device):
    """The loss reductions over a full batch (8x3x256x256 and 8x64x16x16) against float64 sums (Losses.py:14-24, 105-121)."""
    ops = pkg.ops
    a = ops.to_nhwc(ops.rand_uniform((B, 3, S, S), device, seed=SEED + 4)).requires_grad_(True)
    bb = ops.to_nhwc(ops.rand_uniform((B, 3, S, S), device, seed=SEED + 5))
    loss = ops.l1_loss(a, bb)
    loss.backward()
    an, bn = ops.to_nchw_contiguous(a.detach()), ops.to_nchw_contiguous(bb)
    ref = (an.double() - bn.double()).abs().mean().item()
    assert abs(loss.item() - ref) <= 1e-6 * ref
    gref = torch.sign(an - bn) / an.numel()
    assert torch.allclose(ops.to_nchw_contiguous(a.grad), gref, rtol=1e-6, atol=0)
    mu = ops.to_nhwc(ops.randn((B, 64, S // 16, S // 16), device, seed=SEED + 6)).requires_grad_(True)
    lv = ops.to_nhwc(ops.randn((B, 64, S // 16, S // 16), device, seed=SEED + 7) * 6).requires_grad_(True)   # some beyond the +-10 clamp
    kl = ops.kl_loss(mu, lv)
    mun, lvn = ops.to_nchw_contiguous(mu.detach()).double(), ops.to_nchw_contiguous(lv.detach()).double().clamp(-10, 10)
    ref = (-0.5 * (1 + lvn - mun ** 2 - lvn.exp()).mean()).item()
    assert abs(kl.item() - ref) <= 2e-6 * abs(ref)
