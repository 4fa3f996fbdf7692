"""CPU restatement (numpy, float64) of the reference's input transforms — TEST INFRASTRUCTURE ONLY.

Only `tests/` may import this module: it is the checker for the device-side input pipeline
(`vae-cyclegan-implementation_amd/input_pipeline.py`, `csrc/input.hip`), never part of the product path.

What it restates: the torchvision pipelines the reference builds at `/root/reference/train.py:309-319` (summer2winter;
the maps and hypersim builders at :184-190 and :248-262 use subsets of the same ops):

    RandomHorizontalFlip(p) [RandomVerticalFlip(p)] -> RandomResizedCrop(S, scale=(0.33, 1), ratio=(1, 1), BICUBIC)
    -> [ColorJitter(brightness, contrast, saturation, hue)] -> ToTensor          and, for test data, Resize((S, S)) -> ToTensor

torchvision (pinned `torchvision>=0.15.0`, requirements.txt:2) is a third-party dependency that is ABSENT from this image, so
its published algorithms are restated here and anchored on what IS present: Pillow, whose `Image.resize` those transforms
call for PIL inputs (tests/test_input_pipeline.py compares `resample` with `PIL.Image.crop().resize()` — agreement within
the uint8 rounding PIL applies after each of its two passes).

  * resampling: Pillow's separable convolution resize (ImagingResample): for output index o, centre = box0 + (o + 0.5) * scale
    with scale = box_len / S; the filter is stretched by max(scale, 1) (antialiasing when shrinking); taps xmin =
    int(centre - support + 0.5) .. xmax = int(centre + support + 0.5), clipped to the CROP (Pillow crops first); weights
    filter((x - centre + 0.5) / filterscale), normalised to sum 1.  BICUBIC: Keys a = -0.5, support 2; BILINEAR: triangle,
    support 1.  Pillow rounds to uint8 after the horizontal and after the vertical pass; this restatement (and the device
    kernel) keep floating point throughout and scale by 1/255 at the end (ToTensor).
  * flips are applied BEFORE the crop box is drawn (Compose order), i.e. the box lives in flipped-image coordinates.
  * ColorJitter runs, as in the reference, on the uint8 PIL image (torchvision's _functional_pil path): ImageEnhance
    Brightness / Contrast / Color = Image.blend against black / the rounded mean grey / the L image, truncated to uint8 after
    every op; hue = a wrapping integer shift of H in Pillow's uint8 HSV.  Restated from Pillow's C and pinned on Pillow.
  * Everything the reference hands to ToTensor is a uint8 image: `quantize_u8` puts the resampled crop on that grid.
"""
import numpy as np


def _bicubic(x, a=-0.5):
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0,
                    np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))


def _triangle(x):
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


FILTERS = {0: (_bicubic, 2.0), 1: (_triangle, 1.0)}       # 0 = BICUBIC, 1 = BILINEAR


def resample_weights(box0, box_len, out_len, filt):
    """[(first tap, weights)] per output index, taps in (flipped-)image coordinates, clipped to the crop box."""
    fn, support = FILTERS[filt]
    scale = box_len / out_len
    fscale = max(scale, 1.0)
    sup = support * fscale
    out = []
    for o in range(out_len):
        center = (o + 0.5) * scale                       # in crop coordinates (Pillow crops first)
        xmin = max(int(center - sup + 0.5), 0)
        xmax = min(int(center + sup + 0.5), box_len)
        xs = np.arange(xmin, xmax)
        w = fn((xs - center + 0.5) / fscale)
        s = w.sum()
        w = w / s if s != 0 else w
        out.append((box0 + xmin, w))
    return out


def resample(src, box, out_size, flip_h=False, flip_v=False, filt=0, quantize=False):
    """src: (H, W, 3) uint8 — or a float image already in [0, 1] (hypersim's colour modality, jittered on the whole frame before
    the crop: Data_Manager.py:164-171).  box = (y0, x0, h, w) in FLIPPED-image coordinates.  Returns (S, S, 3) float64 in [0, 1]
    (not clamped: bicubic overshoot stays, as ToTensor of a float image would keep it; Pillow's uint8 clips it)."""
    unit = 255.0 if src.dtype == np.uint8 else 1.0
    img = src.astype(np.float64)
    if flip_h:
        img = img[:, ::-1]
    if flip_v:
        img = img[::-1]
    y0, x0, h, w = box
    S = out_size
    wx = resample_weights(x0, w, S, filt)
    wy = resample_weights(y0, h, S, filt)
    tmp = np.zeros((img.shape[0], S, 3))
    for o, (first, wts) in enumerate(wx):
        tmp[:, o] = np.tensordot(img[:, first:first + len(wts)], wts, axes=([1], [0]))
    out = np.zeros((S, S, 3))
    for o, (first, wts) in enumerate(wy):
        out[o] = np.tensordot(tmp[first:first + len(wts)], wts, axes=([0], [0]))
    if quantize:                                             # the uint8 PIL image the reference's resize returns, then ToTensor
        return quantize_u8(out / unit).astype(np.float64) / 255.0
    return out / unit


def quantize_u8(img01):
    """What the reference's tensors actually hold: its transforms run on uint8 PIL images (train.py:309-319), so the resized
    crop is rounded and clipped to 0..255 (Pillow: clip8(value + 0.5)) before ColorJitter / ToTensor see it."""
    return np.clip(np.floor(np.asarray(img01, dtype=np.float64) * 255.0 + 0.5), 0, 255).astype(np.uint8)


# ---- ColorJitter as the reference runs it: on PIL images (train.py:316, Data_Manager.py:164-171), i.e. torchvision's
# _functional_pil path — brightness / contrast / saturation are PIL.ImageEnhance (Image.blend with a degenerate image, uint8
# after every op), hue is an integer shift of the H channel of Pillow's uint8 HSV conversion.  Restated here from Pillow's
# published C (libImaging/Blend.c, Convert.c) and pinned on the installed Pillow in tests/test_input_pipeline.py
# (exhaustively for the two colour-space conversions).
def _to_L(rgb_u8):
    """Pillow RGB -> L: (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16"""
    a = rgb_u8.astype(np.int64)
    return ((a[..., 0] * 19595 + a[..., 1] * 38470 + a[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def _pil_blend(degenerate_u8, img_u8, alpha):
    """Image.blend(degenerate, image, alpha) as ImageEnhance calls it: in C float,  t = in1 + alpha * (in2 - in1),  truncated to
    uint8; outside 0 <= alpha <= 1 clipped to 0..255 first (Blend.c)."""
    alpha = np.float32(alpha)
    d = degenerate_u8.astype(np.float32)
    t = (d + alpha * (img_u8.astype(np.float32) - d)).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, np.trunc(t))).astype(np.uint8)


def pil_rgb_to_hsv(rgb_u8):
    """Pillow Convert.c rgb2hsv_row (float intermediates where the C has float, double where a double literal promotes)."""
    a = rgb_u8
    r, g, b = (a[..., k].astype(np.float32) for k in range(3))
    maxc, minc = a.max(-1), a.min(-1)
    eq = maxc == minc
    cr = (maxc.astype(np.int32) - minc).astype(np.float32)
    crd = np.where(eq, np.float32(1), cr)
    mx = maxc.astype(np.float32)
    s = cr / np.where(eq, np.float32(1), mx)
    rc, gc, bc = (mx - r) / crd, (mx - g) / crd, (mx - b) / crd
    d = np.float64
    h = np.where(a[..., 0] == maxc, (bc - gc).astype(d),
                 np.where(a[..., 1] == maxc, 2.0 + rc.astype(d) - bc.astype(d), 4.0 + gc.astype(d) - rc.astype(d))).astype(np.float32)
    h = np.fmod(h.astype(d) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(d) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(d) * 255.0).astype(np.int64), 0, 255)
    return np.stack([np.where(eq, 0, uh), np.where(eq, 0, us), maxc], -1).astype(np.uint8)


def pil_hsv_to_rgb(hsv_u8):
    """Pillow Convert.c hsv2rgb"""
    d = np.float64
    h, s, v = hsv_u8[..., 0], hsv_u8[..., 1], hsv_u8[..., 2]
    h6 = h.astype(d) * 6.0 / 255.0
    i = np.floor(h6).astype(np.int64)
    f = (h6 - i.astype(np.float32).astype(d)).astype(np.float32)
    fs = (s.astype(d) / 255.0).astype(np.float32)
    vf = v.astype(d)

    def rnd(x):                                              # C round(): half away from zero (all values >= 0 here)
        return np.clip(np.floor(x + 0.5).astype(np.int64), 0, 255)
    p = rnd(vf * (1.0 - fs.astype(d)))
    q = rnd(vf * (1.0 - (fs * f).astype(np.float32).astype(d)))
    t = rnd(vf * (1.0 - fs.astype(d) * (1.0 - f.astype(d))))
    vv = v.astype(np.int64)
    sel = ((vv, t, p), (q, vv, p), (p, vv, t), (p, q, vv), (t, p, vv), (vv, p, q))
    out = np.zeros(hsv_u8.shape, np.int64)
    for k in range(6):
        m = (i % 6) == k
        for c in range(3):
            out[..., c] = np.where(m, sel[k][c], out[..., c])
    for c in range(3):
        out[..., c] = np.where(s == 0, vv, out[..., c])
    return out.astype(np.uint8)


def hue_shift_u8(hue):
    """torchvision _functional_pil.adjust_hue: np_h += np.uint8(hue_factor * 255), wrapping"""
    return int(hue * 255.0) & 0xFF                           # int(): truncation toward zero, as the C cast


def color_jitter_pil(img_u8, brightness, contrast, saturation, hue, order):
    """torchvision ColorJitter.forward on a PIL image (uint8 RGB): the four ops in the drawn `order`
    (0 brightness, 1 contrast, 2 saturation, 3 hue), each returning a uint8 image."""
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    for op in order:
        if op == 0:                                          # ImageEnhance.Brightness: degenerate = black
            img = _pil_blend(np.zeros_like(img), img, brightness)
        elif op == 1:                                        # ImageEnhance.Contrast: degenerate = int(mean(L) + 0.5) everywhere
            L = _to_L(img)
            mean = int(float(L.astype(np.int64).sum()) / L.size + 0.5)
            img = _pil_blend(np.full_like(img, mean), img, contrast)
        elif op == 2:                                        # ImageEnhance.Color: degenerate = the L image
            img = _pil_blend(np.repeat(_to_L(img)[..., None], 3, -1), img, saturation)
        elif op == 3:
            hsv = pil_rgb_to_hsv(img)
            hsv[..., 0] = (hsv[..., 0].astype(np.int64) + hue_shift_u8(hue)).astype(np.uint8)
            img = pil_hsv_to_rgb(hsv)
    return img


def color_jitter(img01, brightness, contrast, saturation, hue, order):
    """The reference's ColorJitter on an image given as floats in [0, 1]: quantised to the uint8 PIL image the reference
    jitters, jittered on that grid, and returned as ToTensor would (uint8 / 255)."""
    return color_jitter_pil(quantize_u8(img01), brightness, contrast, saturation, hue, order).astype(np.float64) / 255.0


def draw_crop(rng, height, width, scale=(0.33, 1.0)):
    """torchvision RandomResizedCrop.get_params with ratio (1, 1): up to ten attempts at a square of area
    U(scale) x image area, then the centre-crop fallback.  Returns (y0, x0, h, w)."""
    area = height * width
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        w = h = int(round(np.sqrt(target)))
        if 0 < w <= width and 0 < h <= height:
            return int(rng.randint(0, height - h + 1)), int(rng.randint(0, width - w + 1)), h, w
    in_ratio = width / height
    if in_ratio < 1.0:
        w, h = width, int(round(width / 1.0))
    elif in_ratio > 1.0:
        h, w = height, int(round(height * 1.0))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w
