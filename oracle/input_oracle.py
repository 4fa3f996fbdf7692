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
  * ColorJitter follows torchvision's tensor-path formulas (`_blend(img1, img2, r) = clamp(r * img1 + (1 - r) * img2, 0, 1)`,
    grayscale = 0.299 R + 0.587 G + 0.114 B, contrast against the mean grey of the current image, hue as an HSV rotation),
    the four ops in the drawn order.
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


def resample(src, box, out_size, flip_h=False, flip_v=False, filt=0):
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
    return out / unit


def _gray(img):
    return 0.299 * img[..., 0] + 0.587 * img[..., 1] + 0.114 * img[..., 2]


def _blend(a, b, r):
    return np.clip(r * a + (1.0 - r) * b, 0.0, 1.0)


def _rgb_to_hsv(img):
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    maxc = img.max(-1)
    minc = img.min(-1)
    eqc = maxc == minc
    cr = maxc - minc
    ones = np.ones_like(maxc)
    s = cr / np.where(eqc, ones, maxc)
    crd = np.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = np.mod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return np.stack([h, s, maxc], -1)


def _hsv_to_rgb(img):
    h, s, v = img[..., 0], img[..., 1], img[..., 2]
    i = np.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.astype(np.int64) % 6
    p = np.clip(v * (1.0 - s), 0.0, 1.0)
    q = np.clip(v * (1.0 - f * s), 0.0, 1.0)
    t = np.clip(v * (1.0 - (1.0 - f) * s), 0.0, 1.0)
    sel = [np.stack(c, -1) for c in ((v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q))]
    out = np.zeros_like(img)
    for k in range(6):
        out = np.where((i == k)[..., None], sel[k], out)
    return out


def color_jitter(img, brightness, contrast, saturation, hue, order):
    """img (S, S, 3) float in [0, 1]; factors as drawn by torchvision's ColorJitter.get_params; `order` a permutation of
    (0 brightness, 1 contrast, 2 saturation, 3 hue).  torchvision/transforms/_functional_tensor.py formulas."""
    img = np.clip(np.asarray(img, dtype=np.float64), 0.0, 1.0)      # a uint8 PIL image: bicubic overshoot was clipped
    for op in order:
        if op == 0:
            img = _blend(img, np.zeros_like(img), brightness)
        elif op == 1:
            img = _blend(img, np.full_like(img, _gray(img).mean()), contrast)
        elif op == 2:
            img = _blend(img, np.repeat(_gray(img)[..., None], 3, -1), saturation)
        elif op == 3:
            hsv = _rgb_to_hsv(img)
            hsv[..., 0] = np.mod(hsv[..., 0] + hue, 1.0)
            img = _hsv_to_rgb(hsv)
    return img


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
