"""Device-side input pipeline: decoded uint8 images in, augmented fp32 NHWC batches out, off the step's critical path.

Replaces the reference's `Data_Manager.py` datasets + torchvision transform pipelines + `DataLoader`
(/root/reference/train.py:174-357, Data_Manager.py:327-451) for the training loop's input side.
The reference decodes, flips, crops, resamples (bicubic), colour-jitters and converts every sample on the host with PIL;
here the host only DECODES (PIL, a thread pool) and DRAWS the random parameters; the pixels are uploaded as uint8 — a
quarter of the fp32 bytes — into a pinned staging arena, and flips + RandomResizedCrop + bicubic resampling (rounded to the
uint8 grid of the PIL image the reference resizes) + ColorJitter (torchvision's PIL path: ImageEnhance blends and the uint8
HSV hue shift, bit-exact against a restatement pinned on Pillow) + ToTensor run as two HIP kernels (`csrc/input.hip`) on a side stream, one batch ahead of the step that consumes them
(double-buffered: batch k+1 is uploaded and transformed while step k trains).

The random draws follow torchvision's published algorithms (RandomResizedCrop.get_params, ColorJitter.get_params, the flip
coin tosses) from a numpy RandomState per pipeline: the SAME distribution as the reference's, not the same stream —
torchvision itself is absent from this image and the reference seeds nothing (its runs are not reproducible either).

    pipe = DeviceInputPipeline(FolderPairs(root, "summer2winter", "train"), batch_size, image_size, device, recipe="summer2winter")
    for batch in pipe:            # {'x': (B,3,S,S) view of NHWC, 'y': ...} on the device, like the reference's DataLoader
        model.training_step(batch)
"""
import ctypes
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _native, ops

# what each reference dataset applies to a TRAINING sample (train.py:184-190, 248-262, 309-319) and to a test sample
RECIPES = {
    "summer2winter": dict(hflip=0.5, vflip=0.0, scale=(0.33, 1.0), jitter=(0.2, 0.2, 0.2, 0.1), shared_draw=False),
    "maps": dict(hflip=0.5, vflip=0.0, scale=(0.33, 1.0), jitter=None, shared_draw=True),       # same RNG state for both halves
    # hypersim: both modalities of a sample go through the SAME random state (Data_Manager.py:160-165), the `color` modality is
    # colour-jittered first, on the whole frame, then flipped / cropped / resized like the others (:169-174).
    # AS WRITTEN, that replay of the RNG state does not give `color` the other modality's geometry: its ColorJitter consumes
    # draws (randperm(4) + four uniforms) BEFORE the flips and the crop are drawn, so those come from further down the
    # stream — a `color` <-> `depth` pair is flipped and cropped independently (an x / y misalignment in the reference's
    # paired hypersim training whenever one modality is `color`).  This build reproduces the reference as written
    # (`color_shares_geometry=False`: an independent geometry draw for the colour modality), like its other quirks (VAEGAN's
    # detach); VCG_HYPERSIM_ALIGNED=1 gives both modalities one geometry instead.
    "hypersim": dict(hflip=0.5, vflip=0.3, scale=(0.33, 1.0), jitter=None, pre_jitter=(0.3, 0.3, 0.3, 0.15), shared_draw=True,
                     color_shares_geometry=os.environ.get("VCG_HYPERSIM_ALIGNED", "0") == "1"),
    "test": dict(hflip=0.0, vflip=0.0, scale=None, jitter=None, shared_draw=False, bilinear=True),   # Resize((S, S)) + ToTensor
}


def draw_crop(rng, height, width, scale):
    """torchvision RandomResizedCrop.get_params, ratio (1, 1): ten attempts at a square of area U(scale) x image area,
    then the centre-crop fallback.  -> (y0, x0, h, w)"""
    area = height * width
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        w = h = int(round(np.sqrt(target)))
        if 0 < w <= width and 0 < h <= height:
            return int(rng.randint(0, height - h + 1)), int(rng.randint(0, width - w + 1)), h, w
    side = min(height, width)                                     # ratio clamps to 1: the largest centred square
    return (height - side) // 2, (width - side) // 2, side, side


def draw_sample(rng, height, width, recipe):
    """One sample's random parameters: (int32[16] geometry, float32[8] colour jitter) as csrc/input.hip reads them
    (the arena offset, slots 0-1, is filled in when the image is placed)."""
    g = np.zeros(16, np.int32)
    j = np.zeros(8, np.float32)
    g[2], g[3] = height, width
    g[8] = int(rng.uniform() < recipe["hflip"]) if recipe["hflip"] > 0 else 0
    g[9] = int(rng.uniform() < recipe["vflip"]) if recipe["vflip"] > 0 else 0
    if recipe.get("scale") is not None:
        g[4:8] = draw_crop(rng, height, width, recipe["scale"])
    else:
        g[4:8] = (0, 0, height, width)
    g[10] = 1 if recipe.get("bilinear") else 0
    g[12] = 1                                  # onto the uint8 grid: the reference resizes a PIL image (train.py:309-319)
    if recipe.get("jitter"):
        j = draw_jitter(rng, recipe["jitter"])
    return g, j


def draw_jitter(rng, strengths):
    """torchvision ColorJitter.get_params: the order of the four ops and their factors -> float32[8] as csrc/input.hip reads them"""
    b, c, s, h = strengths
    j = np.zeros(8, np.float32)
    order = rng.permutation(4)                                    # torchvision: fn_idx = torch.randperm(4)
    j[0] = 1.0
    j[1] = rng.uniform(max(0.0, 1 - b), 1 + b)
    j[2] = rng.uniform(max(0.0, 1 - c), 1 + c)
    j[3] = rng.uniform(max(0.0, 1 - s), 1 + s)
    j[4] = rng.uniform(-h, h)
    j[5] = float(int(order[0]) + 4 * int(order[1]) + 16 * int(order[2]) + 64 * int(order[3]))
    return j


class SyntheticImages:
    """uint8 HWC image pairs of varying size drawn from a RandomState: smooth low-frequency content plus noise, so that
    resampling and jitter have something to act on.  For tests and for benchmarking the pipeline without a dataset."""

    def __init__(self, count, min_side=300, max_side=640, seed=0, paired=False, pre_jitter=(False, False)):
        self.count, self.min_side, self.max_side, self.seed, self.paired = count, min_side, max_side, seed, paired
        self.pre_jitter = pre_jitter                  # which half plays hypersim's `color` modality (jittered before the crop)

    def __len__(self):
        return self.count

    def image(self, idx, which):
        rng = np.random.RandomState((self.seed * 1000003 + idx * 2 + which) % (2 ** 31 - 1))
        size_rng = np.random.RandomState((self.seed * 7919 + idx) % (2 ** 31 - 1)) if self.paired else rng
        h, w = (int(v) for v in size_rng.randint(self.min_side, self.max_side + 1, 2))      # paired: both halves of one image
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        img = np.empty((h, w, 3), np.float32)
        for c in range(3):
            fx, fy, ph = rng.uniform(0.005, 0.05), rng.uniform(0.005, 0.05), rng.uniform(0, 6.28)
            img[..., c] = 127.5 + 90.0 * np.sin(fx * xx + fy * yy + ph) + rng.uniform(-20, 20)
        img += rng.normal(0, 8.0, img.shape)
        return np.clip(img, 0, 255).astype(np.uint8)

    def pair(self, idx, rng):
        return self.image(idx, 0), self.image(idx if self.paired else int(rng.randint(0, self.count)), 1)


class FolderPairs:
    """The reference's on-disk layouts, decoded with PIL (Data_Manager.py:327-451):
    `summer2winter`: <root>/<split>A, <root>/<split>B, x = A[idx % len(A)], y = a uniformly drawn B image (:438);
    `maps`: <root>/<split>/*.jpg, each 2W x H: left half = x (satellite), right half = y (map) (:379-388)."""

    EXT = (".jpg", ".jpeg", ".png")

    def __init__(self, root_dir, layout, split="train"):
        from PIL import Image
        self._Image = Image
        self.layout = layout
        if layout == "summer2winter":
            self.dir_a, self.dir_b = os.path.join(root_dir, f"{split}A"), os.path.join(root_dir, f"{split}B")
            for d in (self.dir_a, self.dir_b):
                if not os.path.isdir(d):
                    raise ValueError(f"Directory not found: {d}")
            self.a = sorted(f for f in os.listdir(self.dir_a) if f.lower().endswith(self.EXT))
            self.b = sorted(f for f in os.listdir(self.dir_b) if f.lower().endswith(self.EXT))
            if not self.a:
                raise ValueError(f"No images found in {self.dir_a}")
            if not self.b:
                raise ValueError(f"No images found in {self.dir_b}")
            print(f"  Loaded {split} split: {len(self.a)} domain A, {len(self.b)} domain B images")
        elif layout == "maps":
            self.dir = os.path.join(root_dir, split)
            if not os.path.isdir(self.dir):
                raise ValueError(f"Directory not found: {self.dir}")
            self.files = sorted(f for f in os.listdir(self.dir) if f.lower().endswith(self.EXT))
            if not self.files:
                raise ValueError(f"No images found in {self.dir}")
            print(f"  Loaded {split} split with {len(self.files)} samples")
        else:
            raise ValueError(f"unknown layout {layout!r}")

    def __len__(self):
        return max(len(self.a), len(self.b)) if self.layout == "summer2winter" else len(self.files)

    def _open(self, path):
        return np.asarray(self._Image.open(path).convert("RGB"))

    def pair(self, idx, rng):
        if self.layout == "summer2winter":
            return (self._open(os.path.join(self.dir_a, self.a[idx % len(self.a)])),
                    self._open(os.path.join(self.dir_b, self.b[int(rng.randint(0, len(self.b)))])))
        img = self._open(os.path.join(self.dir, self.files[idx]))
        half = img.shape[1] // 2
        return np.ascontiguousarray(img[:, :half]), np.ascontiguousarray(img[:, half:2 * half])


class HypersimFolders:
    """The reference's Hypersim tree as PNG frames (Data_Manager.py:18-138; the HDF5 download / conversion that produces it is
    not part of the training path): <root>/<scene>/cam_XX/frame_NNNN_<modality>.png.  A sample needs every requested modality;
    paired: x, y = the two modalities of one frame (one modality: x = y); unpaired: y is the second modality of a uniformly
    drawn frame (:236-239).  `pre_jitter[k]` says whether half k is the `color` modality, which the reference colour-jitters
    before the spatial transform."""

    def __init__(self, root_dir, modalities, paired=True, indices=None):
        from PIL import Image
        self._Image = Image
        if paired and len(modalities) not in (1, 2):
            raise ValueError(f"paired_mode requires 1 or 2 modalities, got {len(modalities)}")
        if not paired and len(modalities) != 2:
            raise ValueError("Unpaired mode requires exactly 2 modalities")
        self.modalities, self.paired = list(modalities), paired
        self.pre_jitter = (modalities[0] == "color", modalities[-1] == "color")
        samples = []
        if not os.path.isdir(root_dir):
            raise ValueError(f"No samples found in {root_dir}")
        for scene in sorted(os.listdir(root_dir)):
            sdir = os.path.join(root_dir, scene)
            if not os.path.isdir(sdir):
                continue
            for cam in sorted(d for d in os.listdir(sdir) if d.startswith("cam_") and os.path.isdir(os.path.join(sdir, d))):
                cdir = os.path.join(sdir, cam)
                first = self.modalities[0]
                for f in sorted(os.listdir(cdir)):
                    if not (f.startswith("frame_") and f.endswith(f"_{first}.png")):
                        continue
                    frame_id = f[:-4].split("_")[1]
                    paths = [os.path.join(cdir, f"frame_{frame_id}_{m}.png") for m in self.modalities]
                    if all(os.path.exists(q) for q in paths):            # a frame that lacks a modality is skipped (:118-122)
                        samples.append(paths)
        if not samples:
            raise ValueError(f"No samples found in {root_dir}")
        self.all_samples = samples
        self.samples = samples if indices is None else [samples[i] for i in indices]
        if indices is None:
            print(f"  Loaded dataset with {len(samples)} samples\n  Modalities: {', '.join(self.modalities)}")

    def subset(self, indices):
        """a view on some of the samples (train / test split, train.py:208-214); unpaired y draws stay within the subset's parent"""
        sub = HypersimFolders.__new__(HypersimFolders)
        sub.__dict__.update(self.__dict__)
        sub.samples = [self.samples[i] for i in indices]
        return sub

    def __len__(self):
        return len(self.samples)

    def _open(self, path):
        return np.asarray(self._Image.open(path).convert("RGB"))

    def pair(self, idx, rng):
        x = self._open(self.samples[idx][0])
        if self.paired:
            return x, (x if len(self.modalities) == 1 else self._open(self.samples[idx][1]))
        j = int(rng.randint(0, len(self.all_samples)))                    # random.randint over the whole dataset (:238)
        return x, self._open(self.all_samples[j][1])


class DeviceInputPipeline:
    """len()-able iterable of {'x', 'y'} device batches (the reference's DataLoader surface, train.py:80-97).

    Two pinned staging arenas and two sets of device buffers: while the consumer trains on batch k, batch k+1 is decoded
    (thread pool), copied host->device and transformed on the side stream; `__next__` only makes the consumer's stream wait
    on the event recorded behind those kernels."""

    def __init__(self, source, batch_size, image_size, device, recipe="summer2winter", shuffle=True, drop_last=False, seed=0,
                 num_workers=4, same_xy=False, arena_bytes=None, rank=0, world=1, even_shards=None):
        """`rank` / `world`: data parallelism — every rank draws the SAME per-epoch permutation (from `seed`, which must then
        be equal on all ranks) and takes every world-th sample of it, truncated to equal length, so an epoch passes over the
        data once and no sample appears twice in a global batch; the augmentation draws come from a per-rank stream."""
        self.src, self.b, self.s, self.dev = source, batch_size, image_size, device
        self.recipe = RECIPES[recipe] if isinstance(recipe, str) else recipe
        self.shuffle, self.drop_last, self.same_xy = shuffle, drop_last, same_xy
        self.rank, self.world = int(rank), max(1, int(world))
        self.order_rng = np.random.RandomState(seed)                     # the permutation: identical on every rank
        self.rng = np.random.RandomState((seed * 64 + self.rank) % (2 ** 31 - 1)) if self.world > 1 else self.order_rng
        self.pool = ThreadPoolExecutor(max_workers=max(1, num_workers))
        self.prep = ThreadPoolExecutor(max_workers=1)                     # runs _prepare one batch ahead; its Future carries exceptions
        # even_shards (default: for shuffled = training loaders): every rank gets len // world samples, so that all ranks take the
        # same number of steps (each step ends in a gradient exchange).  A validation loader (shuffle=False) has no collective
        # inside its steps: it covers the WHOLE set — rank r takes samples r, r + world, ... — and `train.validate` weights the
        # ranks by what they saw (up to world - 1 test samples used to be dropped, and a test split smaller than
        # `world` gave every rank an empty loader and a division by zero)
        self.even_shards = bool(shuffle) if even_shards is None else bool(even_shards)
        if self.world > 1 and not self.even_shards:
            n = len(range(self.rank, len(source), self.world))
        else:
            n = len(source) // self.world
        self.nbatches = n // batch_size if drop_last else (n + batch_size - 1) // batch_size
        self.arena_bytes = arena_bytes or max(2 * batch_size * 1024 * 1024 * 3, 1 << 22)   # grown on demand
        self.slots = [self._make_slot() for _ in range(2)]
        self.stream = torch.cuda.Stream(device=device)
        self.last_draws = None                   # (geometry, jitter, sources) of the batch handed out last: tests read it

    def _make_slot(self):
        pin = torch.empty(self.arena_bytes, dtype=torch.uint8).pin_memory()
        return {"pin": pin, "arena": torch.empty(self.arena_bytes, dtype=torch.uint8, device=self.dev),
                "gpin": torch.empty((2 * self.b, 16), dtype=torch.int32).pin_memory(),
                "jpin": torch.empty((2 * self.b, 8), dtype=torch.float32).pin_memory(),
                "g": torch.empty((2 * self.b, 16), dtype=torch.int32, device=self.dev),
                "j": torch.empty((2 * self.b, 8), dtype=torch.float32, device=self.dev),
                "out": torch.empty((2 * self.b, self.s, self.s, 4), dtype=torch.float32, device=self.dev),
                "fpin": torch.empty((2 * self.b, 8), dtype=torch.int32).pin_memory(),      # pre-jitter: frames, var (csrc/input.hip)
                "vpin": torch.empty((2 * self.b, 4), dtype=torch.int32).pin_memory(),
                "pjpin": torch.empty((2 * self.b, 8), dtype=torch.float32).pin_memory(),
                "f": torch.empty((2 * self.b, 8), dtype=torch.int32, device=self.dev),
                "v": torch.empty((2 * self.b, 4), dtype=torch.int32, device=self.dev),
                "pj": torch.empty((2 * self.b, 8), dtype=torch.float32, device=self.dev),
                "fbuf": None,
                "event": torch.cuda.Event(), "free": torch.cuda.Event(), "n": 0, "draws": None}

    def __len__(self):
        return self.nbatches

    def _prepare(self, slot, indices):
        """decode (pool) + draw + pack into the pinned arena, then enqueue upload and kernels on the side stream"""
        with torch.cuda.device(self.dev):          # a fresh thread's current device is 0: pin / allocate on THIS rank's device
            self._prepare_on_device(slot, indices)

    def _prepare_on_device(self, slot, indices):
        # the per-sample seeds (the unpaired y draws) are drawn HERE, in index order, not inside the pool's worker threads:
        # their completion order is not deterministic and the run would not be reproducible for a given seed
        seeds = [int(self.rng.randint(0, 2 ** 31 - 1)) for _ in indices]
        pairs = list(self.pool.map(lambda a: self.src.pair(a[0], np.random.RandomState(a[1])), zip(indices, seeds)))
        imgs = [p[0] for p in pairs] + [p[1] for p in pairs]       # first the x images, then the y images
        need = sum(im.size for im in imgs)
        if need > slot["pin"].numel():                             # a batch of larger images than any before: grow both sides
            slot["free"].synchronize()
            cap = int(need * 1.5)
            slot["pin"] = torch.empty(cap, dtype=torch.uint8).pin_memory()
            slot["arena"] = torch.empty(cap, dtype=torch.uint8, device=self.dev)
        nb = len(pairs)
        geo, jit = np.zeros((2 * nb, 16), np.int32), np.zeros((2 * nb, 8), np.float32)
        # hypersim as written: exactly one `color` modality breaks the shared geometry (RECIPES["hypersim"])
        col = getattr(self.src, "pre_jitter", (False, False)) if self.recipe.get("pre_jitter") else (False, False)
        shared = self.recipe.get("shared_draw") and (col[0] == col[1] or self.recipe.get("color_shares_geometry", True))
        for k in range(nb):
            gx, jx = draw_sample(self.rng, imgs[k].shape[0], imgs[k].shape[1], self.recipe)
            if shared and getattr(self.src, "paired", True) and imgs[nb + k].shape == imgs[k].shape:
                gy, jy = gx.copy(), jx.copy()                      # maps: the same RNG state transforms both halves
            else:
                gy, jy = draw_sample(self.rng, imgs[nb + k].shape[0], imgs[nb + k].shape[1], self.recipe)
            geo[k], jit[k], geo[nb + k], jit[nb + k] = gx, jx, gy, jy
        slot["free"].synchronize()                                 # the kernels that read this slot's arena last have finished
        pin = slot["pin"].numpy()
        off = 0
        for k, im in enumerate(imgs):
            flat = np.ascontiguousarray(im).reshape(-1)
            pin[off:off + flat.size] = flat
            geo.view(np.uint32)[k, 0], geo.view(np.uint32)[k, 1] = off & 0xFFFFFFFF, off >> 32
            off += flat.size
        # hypersim's colour modality: ColorJitter on the whole frame before the crop.  Those frames are unpacked to float4 in
        # `fbuf`, jittered there, and the resample reads them from it (geometry slot 11 = 1, offset = pixel index)
        pre = self.recipe.get("pre_jitter")
        which = getattr(self.src, "pre_jitter", (False, False)) if pre else (False, False)
        pj_rows = [k for k in range(2 * nb) if which[0 if k < nb else 1]]
        frames, var, pjit = np.zeros((2 * nb, 8), np.int32), np.zeros((2 * nb, 4), np.int32), np.zeros((2 * nb, 8), np.float32)
        fpx = 0
        for r, k in enumerate(pj_rows):
            npx = imgs[k].shape[0] * imgs[k].shape[1]
            frames.view(np.uint32)[r, 0], frames.view(np.uint32)[r, 1] = geo.view(np.uint32)[k, 0], geo.view(np.uint32)[k, 1]
            frames[r, 2] = npx
            frames.view(np.uint32)[r, 3], frames.view(np.uint32)[r, 4] = fpx & 0xFFFFFFFF, fpx >> 32
            var.view(np.uint32)[r, 0], var.view(np.uint32)[r, 1], var[r, 2] = fpx & 0xFFFFFFFF, fpx >> 32, npx
            if self.same_xy and k >= nb:
                pjit[r] = pjit[pj_rows.index(k - nb)]                 # one modality: x = y is ONE transformed image
            else:
                pjit[r] = draw_jitter(self.rng, pre)
            geo.view(np.uint32)[k, 0], geo.view(np.uint32)[k, 1], geo[k, 11] = fpx & 0xFFFFFFFF, fpx >> 32, 1
            fpx += npx
        if pj_rows and (slot["fbuf"] is None or slot["fbuf"].numel() < fpx * 4):
            slot["fbuf"] = torch.empty(int(fpx * 4 * 1.25), dtype=torch.float32, device=self.dev)
        slot["gpin"][:2 * nb].copy_(torch.from_numpy(geo))
        slot["jpin"][:2 * nb].copy_(torch.from_numpy(jit))
        if pj_rows:
            slot["fpin"][:2 * nb].copy_(torch.from_numpy(frames))
            slot["vpin"][:2 * nb].copy_(torch.from_numpy(var))
            slot["pjpin"][:2 * nb].copy_(torch.from_numpy(pjit))
        lib = _native.lib()
        with torch.cuda.stream(self.stream):
            slot["arena"][:off].copy_(slot["pin"][:off], non_blocking=True)
            slot["g"][:2 * nb].copy_(slot["gpin"][:2 * nb], non_blocking=True)
            slot["j"][:2 * nb].copy_(slot["jpin"][:2 * nb], non_blocking=True)
            st = ctypes.c_void_p(self.stream.cuda_stream)
            P = lambda t: ctypes.c_void_p(t.data_ptr())
            if pj_rows:
                for dst, srcp in (("f", "fpin"), ("v", "vpin"), ("pj", "pjpin")):
                    slot[dst][:2 * nb].copy_(slot[srcp][:2 * nb], non_blocking=True)
                _native.check(lib.vcg_input_prejitter(P(slot["arena"]), P(slot["f"]), P(slot["pj"]), P(slot["v"]), P(slot["fbuf"]),
                                                      len(pj_rows), st), "vcg_input_prejitter")
            _native.check(lib.vcg_input_resample(P(slot["arena"]), P(slot["fbuf"]) if pj_rows else None, P(slot["g"]), P(slot["out"]),
                                                 2 * nb, self.s, st), "vcg_input_resample")
            if self.recipe.get("jitter"):
                _native.check(lib.vcg_input_color_jitter(ctypes.c_void_p(slot["out"].data_ptr()), ctypes.c_void_p(slot["j"].data_ptr()),
                                                         2 * nb, self.s, st), "vcg_input_color_jitter")
            slot["event"].record(self.stream)
        slot["n"] = nb
        slot["draws"] = (geo, jit, imgs, {k: pjit[r] for r, k in enumerate(pj_rows)})

    def __iter__(self):
        n = len(self.src)
        order = self.order_rng.permutation(n) if self.shuffle else np.arange(n)
        if self.world > 1:                                         # this rank's shard: every world-th sample, equal length on all ranks
            order = (order[:n - n % self.world] if self.even_shards else order)[self.rank::self.world]
        batches = [order[i * self.b:(i + 1) * self.b] for i in range(self.nbatches)]
        if not batches:
            return
        pending = self.prep.submit(self._prepare, self.slots[0], batches[0])
        for k in range(len(batches)):
            # .result() re-raises whatever _prepare raised (a truncated file, a failed vcg_input_* call, an allocation error):
            # a bare thread would die silently and this loop would hand out the slot's previous batch
            pending.result()
            slot = self.slots[k % 2]
            if k + 1 < len(batches):                               # batch k+1 is prepared while the caller trains on batch k
                pending = self.prep.submit(self._prepare, self.slots[(k + 1) % 2], batches[k + 1])
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_event(slot["event"])
            nb = slot["n"]
            # the consumer gets its own tensors (2 x B x S x S x 4 floats: a device-to-device copy, 4 MB per 8 images), so
            # the slot can be refilled while those are still referenced by the autograd graph of the step
            out = slot["out"][:2 * nb].clone()
            slot["free"].record(cur)
            self.last_draws = slot["draws"]
            x = ops.logical_of(out[:nb], 3)
            y = x if self.same_xy else ops.logical_of(out[nb:2 * nb], 3)
            yield {"x": x, "y": y}
