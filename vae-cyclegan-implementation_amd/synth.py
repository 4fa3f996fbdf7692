"""Deterministic synthetic tensors: integer hash -> uniform -> Box-Muller, numpy only.

Every tensor is a pure function of (seed, name): parameters keyed by state_dict name,
batches keyed by ("x"|"y", step, rank), eps keyed by call index.  The golden fixtures under
tests/golden were produced by feeding exactly these tensors to the reference, so the same
inputs can be rebuilt on any machine without shipping them.
"""
import os
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def stream_id(name):
    return np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF)


def _bits(n, seed, name, lane):
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) ^ (stream_id(name) << np.uint64(32)) ^ np.uint64(lane))
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix64(base + idx * np.uint64(0x9E3779B97F4A7C15))


def uniform(shape, seed, name):
    """float32 in [0, 1)"""
    n = int(np.prod(shape))
    u = (_bits(n, seed, name, 0) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.astype(np.float32).reshape(shape)


def normal(shape, seed, name, std=1.0):
    """float32 N(0, std^2) by Box-Muller in float64"""
    n = int(np.prod(shape))
    u1 = ((_bits(n, seed, name, 1) >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    u2 = (_bits(n, seed, name, 2) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (z * std).astype(np.float32).reshape(shape)


def state_dict_like(shapes, seed, bias_std=0.0):
    """{name: ndarray} reproducing the reference's init statistics (Networks.py:168-178, 1893-1903):
    conv weights Kaiming-normal fan_out (std = sqrt(2 / (Cout*kh*kw))), biases zero (or N(0, bias_std^2) to
    exercise the bias paths in parity tests), spectral-norm u/v unit vectors."""
    def one(item):
        name, shape = item[0], tuple(item[1])
        if name.endswith("weight_u") or name.endswith("weight_v"):
            v = normal(shape, seed, name).astype(np.float64)
            v = v / max(np.sqrt((v * v).sum()), 1e-12)
            return v.astype(np.float32)
        if name.endswith("weight") or name.endswith("weight_orig"):
            fan_out = shape[0] * int(np.prod(shape[2:]))
            return normal(shape, seed, name, std=float(np.sqrt(2.0 / fan_out)))
        if name.endswith("bias"):
            return normal(shape, seed, name, std=bias_std) if bias_std > 0 else np.zeros(shape, np.float32)
        raise KeyError(f"don't know how to synthesise {name}")

    # every tensor is a pure function of (seed, name): build them on a few threads (numpy releases the GIL); a full
    # CycleVAEGAN (138 M values) takes ~50 s on one core, and every full-model test pays it
    items = list(shapes.items())
    with ThreadPoolExecutor(max_workers=max(1, min(6, os.cpu_count() or 1))) as pool:
        return dict(zip((k for k, _ in items), pool.map(one, items)))


def batch(n, size, seed, step=0, rank=0):
    """x, y ~ U[0,1), shape (n, 3, size, size)"""
    x = uniform((n, 3, size, size), seed, f"x/{step}/{rank}")
    y = uniform((n, 3, size, size), seed, f"y/{step}/{rank}")
    return x, y


def eps_list(count, shape, seed, step=0, rank=0):
    return [normal(shape, seed, f"eps/{step}/{rank}/{i}") for i in range(count)]
