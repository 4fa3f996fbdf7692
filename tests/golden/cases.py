"""Case table shared by make_golden.py (which runs the reference) and the tests (which do not)."""

SEED = 20261003
LR = 2e-4
LAMBDAS = dict(lambda_kl=1e-5, lambda_gan=1.0, lambda_identity=5.0, lambda_cycle=10.0, lambda_recon=1.0)

ATOM_CASES = {
    # name: (class, ctor args, ctor kwargs, input shape, input scale)
    "casb_stem": ("CaSb", (3, 8, 7), {}, (2, 3, 12, 10), 1.0),
    "casb_head": ("CaSb", (8, 3, 7), {"activation": "Identity", "use_norm": False}, (2, 8, 9, 11), 1.0),
    "casb_disc0": ("CaSb", (3, 8, 4), {"stride": 2, "padding": 1, "activation": "LeakyReLU", "use_norm": False}, (2, 3, 12, 16), 1.0),
    "casb_disc1": ("CaSb", (8, 16, 4), {"stride": 2, "padding": 1, "activation": "LeakyReLU"}, (2, 8, 12, 8), 1.0),
    "d": ("D", (8, 16), {}, (2, 8, 12, 8), 1.0),
    "d_wide": ("D", (16, 40), {}, (1, 16, 8, 20), 1.0),
    "r": ("R", (8,), {}, (2, 8, 6, 10), 1.0),
    "u": ("U", (16, 8), {}, (2, 16, 5, 6), 1.0),
    "u_shuf": ("U", (32, 16), {}, (1, 32, 4, 6), 1.0),
    "s": ("S", (8, 12), {}, (1, 8, 7, 5), 1.0),
}
ATOM_BIAS_STD = 0.1
STEP_BIAS_STD = 0.02
DISC_BIAS_STD = 0.05


# ---- tensor checksum shared by the generator and the tests ---------------------------------
# [mean, L2 norm, 8 projections on fixed pseudo-random +-1 vectors, 16 sampled elements].
# The projections bound the L2 distance between two tensors (for d = a - b, <d, r> has standard
# deviation ||d||), which is the meaningful parity measure for gradients of a deep fp32 network:
# individual elements of such gradients differ by >1e-3 of their scale between two equally valid
# fp32 evaluations (measured: oracle vs reference on CPU), while ||d||/||b|| stays ~1e-5.
N_PROJ = 8
N_SAMPLES = 16


def sample_idx(numel, k=N_SAMPLES):
    import numpy as np
    return (np.arange(k, dtype=np.int64) * 2654435761) % numel


def _signs(numel, k):
    import numpy as np
    idx = np.arange(numel, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = (idx + np.uint64(k + 1) * np.uint64(0x9E3779B97F4A7C15)) * np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(29)
        h *= np.uint64(0x94D049BB133111EB)
    return 1.0 - 2.0 * ((h >> np.uint64(40)) & np.uint64(1)).astype(np.float64)


def checksum(a):
    import numpy as np
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    proj = [float(np.dot(a, _signs(a.size, k))) for k in range(N_PROJ)]
    return np.concatenate([[a.mean(), np.linalg.norm(a)], proj, a[sample_idx(a.size)]])
