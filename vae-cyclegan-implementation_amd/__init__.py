"""MI355X-native VAE-CycleGAN training path (drop-in for the reference's Networks.py /
Losses.py / train.py surface).  The directory name is not a Python identifier; import it with
`importlib.import_module("vae-cyclegan-implementation_amd")` or through the repo-root shim
`vcg_amd`.
"""
from . import _native, ops, optim, synth  # noqa: F401
from . import Losses, Networks, input_pipeline, parallel, utils  # noqa: F401

__all__ = ["_native", "ops", "optim", "synth", "Losses", "Networks", "input_pipeline", "parallel", "utils"]
