"""MI355X-native VAE-CycleGAN training path (drop-in for the reference's Networks.py /
Losses.py / train.py surface).  The directory name is not a Python identifier; import it with
`importlib.import_module("vae-cyclegan-implementation_amd")` or through the repo-root shim
`vcg_amd`.
"""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: a step is ~1 400 dependent launches, and with the default
# the dispatch gap between two of them is what 2.7 % of the step goes to (measured: 174.5 -> 179.2 images/s on the same box,
# DESIGN.md §5).  A HIP runtime setting: it has to be in the environment before the first HIP call of the process, so it is
# set on import (importing torch does not initialise HIP); an explicit value in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import _native, ops, optim, synth  # noqa: F401,E402
from . import Losses, Networks, input_pipeline, parallel, utils  # noqa: F401,E402

__all__ = ["_native", "ops", "optim", "synth", "Losses", "Networks", "input_pipeline", "parallel", "utils"]
