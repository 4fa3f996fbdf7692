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
# The step runs on two streams (data-gradient chain / weight gradients, ops.wgrad_overlap) and the HIP runtime spreads a
# process's streams over GPU_MAX_HW_QUEUES hardware queues, 4 by default.  A process group adds streams of its own (RCCL's, the
# reducer's launch stream), and with 4 queues the weight-gradient stream then shares a queue with the main stream: the two
# serialise and the overlap is gone — a ONE-rank RCCL run of the bench step measured 42.7 ms against 36.7 without a process
# group, with not one collective waited for (profiles/r04_dp_one_rank.txt); with 8 queues 36.3 against 35.9.  Same rule as
# above: in the environment before the first HIP call; an explicit value wins.  (SEVERAL processes on one card — a rehearsal, never
# the deployment — want 4: 2 x 8 queues oversubscribe the card's queue slots and a two-rank step was seen to hang.)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _native, ops, optim, synth  # noqa: F401,E402
from . import Losses, Networks, input_pipeline, parallel, utils  # noqa: F401,E402
from . import custom_ops  # noqa: F401,E402  (registers torch.ops.vcg.*)

__all__ = ["_native", "ops", "optim", "synth", "Losses", "Networks", "input_pipeline", "parallel", "utils", "custom_ops"]
