"""Writes a small synthetic tree in the reference's Hypersim layout (<root>/hypersim/<scene>/cam_XX/frame_NNNN_<modality>.png,
Data_Manager.py:18-138) so that `train.py --dataset hypersim` can be exercised without the dataset:

    python tools/make_hypersim_sample.py /tmp/ds
    python train.py --architecture cyclevaegan --dataset hypersim --data_dir /tmp/ds --source_modality color \\
        --target_modality depth --paired --batch_size 2 --epochs 1 --test_split 0.2
"""
import os
import sys

import numpy as np
from PIL import Image


def main(root):
    rng = np.random.RandomState(0)
    for scene, cam, n in (("ai_001_001_unknown", "cam_00", 14), ("ai_001_002_kitchen", "cam_01", 10)):
        d = os.path.join(root, "hypersim", scene, cam)
        os.makedirs(d, exist_ok=True)
        for f in range(n):
            for m in ("color", "depth"):
                yy, xx = np.mgrid[0:240, 0:320]
                img = np.stack([127 + 100 * np.sin(0.02 * xx * (c + 1) + 0.03 * yy + f) for c in range(3)], -1)
                img = img + rng.normal(0, 5, (240, 320, 3))
                Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(d, f"frame_{f:04d}_{m}.png"))
    print(f"wrote {root}/hypersim")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/ds")
