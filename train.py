#!/usr/bin/env python3
"""`python train.py --architecture ... --dataset synthetic ...` — the reference's entry-point name.
The implementation lives in vae-cyclegan-implementation_amd/train.py."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
_impl = importlib.import_module("vae-cyclegan-implementation_amd.train")
create_model, train_epoch, build_parser, main = _impl.create_model, _impl.train_epoch, _impl.build_parser, _impl.main

if __name__ == "__main__":
    main(build_parser().parse_args())
