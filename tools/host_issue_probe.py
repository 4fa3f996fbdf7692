"""How long does the HOST need to issue one CycleVAEGAN step, against how long the GPU needs to run it?

`training_step` ends with one device->host copy of the metric vector (Networks._metrics_to_host); everything before it
is asynchronous launches.  This probe stamps the host clock when that read-back is entered (= every kernel of the step
has been issued) and when it returns (= the GPU has finished): if the first is close to the second, the step is
host-bound in places and faster kernels will not show.

    python tools/host_issue_probe.py [--steps 10]
"""
import argparse
import gc
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--defer", action="store_true", help="what-if: copy the metrics to pinned memory without waiting for them "
                    "(no per-step host sync); reports the mean step time over --steps against the synchronous loop")
    args = ap.parse_args()
    pkg = importlib.import_module("vae-cyclegan-implementation_amd")
    ops, N = pkg.ops, pkg.Networks
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = N.CycleVAEGAN(latent_dim=64, paired=False).to(dev).train()
    model.configure_optimizers(lr=2e-4)
    model.configure_loss(lambda_kl=1e-5, lambda_gan=1.0, lambda_identity=5.0, lambda_cycle=10.0, lambda_recon=1.0)
    ops.manual_seed(4321)
    B, S = args.batch, 256
    nq = (B * 3 * S * S + 3) // 4
    pool = [{"x": ops.to_nhwc(ops.rand_uniform((B, 3, S, S), dev, seed=1234, offset=2 * i * nq)),
             "y": ops.to_nhwc(ops.rand_uniform((B, 3, S, S), dev, seed=1234, offset=(2 * i + 1) * nq))} for i in range(4)]
    stamps = []
    inner = N._metrics_to_host

    def stamped(named, reducer=None):
        stamps.append(time.perf_counter())
        return inner(named, reducer)

    if args.defer:
        def deferred(named, reducer=None):
            keys = list(named.keys())
            vec = torch.stack([named[k].detach().reshape(()) for k in keys])
            host = torch.empty(vec.shape, dtype=vec.dtype, pin_memory=True)
            host.copy_(vec, non_blocking=True)
            return dict.fromkeys(keys, 0.0)

        for mode, fn in (("synchronous read-back", inner), ("deferred read-back", deferred), ("synchronous read-back", inner),
                         ("deferred read-back", deferred)):
            N._metrics_to_host = fn
            for i in range(3):
                model.training_step(pool[i % 4])
            gc.collect()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                model.training_step(pool[i % 4])
            torch.cuda.synchronize()
            print(f"{mode}: {(time.perf_counter() - t0) / args.steps * 1e3:.2f} ms/step over {args.steps} steps")
        return
    N._metrics_to_host = stamped
    for i in range(3):
        model.training_step(pool[i % 4])
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    rows = []
    for i in range(args.steps):
        stamps.clear()
        t0 = time.perf_counter()
        model.training_step(pool[i % 4])
        t1 = time.perf_counter()
        rows.append(((stamps[-1] - t0) * 1e3, (t1 - t0) * 1e3, len(stamps)))
    for issue, total, n in rows:
        print(f"host issued the step in {issue:7.2f} ms; step done after {total:7.2f} ms ({n} read-back(s))")
    issue = sorted(r[0] for r in rows)[len(rows) // 2]
    total = sorted(r[1] for r in rows)[len(rows) // 2]
    print(f"median: issue {issue:.2f} ms, step {total:.2f} ms -> the host is idle {100 * (1 - issue / total):.0f} % of the step")


if __name__ == "__main__":
    main()
