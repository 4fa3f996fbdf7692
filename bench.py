#!/usr/bin/env python3
"""Headline benchmark: train images/sec of the 256x256 VAE-CycleGAN step (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one `CycleVAEGAN.training_step` (generator update + discriminator update) on one
synthetic batch per rank; batches are generated on the device before the timed region.  Rank 0
prints ONE JSON line.  `roofline` is measured live with HIP events around every launch of the
dominant DEVICE kernel during two extra instrumented steps (events recorded inside libvcg, vcg_profile_enable); `cpu_baseline` times the
oracle's CPU restatement of the same step on a bounded sample (rank 0, N=1 only).
"""
import argparse
import gc
import importlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes needs dmabuf IPC on this driver
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")        # kernel arguments in device memory: shorter dispatch gaps (package __init__)
if os.environ.get("VCG_ONE_DEVICE") == "1":
    # the gloo rehearsal puts SEVERAL processes on one card: with 8 hardware queues each the card's queue slots are oversubscribed
    # and the second step never finishes (both ranks wait in GradReducer.finish for a bucket; with 4 it takes 3 s) — measured, round 4
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")            # one hardware queue per stream once RCCL has added its own (package __init__)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
# The GEMMs run on the 16-bit matrix pipe with every fp32 operand scaled by a per-tensor power of two and split into two fp16
# pieces, three cross products accumulated in fp32 (csrc/vcg_common.h, csrc/gemm_split.hip): fp32-level results at 3 fp16 MFMAs
# per fp32 multiply-add (rounds 1-2: three bf16 pieces, 6 MFMAs).
BF16_MFMA_PEAK_TFLOPS = 16 * FP32_MFMA_PEAK_TFLOPS          # same guide: BF16 / FP16 dense = 16x the F32 MFMA rate (~2.5 PF)
SPLIT_MFMAS_PER_PRODUCT = 3
BF16X3_FP32_EQUIV_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / SPLIT_MFMAS_PER_PRODUCT   # 838.9: what that pipe can deliver in fp32-equivalent FLOPs
WORKLOADS = {
    "cyclevaegan": "cyclevaegan unpaired, 3x256x256 synthetic summer<->winter, per-GPU batch 8, latent 64 (BASELINE.json configs[3]/[4])",
    "vae": "vae latent 1024, 3x256x256 synthetic, batch 16 (BASELINE.json configs[2])",
    "autoencoder": "autoencoder, 3x256x256 synthetic, batch 16 (BASELINE.json configs[1])",
    # the other architectures of the reference's factory: same kernels, other wiring; not headline configs
    "cycleaegan": "cycleaegan unpaired, 3x256x256 synthetic, per-GPU batch 8",
    "cycleae": "cycleae unpaired, 3x256x256 synthetic, per-GPU batch 8",
    "cyclevae": "cyclevae unpaired, 3x256x256 synthetic, per-GPU batch 8, latent 64",
    "doubleae": "doubleae, 3x256x256 synthetic, per-GPU batch 8",
    "doublevae": "doublevae, 3x256x256 synthetic, per-GPU batch 8, latent 64",
    "aegan": "aegan, 3x256x256 synthetic, per-GPU batch 8",
    "vaegan": "vaegan, 3x256x256 synthetic, per-GPU batch 8, latent 64",
}
SAME_XY = ("autoencoder", "vae")          # the reference trains these on (x, x) (train.py:448-452: same modality)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cyclevaegan", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 8 cyclevaegan, 16 ae/vae)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--latent", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager-child", action="store_true", help=argparse.SUPPRESS)      # internal: the rocm_eager_baseline child process
    ap.add_argument("--no-eager-baseline", action="store_true",
                    help="skip the stock PyTorch-ROCm (MIOpen / rocBLAS eager) run of the same step beside cpu_baseline")
    args = ap.parse_args()

    if args.eager_child:
        # a process of its own (spawned by rocm_eager_baseline below): a MIOpen fault or hang there cannot take the bench line with it
        pkg = importlib.import_module("vae-cyclegan-implementation_amd")
        wl = args.workload
        B = args.batch or (16 if wl in SAME_XY else 8)
        latent = args.latent or (1024 if wl == "vae" else 64)
        torch.cuda.set_device(0)
        print(json.dumps(_eager_baseline_run(pkg, wl, B, args.size, latent, torch.device("cuda", 0))), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    trace_on = os.environ.get("VCG_BENCH_TRACE") == "1"
    t_trace = time.perf_counter()
    if trace_on:
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("VCG_BENCH_TRACE_AFTER", "90")), exit=False)     # where every thread is, if it stalls

    def trace(what):
        if trace_on:
            sys.stderr.write(f"[bench rank {rank} +{time.perf_counter() - t_trace:7.2f}s] {what}\n")
            sys.stderr.flush()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # VCG_FORCE_DIST=1: a ONE-rank process group (the only RCCL run a one-GPU box allows): the step takes the whole data-parallel
    # path — bucket launches from inside the backward on the reporting stream, RCCL's own stream, the waits before the optimizers,
    # the metric average — with nothing to exchange; the line then carries process_group / exchange like an N > 1 line
    dist_on = world > 1 or os.environ.get("VCG_FORCE_DIST") == "1"
    if dist_on and world == 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", "29531")
    # VCG_DIST_BACKEND=gloo + VCG_ONE_DEVICE=1: rehearsal of the N>1 path on a one-GPU box (all ranks on cuda:0,
    # exchange through the host); the driver's runs use the default: one rank per GPU, nccl (= RCCL over xGMI)
    backend = os.environ.get("VCG_DIST_BACKEND", "nccl")
    if os.environ.get("VCG_ONE_DEVICE") == "1":
        local_rank = 0
        # several ranks on one card: main + side + gloo's pool streams of every process oversubscribe the device's
        # hardware queues and the rehearsal crawls (4 s/step); it checks the exchange logic, so one stream is enough
        os.environ.setdefault("VCG_WGRAD_OVERLAP", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pkg = importlib.import_module("vae-cyclegan-implementation_amd")
    if os.path.exists(pkg._native.LIB_PATH) and os.environ.get("VCG_PRECREATE", "0") == "1":
        pkg.ops.create_streams(dev)            # diagnostic only (tools/dp_variants.sh): measured 42.8 ms against 34.3 without — see ops.create_streams
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on STDOUT when it creates a communicator; this program's stdout is one JSON line, so the
        # banner goes to stderr: fd 1 points at fd 2 while the group and its communicator are set up (device_id = eager init)
        sys.stdout.flush()
        saved_out = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, pg_options=pkg.parallel.nccl_options())     # default priority unless VCG_NCCL_PRIORITY=high
                dist.barrier()
            else:
                dist.init_process_group(backend)
        finally:
            sys.stdout.flush()
            os.dup2(saved_out, 1)
            os.close(saved_out)

    ops, N = pkg.ops, pkg.Networks
    # A tree without the built library (it normally travels with it): rank 0 builds, the others wait — never N concurrent
    # hipcc runs.  The decision is rank 0's alone and every rank takes the same collectives whatever it sees on disk (a rank
    # that started late and found the file rank 0 had just linked used to skip the barrier rank 0 was sitting in); the
    # link goes to a temporary name and is renamed into place (_native.build), so no rank maps a half-written file.
    if dist_on:
        flag = torch.tensor([0 if os.path.exists(pkg._native.LIB_PATH) else 1], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.broadcast(flag, src=0)
        if int(flag.item()) and rank == 0:
            pkg._native.build(verbose=False)
        dist.barrier()
    elif not os.path.exists(pkg._native.LIB_PATH):
        pkg._native.build(verbose=False)
    pkg._native.lib()                      # fail loudly if the HIP extension is missing
    trace("process group and library ready")

    wl = args.workload
    B = args.batch or (16 if wl in SAME_XY else 8)
    S = args.size
    latent = args.latent or (1024 if wl == "vae" else 64)
    torch.manual_seed(1234)                # identical random-init replicas on every rank
    model = {"cyclevaegan": lambda: N.CycleVAEGAN(latent_dim=latent, paired=False), "vae": lambda: N.VariationalAutoencoder(latent_dim=latent),
             "autoencoder": N.Autoencoder, "cycleaegan": lambda: N.CycleAEGAN(paired=False), "cycleae": lambda: N.CycleAE(paired=False),
             "cyclevae": lambda: N.CycleVAE(latent_dim=latent, paired=False), "doubleae": N.DoubleAutoencoder,
             "doublevae": lambda: N.DoubleVariationalAutoencoder(latent_dim=latent), "aegan": N.AEGAN,
             "vaegan": lambda: N.VAEGAN(latent_dim=latent)}[wl]()
    model = model.to(dev).train()
    model.configure_optimizers(lr=2e-4)
    model.configure_loss(lambda_kl=1e-5, lambda_gan=1.0, lambda_identity=5.0, lambda_cycle=10.0, lambda_recon=1.0)
    red = None
    if dist_on:
        red = pkg.parallel.attach(model)
        if os.environ.get("VCG_DP_NULL_EXCHANGE") == "1":
            # diagnostic (tools/dp_one_rank.sh): the reducer's bookkeeping and stream ordering with the collective itself left out
            class _Done:
                def wait(self):
                    return True
            _real = dist.all_reduce
            dist.all_reduce = lambda t, *a, **k: _Done() if k.get("async_op") else _real(t, *a, **k)
        if os.environ.get("VCG_DP_TIME_CALLS") == "1":
            # diagnostic: host time spent inside the asynchronous all_reduce calls (they are issued from the autograd thread)
            _real3 = dist.all_reduce
            _host = {"n": 0, "s": 0.0}

            def _timed_all_reduce(t, *a, **k):
                if not k.get("async_op"):
                    return _real3(t, *a, **k)
                t0_ = time.perf_counter()
                w = _real3(t, *a, **k)
                _host["s"] += time.perf_counter() - t0_
                _host["n"] += 1
                return w
            dist.all_reduce = _timed_all_reduce
            import atexit
            atexit.register(lambda: sys.stderr.write(f"async all_reduce calls: {_host['n']}, mean host time {1e6 * _host['s'] / max(_host['n'], 1):.1f} us\n"))
        busy = int(os.environ.get("VCG_DP_EMULATE_BUSY", "0"))
        if busy > 0 and world == 1:
            # diagnostic (tools/dp_variants.sh): a ONE-rank all_reduce moves nothing, so RCCL's stream is idle; make it as busy as an
            # N-rank exchange would — `busy` device copies of the bucket on the collectives' stream (a one-rank all_gather is a copy)
            # behind every asynchronous all_reduce — to see what a busy collective stream does to the step's other streams
            _real2 = dist.all_reduce
            _scratch = {}

            def _busy_all_reduce(t, *a, **k):
                w = _real2(t, *a, **k)
                if k.get("async_op"):
                    buf = _scratch.get(t.numel())
                    if buf is None:
                        buf = _scratch[t.numel()] = torch.empty_like(t)
                    for _ in range(busy):
                        w = dist.all_gather_into_tensor(buf, t, async_op=True)
                return w
            dist.all_reduce = _busy_all_reduce
        pkg.parallel.broadcast_parameters(model)
    trace("model built, parameters broadcast")
    ops.manual_seed(4321 + rank)

    # synthetic batches resident in HBM: x, y ~ U[0,1), distinct per rank and per pool slot
    pool = []
    nquads = (B * 3 * S * S + 3) // 4
    for i in range(4):
        base = ((rank * 4 + i) * 2) * nquads
        x = ops.to_nhwc(ops.rand_uniform((B, 3, S, S), dev, seed=1234, offset=base))
        y = ops.to_nhwc(ops.rand_uniform((B, 3, S, S), dev, seed=1234, offset=base + nquads))
        pool.append({"x": x, "y": x if wl in SAME_XY else y})

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # VCG_MAIN_PRIORITY=high: the step's own stream gets HIP's high priority (the weight-gradient side stream keeps the normal one),
    # so that where both have workgroups ready the critical path's go first (A/B measurement; default: the caller's stream as is)
    main_stream = None
    if os.environ.get("VCG_MAIN_PRIORITY") == "high":
        main_stream = torch.cuda.Stream(device=dev, priority=-1)
        main_stream.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(main_stream)
    for i in range(args.warmup):
        model.training_step(pool[i % len(pool)])
        trace(f"warm-up step {i} issued")
    # a full collection of Python's cyclic GC walks every long-lived object (modules, parameters, ctypes tables): ~30 ms
    # on the host, and since each step ends with a metric read-back the GPU idles for all of it.  Collect once now and
    # move the survivors out of the collector's way; the GC stays on for what the steps allocate.
    gc.collect()
    gc.freeze()
    barrier()
    if red is not None:
        red.exposed_ms()                       # drop the warm-up's events
        red.stats["buckets_from_backward"] = red.stats["buckets_at_start"] = 0
        red.wait_log.clear()
        red.log.clear()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = model.training_step(pool[i % len(pool)])
        trace(f"timed step {i} done")
    barrier()
    elapsed = time.perf_counter() - t0
    exposed_ms = None
    if dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
        # what the compute stream still waited for in GradReducer.finish() (HIP events around the waits), per step, and where
        # the buckets were launched from; the max over ranks, like the time
        ex = torch.tensor([red.exposed_ms() / args.steps], dtype=torch.float64, device=dev)
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        exposed_ms = ex.item()
    bad = [k for k, v in last.items() if v != v or abs(v) == float("inf")]
    if bad:
        raise SystemExit(f"non-finite metrics after the timed steps: {bad}")
    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = B * world * args.steps / elapsed

    out = {
        "metric": "train images/sec (256x256 VAE-CycleGAN step)" if wl == "cyclevaegan" else f"train images/sec ({wl} step)",
        "value": round(images_per_s, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "arithmetic": "fp32 in, fp32 out, fp32 accumulate; GEMM products as 2 x fp16 split operands scaled by a per-tensor power of two "
                      "(3 fp16 MFMAs per fp32 multiply-add), rounding error 1.3-4.1e-7 against float64 where PyTorch-CPU fp32 has 1.0-10e-7 "
                      "(profiles/r04_conv_accuracy.txt)",
        "config": {"workload": WORKLOADS[wl] if (S == 256) else f"{wl} {S}x{S} batch {B}", "per_gpu_batch": B, "global_batch": B * world,
                   "image_size": S, "latent_dim": latent, "parallelism": f"dp{world}", "init": "random (reference init statistics)"},
    }

    if dist_on:
        st = red.stats
        out["exchange_exposed_ms"] = round(exposed_ms, 3)
        # what the process group actually saw: the first SCALE record must show that RCCL ran over N ranks on N devices
        devs = torch.tensor([float(torch.cuda.current_device())], dtype=torch.float64, device=dev)
        dlist = [torch.zeros_like(devs) for _ in range(world)]
        dist.all_gather(dlist, devs)
        out["process_group"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank0_device_count": torch.cuda.device_count(),
                                "device_of_rank": [int(t.item()) for t in dlist],
                                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                                "VCG_BUCKET_MB": os.environ.get("VCG_BUCKET_MB", "128 (default)")}
        out["exchange"] = {"buckets_launched_from_inside_backward": st["buckets_from_backward"], "buckets_launched_after_backward": st["buckets_at_start"],
                           "buckets_ordered_after_a_second_stream": sum(1 for _, _, w in red.wait_log if w),
                           "bucket_bytes": red.bucket_elems * 4, "is": "sum-all-reduce (RCCL) of contiguous slices of each optimizer's flat "
                           "gradient buffer, launched as their last weight gradient is issued; exchange_exposed_ms = time the compute stream "
                           "spent waiting for them before the optimizer steps, per step, max over ranks"}
    if not args.no_roofline:
        # every rank runs the two extra (untimed) steps: with N > 1 they contain the gradient exchange, which all
        # ranks must enter; rank 0 reports its own kernels
        roof = measure_roofline(pkg, model, pool, wl, B, S, latent, ms_per_step, rank)
        if rank == 0:
            out.update(roof)
    if rank == 0 and not dist_on and not args.no_cpu_baseline and wl in ("cyclevaegan", "vae", "autoencoder"):
        out["cpu_baseline"] = cpu_baseline(pkg, wl, S, latent)
    if rank == 0 and not dist_on and not args.no_eager_baseline and wl in ("cyclevaegan", "vae", "autoencoder"):
        del model, pool
        gc.unfreeze()
        gc.collect()
        torch.cuda.empty_cache()
        out["rocm_eager_baseline"] = rocm_eager_baseline(wl, B, S, latent)
    if dist_on:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


PROFILE_ROUND = "r04"          # profiles/<round>_pmc_*.json: the committed PMC passes `traffic` and `mfma_pipe_util` are quoted from


def read_kernel_profile(lib):
    """{device kernel: (launches, seconds, executed FLOPs)} since vcg_profile_enable(1): HIP events recorded by the library
    around each MFMA kernel launch, on the stream the kernel was launched on."""
    import ctypes
    n = lib.vcg_profile_read(None, 0)
    buf = ctypes.create_string_buffer(max(int(n), 1) + 64)
    lib.vcg_profile_read(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms, fl = line.split("\t")
        c0, s0, f0 = out.get(name, (0, 0.0, 0.0))
        out[name] = (c0 + int(cnt), s0 + float(ms) * 1e-3, f0 + float(fl))
    return out


def measure_roofline(pkg, model, pool, wl, B, S, latent, ms_per_step, rank=0):
    """Two extra (untimed) steps on ONE stream with HIP events around every launch: per C-ABI call from Python (kernel
    families, algorithmic direct-convolution FLOPs) and per device kernel inside the library (the FLOPs that launch
    executes).  `roofline` describes the dominant DEVICE kernel against the pipe it runs on."""
    ops = pkg.ops
    lib = pkg._native.lib()
    ops.PROFILE = []
    read_kernel_profile(lib)                                      # drop anything recorded earlier
    lib.vcg_profile_enable(1)
    overlap, ops.OVERLAP_ENABLED = ops.OVERLAP_ENABLED, False     # kernel rates are measured one family at a time:
    nsteps = 2
    for i in range(nsteps):                                       # the timed steps above run the weight gradients on a
        model.training_step(pool[i % len(pool)])                  # side stream, where co-running families stretch each other
    torch.cuda.synchronize()
    lib.vcg_profile_enable(0)
    ops.OVERLAP_ENABLED = overlap
    recs, ops.PROFILE = ops.PROFILE, None
    dev = read_kernel_profile(lib)
    fam, shapes = {}, {}
    for name, flops, e0, e1, tag in recs:
        dt = e0.elapsed_time(e1) * 1e-3
        f = fam.setdefault(name, [0.0, 0.0, 0])
        f[0] += flops
        f[1] += dt
        f[2] += 1
        if tag:
            s = shapes.setdefault((name, tag), [0.0, 0.0, 0])
            s[0] += flops
            s[1] += dt
            s[2] += 1
    dump = os.environ.get("VCG_BENCH_SHAPES") if rank == 0 else None
    if dump:                                   # per-shape table for kernel work (not part of the JSON line)
        with open(dump, "w") as fh:
            for (name, tag), (fl, sec, cnt) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                fh.write(f"{name:11s} {tag:34s} n={cnt // nsteps:3d} ms/step={sec / nsteps * 1e3:8.3f} us/launch={sec / cnt * 1e6:9.1f} "
                         f"TF={fl / sec / 1e12:7.2f}\n")
            for name, (cnt, sec, fl) in sorted(dev.items(), key=lambda kv: -kv[1][1]):
                fh.write(f"device {name:28s} n={cnt // nsteps:3d} ms/step={sec / nsteps * 1e3:8.3f} us/launch={sec / cnt * 1e6:9.1f} "
                         f"executed TF={fl / sec / 1e12:7.2f}\n")
    conv = {k: v for k, v in fam.items() if k.startswith("conv_")}
    kernels = {k: {"launches_per_step": v[2] // nsteps, "ms_per_step": round(v[1] / nsteps * 1e3, 3),
                   "tflops": round(v[0] / max(v[1], 1e-12) / 1e12, 2) if v[0] else None} for k, v in sorted(fam.items())}
    # necessary conv FLOPs of one step: fwd + bwd-data + bwd-weight, nothing redundant
    step_flops = sum(v[0] for v in conv.values()) / nsteps

    def pipe_peak(name):              # the dense peak of the matrix pipe a device kernel issues to, in fp32-equivalent FLOPs
        return FP32_MFMA_PEAK_TFLOPS if "fp32 MFMA" in name else BF16X3_FP32_EQUIV_PEAK_TFLOPS
    device = {}
    for name, (cnt, sec, fl) in dev.items():
        tf = fl / max(sec, 1e-12) / 1e12
        device[name] = {"launches_per_step": cnt // nsteps, "ms_per_step": round(sec / nsteps * 1e3, 3),
                        "avg_launch_ms": round(sec / cnt * 1e3, 4), "executed_tflops": round(tf, 2),
                        "frac_of_pipe_peak": round(tf / pipe_peak(name), 4)}
    dom = max(dev, key=lambda k: dev[k][1])
    cnt, sec, fl = dev[dom]
    ach = fl / sec / 1e12
    headline = wl == "cyclevaegan" and B == 8 and S == 256
    traffic = profiled_traffic(dom) if headline else None
    mfma_util = profiled_mfma_util() if headline else None
    famdom = max(conv, key=lambda k: conv[k][1])
    ffl, fsec, fn = conv[famdom]
    fach = ffl / fsec / 1e12
    roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": round(pipe_peak(dom), 1), "unit": "TFLOP/s",
            "frac": round(ach / pipe_peak(dom), 4),
            "peak_is": "dense fp16 MFMA peak (2516.8 TFLOP/s, v_mfma_f32_32x32x16_f16) / 3: the pipe this kernel executes on, in "
                       "fp32-equivalent FLOPs — every fp32 product is three fp16 MFMAs (2-way operand split with a per-tensor "
                       "power-of-two scale, fp32 accumulate, fp32-level rounding; rounds 1-2 issued six bf16 MFMAs per product, "
                       "peak 419.5)" if pipe_peak(dom) != FP32_MFMA_PEAK_TFLOPS else "fp32 MFMA (v_mfma_f32_32x32x2_f32)",
            "flops_counted": "the FLOPs the launch executes: 2*M*N*K of its GEMM (for a Winograd layer the 16 transformed GEMMs, "
                             "2.25x fewer than the direct convolution's)",
            "launches": cnt // nsteps, "avg_launch_ms": round(sec / cnt * 1e3, 4),
            "measured": "HIP events recorded by libvcg around each launch of this device kernel, on its launch stream, in two extra "
                        "steps run on ONE stream (the timed steps overlap weight gradients with data gradients on a second stream)",
            "frac_vs_fp32_mfma_target": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
            "frac_vs_fp32_mfma_target_is": "the same rate against the 157.3 TFLOP/s fp32-MFMA roof BASELINE.json's north_star names "
                                           "(target >= 0.40)",
            "traffic": None if traffic is None else traffic["bytes_per_launch"],
            "traffic_detail": traffic, "mfma_pipe_util": mfma_util,
            "device_kernels": device,
            "family": {"kernel": famdom, "achieved_algorithmic_tflops": round(fach, 2),
                       "is": "the C-ABI call family with the most time (one call = several device kernels: transforms, GEMM, reduces); "
                             "FLOPs = the direct convolution's 2*M*Cout*K per call, the layer's algorithmic work",
                       "frac_of_split_pipe_peak": round(fach / BF16X3_FP32_EQUIV_PEAK_TFLOPS, 4),
                       "frac_vs_fp32_mfma_target": round(fach / FP32_MFMA_PEAK_TFLOPS, 4),
                       "launches": fn // nsteps, "avg_launch_ms": round(fsec / fn * 1e3, 4)}}
    if traffic is not None:
        # which roof is nearer for this kernel: its fabric bytes per second against HBM3E's 8 TB/s, or its MFMA rate
        hbm_frac = traffic["bytes_per_launch"] / (sec / cnt) / 8.0e12
        roof["hbm_frac_from_profiled_traffic"] = round(hbm_frac, 4)
        roof["bound"] = "hbm" if hbm_frac > roof["frac"] else "mfma"
    return {
        "roofline": roof,
        "step_conv_tflops": round(step_flops / (ms_per_step * 1e-3) / 1e12, 2),
        "step_conv_frac_of_fp32_mfma_target": round(step_flops / (ms_per_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
        "kernels": kernels,
    }


def _profile_json(stem):
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{stem}.json")
    try:
        with open(path) as fh:
            return json.load(fh), os.path.relpath(path, ROOT)
    except (OSError, ValueError):
        return None, None


def profiled_mfma_util():
    """MFMA-pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles)) of every MFMA kernel of the step, from the
    committed PMC pass over this command (tools/final_profiles.sh, tools/pmc_mfma_util.py).  NOT measured in this run
    (counters need the profiler): the record says so and names the commit the pass was taken at.  None when the file is missing."""
    js, path = _profile_json("pmc_mfma_util")
    if js is None:
        return None
    return {"measured_in_this_run": False, "source": path, "head": js.get("head", "unknown"),
            "kernels": {k: round(v["mfma_util"], 4) for k, v in js["kernels"].items()},
            "all_mfma_kernels_of_the_step": round(js["conv_kernels_mfma_util"], 4),
            "is": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over bench.py, one stream"}


def profiled_traffic(kernel):
    """Fabric-side bytes per launch of one device kernel from the committed rocprofv3 PMC passes over this command at the
    headline config (tools/final_profiles.sh, tools/pmc_summary.py: separate FETCH_SIZE / WRITE_SIZE passes, read = 2 x
    FETCH_SIZE on gfx950, Infinity-Cache hits included).  NOT measured in this run; None when the file or kernel is missing."""
    js, path = _profile_json("pmc_step_traffic")
    if js is None:
        return None
    # (the PMC pass names a kernel with its template arguments, the in-library profile scope without: `k_gemm_planes_dma<true, 16>`)
    names = [n for n in js.get("kernels", {}) if n == kernel or n.startswith(kernel + "<")]
    if not names:
        return None
    k = {f: sum(js["kernels"][n][f] for n in names) for f in ("read", "write", "launches")}
    return {"measured_in_this_run": False, "source": path, "head": js.get("head", "unknown"),
            "bytes_per_launch": round((k["read"] + k["write"]) / max(k["launches"], 1e-9)),
            "read_bytes_per_launch": round(k["read"] / max(k["launches"], 1e-9)),
            "write_bytes_per_launch": round(k["write"] / max(k["launches"], 1e-9)),
            "step_total_bytes": round(js["all"]["read"] + js["all"]["write"])}


def cpu_baseline(pkg, wl, S, latent):
    """The oracle's CPU restatement of the same step on this box's host cores: batch 2, one untimed step then
    two timed ones — a bounded ~10-30 s sample of the workload (the full batch-8 step takes ~40 s on 8 cores)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    oracle = importlib.import_module("vcg_oracle")
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    Nn = pkg.Networks
    torch.manual_seed(1234)
    if wl == "cyclevaegan":
        model = Nn.CycleVAEGAN(latent_dim=latent, paired=False)
    elif wl == "vae":
        model = Nn.VariationalAutoencoder(latent_dim=latent)
    else:
        model = Nn.Autoencoder()
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    del model
    b = 2
    x, y = (torch.from_numpy(a) for a in pkg.synth.batch(b, S, 1234))
    times = []
    state = {}
    for it in range(3):
        t0 = time.perf_counter()
        if wl == "cyclevaegan":
            eps = [torch.from_numpy(e) for e in pkg.synth.eps_list(6, (b, latent, S // 16, S // 16), 4321, step=it)]
            oracle.cyclevaegan_step(P, state, x, y, eps, 2e-4, False)
        elif wl == "vae":
            eps = torch.from_numpy(pkg.synth.eps_list(1, (b, latent, S // 16, S // 16), 4321, step=it)[0])
            oracle.vae_step(P, state, x, x, eps, 2e-4)
        else:
            oracle.autoencoder_step(P, state, x, x, 2e-4)
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / len(times[1:])
    ref = {"cyclevaegan": (0.200, 8), "vae": (0.89, 16), "autoencoder": (1.39, 16)}[wl]
    return {"value": round(b / t, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (functional PyTorch fp32 CPU restatement) {wl} step, batch {b}, {S}x{S}, mean of steps 2-3 of 3 "
                      f"({sum(times):.1f} s of CPU work in all)",
            # the reference's own module cannot travel to this box; its timing was taken where it runs
            "reference_in_build_container": {"value": ref[0], "unit": "images/s", "cores": 8, "kind": "reference", "batch": ref[1],
                                             "sample": f"/root/reference {wl} training_step, batch {ref[1]}, {S}x{S}, torch.set_num_threads(8), "
                                                       "median of 3 steps after warm-up, taken in the 8-core build container "
                                                       "(BASELINE.md §2); not re-measured in this run"}}


def rocm_eager_baseline(wl, B, S, latent, timeout=240):
    """`_eager_baseline_run` in a CHILD process (this script with --eager-child), after this process has measured and freed its own
    model: whatever stock MIOpen / ATen do on this box — a fault, an exhaustive search that never ends — the measured line above
    is already complete and is printed regardless.  (A child process, not an exec: the parent has initialised the GPU.)"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--eager-child", "--workload", wl, "--batch", str(B), "--size", str(S), "--latent", str(latent)]
    env = dict(os.environ)
    env.setdefault("MIOPEN_FIND_MODE", "FAST")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired:
        return {"value": None, "error": f"the stock-PyTorch child did not finish within {timeout} s"}
    for line in reversed(res.stdout.splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                break
    return {"value": None, "error": f"child exited with {res.returncode}: {(res.stderr or res.stdout)[-300:]}"}


def _eager_baseline_run(pkg, wl, B, S, latent, dev):
    """The same step on the same GPU through stock PyTorch-ROCm eager ops (MIOpen convolutions, ATen instance_norm / losses,
    autograd): the oracle's functional restatement — the reference's algorithm as written, six VAE forwards and the
    discriminator re-forward included — with its tensors on cuda:0 at the bench's batch size.  Outside the timed region, after
    the hand-written path has been measured and its model freed; a baseline beside cpu_baseline, never a fallback: nothing in the
    product can reach it.  One untimed step (MIOpen picks its algorithms there) and two timed ones."""
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")          # the default exhaustive find takes minutes for ~40 conv shapes
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    oracle = importlib.import_module("vcg_oracle")
    Nn = pkg.Networks
    torch.manual_seed(1234)
    if wl == "cyclevaegan":
        model = Nn.CycleVAEGAN(latent_dim=latent, paired=False)
    elif wl == "vae":
        model = Nn.VariationalAutoencoder(latent_dim=latent)
    else:
        model = Nn.Autoencoder()
    P = {k: v.detach().clone().to(dev) for k, v in model.state_dict().items()}
    del model
    x = pkg.ops.rand_uniform((B, 3, S, S), dev, seed=1234, offset=0)
    y = pkg.ops.rand_uniform((B, 3, S, S), dev, seed=1234, offset=1 << 22)
    times, state = [], {}
    try:
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if wl == "cyclevaegan":
                eps = [torch.randn((B, latent, S // 16, S // 16), device=dev) for _ in range(6)]
                oracle.cyclevaegan_step(P, state, x, y, eps, 2e-4, False)
            elif wl == "vae":
                oracle.vae_step(P, state, x, x, torch.randn((B, latent, S // 16, S // 16), device=dev), 2e-4)
            else:
                oracle.autoencoder_step(P, state, x, x, 2e-4)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    except Exception as e:  # noqa: BLE001 — a baseline that cannot run must not take the measured line with it
        return {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
    t = sum(times[1:]) / len(times[1:])
    return {"value": round(B / t, 2), "unit": "images/s", "ms_per_step": round(t * 1e3, 2), "kind": "stock PyTorch-ROCm eager ops",
            "torch": torch.__version__, "batch": B,
            "sample": f"oracle/vcg_oracle.py {wl} step as the reference writes it (fp32, MIOpen / ATen kernels, autograd) on cuda:0, batch {B}, "
                      f"{S}x{S}; mean of steps 2-3 of 3 (first step {times[0]:.1f} s: MIOpen algorithm search, MIOPEN_FIND_MODE="
                      f"{os.environ.get('MIOPEN_FIND_MODE')})"}


if __name__ == "__main__":
    main()
