"""Plain data parallelism for the training step: one process per GPU, RCCL over xGMI.

The reference is single-process; this is the new piece the north_star asks for.
Every op of the step is per-sample independent (InstanceNorm, one discriminator scalar per image)
and every loss is a batch mean, so averaging per-rank gradients over equal shards IS the
big-batch gradient.  The only exchange is therefore one sum-all-reduce per optimizer, cut into
buckets that are launched FROM INSIDE the backward pass as they complete:

  * gradients already sit in ONE flat fp32 buffer per optimizer (optim.FusedAdam), in parameter order, and the
    backward kernels accumulate into it directly.  A bucket is a run of whole parameters (>= `bucket_bytes`), i.e. a
    contiguous slice: no flatten / unflatten copies;
  * the backward reports every weight gradient it has issued (`ops.GRAD_READY_HOOK` -> `note`).  The first step
    learns how many reports each parameter receives per backward (G's parameters are used by two forwards of the
    step, the shared encoder of the Double models too); from the second step on a bucket is all-reduced the moment
    its last report arrives — F's gradients while G(x) is still being differentiated, G's decoder while its encoder
    is — and only the tail of the exchange is left for the discriminator backward to hide;
  * the collective is ordered after EVERY stream that reported a gradient of the bucket (the side stream of
    `ops.wgrad_overlap`, and the main stream for bias gradients and the full-map layer that ends a discriminator): it is
    issued under a launch stream of its own that has been made to wait for those streams (`_order_after`), so RCCL's
    stream waits for exactly the kernels that wrote the slice and neither the data-gradient chain on the main stream
    nor the weight-gradient stream ever waits for the other;
  * the 1/world scaling is folded into the fused Adam launch (`grad_scale`): no extra pass over the gradients;
  * gradients the generator phase deposits on the discriminators as a by-product are never produced here
    (`ops.no_wgrad`), so nothing spurious is reduced.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are large (default 128 MiB) so that every RCCL launch can
keep all links busy.  `stats` accumulates what the step still waits for (HIP events around `finish`): bench.py prints it
as `exchange_exposed_ms` for N > 1.
"""
import os

import torch
import torch.distributed as dist


# VCG_DP_FROM_BACKWARD=0: every bucket is exchanged after the backward (at `start`), none from inside it — the reference point of
# tests/test_gpu_parity.py::test_two_rank_step_equals_the_big_batch_step, which requires the in-backward launches to change no bit
FROM_BACKWARD = os.environ.get("VCG_DP_FROM_BACKWARD", "1") != "0"
USE_LAUNCH_STREAM = os.environ.get("VCG_DP_LAUNCH_STREAM", "1") != "0"       # 0: collectives issued under the last reporter's stream (A/B)


def default_bucket_bytes():
    """VCG_BUCKET_MB: size at which a gradient bucket closes.  Default 128 MiB: xGMI is point-to-point (few large collectives),
    and every collective issued from inside the backward costs the step 0.2 - 0.3 ms whatever its size (one-rank RCCL runs of
    the bench step, profiles/r04_dp_one_rank.txt: 64 MiB = 8 launches +1.9 ms, 128 MiB = 5 launches +1.4 ms over the step
    without a process group)."""
    return max(1, int(float(os.environ.get("VCG_BUCKET_MB", "128")) * (1 << 20)))


class _Plan:
    """Buckets of one optimizer: runs of whole parameters of its flat gradient buffer."""

    def __init__(self, optimizer, bucket_elems):
        params = getattr(optimizer, "params", None)
        total = optimizer.flat_grad.numel()
        self.buckets = []                        # (lo, hi) element ranges
        self.bucket_of = {}                      # id(param) -> bucket index
        if params:
            offs = list(optimizer.offsets) + [total]
            lo, members = 0, []
            for i, p in enumerate(params):
                members.append(p)
                if offs[i + 1] - lo >= bucket_elems or i + 1 == len(params):
                    for q in members:
                        self.bucket_of[id(q)] = len(self.buckets)
                    self.buckets.append((lo, offs[i + 1]))
                    lo, members = offs[i + 1], []
        else:                                    # a bare flat buffer (tests): fixed-size slices
            for lo in range(0, total, bucket_elems):
                self.buckets.append((lo, min(total, lo + bucket_elems)))
        self.expected = None                     # id(param) -> reports per backward, learned in the first one
        self.learning = {}
        self.remaining = None                    # per bucket: reports still to come in this backward
        self.launched = set()
        self.streams = [[] for _ in self.buckets]  # per bucket: the distinct streams that reported into it in this backward


class GradReducer:
    def __init__(self, group=None, bucket_bytes=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket_elems = max(1, (bucket_bytes if bucket_bytes is not None else default_bucket_bytes()) // 4)
        self._plans = {}
        self._pending = {}
        self._armed = None                       # the optimizer whose backward is running
        self._lstream = None                     # see _launch_stream
        self.log = []                            # (tag, bucket, lo, hi, "backward" | "start") in launch order
        self.wait_log = []                       # (bucket, launch stream, [other producer streams it was made to wait for])
        self.stats = {"exposed_ms_events": [], "buckets_from_backward": 0, "buckets_at_start": 0}

    def _plan(self, optimizer):
        pl = self._plans.get(id(optimizer))
        if pl is None or pl.total != optimizer.flat_grad.numel():
            pl = _Plan(optimizer, self.bucket_elems)
            pl.total = optimizer.flat_grad.numel()
            self._plans[id(optimizer)] = pl
        return pl

    # ---- called by the model's training_step -------------------------------------------------------------------
    def begin(self, optimizer):
        """Right after `optimizer.zero_grad()`: arm the countdowns for the backward that fills its flat gradient."""
        if self._pending.get(id(optimizer)):
            raise RuntimeError("gradient exchange still pending on this optimizer: finish() it before the next backward")
        pl = self._plan(optimizer)
        pl.launched = set()
        pl.learning = {}
        pl.streams = [[] for _ in pl.buckets]
        if pl.expected is not None:
            pl.remaining = [0] * len(pl.buckets)
            pl.left = dict(pl.expected)
            for pid, n in pl.expected.items():
                pl.remaining[pl.bucket_of[pid]] += n
        self._armed = optimizer
        self._pending[id(optimizer)] = []
        optimizer.grad_scale = 1.0 / self.world
        optimizer._exchange_pending = True

    def note(self, wparam, bparam=None, stream=None):
        """A backward kernel sequence that accumulates into `wparam.grad` (and `bparam.grad`) has been issued on `stream`."""
        opt = self._armed
        if opt is None:
            return
        pl = self._plans[id(opt)]
        for prm in (wparam, bparam):             # the bias gradient comes out of the same call, and may sit in another bucket
            if prm is None or not getattr(prm, "requires_grad", True):
                continue
            pid = id(prm)
            if pid not in pl.bucket_of:
                continue                         # a parameter of another optimizer (e.g. D's data-gradient-only pass)
            if pl.expected is None:              # first backward: learn the counts, exchange everything at start()
                pl.learning[pid] = pl.learning.get(pid, 0) + 1
                continue
            b = pl.bucket_of[pid]
            if b in pl.launched or pl.left.get(pid, 0) <= 0:
                raise RuntimeError("a weight gradient was accumulated into a bucket whose exchange is already in flight "
                                   "(this backward uses a parameter more often than the first one did)")
            pl.left[pid] -= 1
            pl.remaining[b] -= 1
            if stream is not None and not any(s is stream or s == stream for s in pl.streams[b]):
                pl.streams[b].append(stream)
            if pl.remaining[b] == 0 and FROM_BACKWARD:
                self._launch(opt, pl, b, stream, "backward")

    def start(self, optimizer):
        """After the backward: exchange every bucket that has not been launched from inside it."""
        pl = self._plan(optimizer)
        if id(optimizer) not in self._pending:   # begin() was not called (plain use): everything goes now
            self._pending[id(optimizer)] = []
            pl.launched = set()
            optimizer.grad_scale = 1.0 / self.world
            optimizer._exchange_pending = True
        if self._armed is optimizer:
            self._armed = None
            if pl.expected is None and pl.learning:
                pl.expected = dict(pl.learning)
        for b in range(len(pl.buckets)):
            if b not in pl.launched:
                self._launch(optimizer, pl, b, None, "start")

    def _launch_stream(self, device):
        """The stream the collectives are issued under (one per reducer): RCCL orders its own stream after whatever stream is
        current at the call, so this one is made to wait for the producers of a bucket and nothing else ever waits for IT but
        RCCL.  Round 4: the collective used to be issued under the last reporter's stream after making THAT stream wait for
        the other producers — when the last report came from the main stream (a bias gradient out of the InstanceNorm
        backward, a discriminator's full-map layer) the data-gradient chain stood still until the weight-gradient stream had
        caught up, at every bucket: a one-rank RCCL run of the bench step measured 40.4 ms against 34.8 without a reducer."""
        if self._lstream is None:
            from . import ops
            self._lstream = ops.launch_stream(device)      # created early by ops.create_streams where the caller did that
        return self._lstream

    def _order_after(self, b, stream, others, is_cuda):
        """Order the collective of bucket b after EVERY stream that wrote into it during this backward: the last reporter's
        (`stream`) and the others — a discriminator's full-map layer reports from the main stream while its conv layers report
        from the side stream, and a bucket ordered after its last reporter alone would be reduced while another stream is
        still accumulating into it.  Returns the stream to issue the collective under."""
        waited = [s for s in others if not (s is stream or s == stream)]
        launch = stream
        if is_cuda and not USE_LAUNCH_STREAM:
            for s in waited:                     # round 3's form (diagnostic): the last reporter's stream waits for the others
                stream.wait_stream(s)
        elif is_cuda:
            launch = self._launch_stream(stream.device)
            launch.wait_stream(stream)
            for s in waited:
                launch.wait_stream(s)
        self.wait_log.append((b, stream, waited))
        return launch

    def _launch(self, optimizer, pl, b, stream, where):
        lo, hi = pl.buckets[b]
        flat = optimizer.flat_grad
        if stream is not None and flat.is_cuda:
            launch = self._order_after(b, stream, pl.streams[b], True)
            with torch.cuda.stream(launch):      # RCCL orders itself after the CURRENT stream, which follows every producer
                work = dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            if stream is not None:
                self._order_after(b, stream, pl.streams[b], False)
            if flat.is_cuda:
                from . import ops
                ops.join_side_streams()          # leftovers at start(): every producer stream first
            work = dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        pl.launched.add(b)
        self._pending[id(optimizer)].append(work)
        self.log.append((getattr(optimizer, "tag", ""), b, lo, hi, where))
        self.stats["buckets_from_backward" if where == "backward" else "buckets_at_start"] += 1

    def finish(self, optimizer):
        """Make the compute stream wait for that exchange (before optimizer.step())."""
        works = self._pending.pop(id(optimizer), [])
        timed = optimizer.flat_grad.is_cuda and works
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in works:
            w.wait()
        if timed:
            e1.record()
            self.stats["exposed_ms_events"].append((e0, e1))
        optimizer._exchange_pending = False

    def exposed_ms(self, reset=True):
        """Milliseconds the compute stream spent waiting in finish() since the last call (synchronises)."""
        if self.stats["exposed_ms_events"]:
            torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.stats["exposed_ms_events"])
        if reset:
            self.stats["exposed_ms_events"] = []
        return ms

    # ---- small collectives -----------------------------------------------------------------------------------------
    def average_metrics(self, vec):
        """Logged metrics are means over the global batch: average the per-rank scalars."""
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.group)
        vec.mul_(1.0 / self.world)          # 19 floats; bookkeeping, not on the step's critical path
        return vec

    def any_rank(self, flag):
        """True on every rank when `flag` is true on any: the decision to skip an update must be collective, or the ranks
        that skipped never join the gradient exchange the others entered."""
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32,
                         device="cuda" if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t.item() > 0)


def nccl_options():
    """`pg_options` for `init_process_group("nccl", ...)`.  VCG_NCCL_PRIORITY=high takes the collectives' stream from PyTorch's
    high-priority pool (separate hardware queues from every normal-priority stream); measured on a one-rank RCCL group it makes no
    difference to the step (34.3 ms either way against 32.5 without a process group, profiles/r04_dp_one_rank.txt), so the
    default stays PyTorch's; the knob is for the first multi-GPU measurements."""
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = os.environ.get("VCG_NCCL_PRIORITY", "normal") == "high"
    return opts


def broadcast_parameters(model, src=0, group=None):
    """Identical replicas (parameters and spectral-norm buffers).  The broadcast writes through `.data`, which bumps
    neither a tensor's version counter nor its optimizer's step epoch — the two things `ConvSpec.packed()` keys its cache
    on — so every weight pack is invalidated explicitly: a rank that had already run a forward (warm-up, validation, a
    checkpoint load) would otherwise keep convolving with its pre-broadcast weights."""
    from . import ops
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=src, group=group)
    ops.PARAM_EPOCH[0] += 1
    seen = set()
    for p in model.parameters():
        ep = getattr(p, "_vcg_epoch", None)
        if ep is not None and id(ep) not in seen:
            seen.add(id(ep))
            ep[0] += 1


def attach(model, group=None, bucket_bytes=None):
    """Give `model.training_step` a gradient exchange; returns the reducer."""
    from . import ops
    red = GradReducer(group, bucket_bytes)
    model.grad_reducer = red
    ops.GRAD_READY_HOOK[0] = red.note
    for name in ("optimizer", "optimizer_G", "optimizer_D"):
        opt = getattr(model, name, None)
        if opt is not None:
            opt.tag = name
    return red
