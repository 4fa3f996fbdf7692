"""Fused Adam over one flat fp32 parameter buffer.

Replaces torch.optim.Adam at the reference's call sites (Networks.py:312, 894, 1928-1935:
lr from --lr, betas (0.5, 0.999), eps 1e-8, no weight decay).  All parameters handed to
the optimizer are re-homed as views of a single contiguous buffer; `.grad` of each is a
view of a second buffer that the conv backward kernels accumulate into directly.  One
optimizer step is therefore ONE kernel launch reading/writing 28 B per parameter, and a
data-parallel gradient exchange is an all-reduce over contiguous slices of `flat_grad`.

`state_dict()` / `load_state_dict()` speak torch.optim.Adam's format (per-parameter
`step`, `exp_avg`, `exp_avg_sq`, one param group) so optimizer states move between the
reference and this implementation.
"""
import torch

from . import ops

_ALIGN = 4  # elements; keeps every view 16-byte aligned for float4 access


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        plist = []
        seen = set()
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                plist.append(p)
        if not plist:
            raise ValueError("optimizer got an empty parameter list")
        dev = plist[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam needs parameters on the GPU (call model.to('cuda') before configure_optimizers)")
        for p in plist:
            if p.device != dev or p.dtype != torch.float32:
                raise RuntimeError("FusedAdam: all parameters must be float32 on one device")
        self.params = plist
        self.offsets = []
        off = 0
        for p in plist:
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = off
        self.flat_param = torch.empty(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.empty(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.empty(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.empty(off, dtype=torch.float32, device=dev)
        for buf in (self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq):
            ops.fill_(buf, 0.0)
        with torch.no_grad():
            for p, o in zip(plist, self.offsets):
                view = self.flat_param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)          # one-time re-homing of the initial values
                p.data = view
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
        self.step_count = 0                     # steps taken by the parameters' common counter (max over `steps`)
        self.steps = [0] * len(plist)           # torch.optim.Adam keeps one counter per parameter; so do we
        self.grad_scale = 1.0
        self._epoch = [0]                       # bumped by step(); ConvSpec.packed() keys its cache on it
        for p in plist:
            p._vcg_epoch = self._epoch
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                             foreach=None, capturable=False, differentiable=False, fused=None,
                             decoupled_weight_decay=False)
        self.param_groups = [dict(self.defaults, params=plist)]
        ops.PARAM_EPOCH[0] += 1

    # -- torch.optim.Optimizer surface -------------------------------------------------
    def zero_grad(self, set_to_none=False):
        """Gradients are accumulation targets of the backward kernels, so they are zeroed, never dropped."""
        if getattr(self, "_exchange_pending", False):
            raise RuntimeError("zero_grad() while this optimizer's gradient exchange is in flight (parallel.GradReducer): "
                               "finish() the exchange and step() first")
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
        ops.fill_(self.flat_grad, 0.0)

    def step(self, grad_scale=None, repack=True):
        """`grad_scale` multiplies the gradient inside the fused launch (1/world after a summing
        all-reduce; parallel.GradReducer sets `self.grad_scale`).  `repack=False` leaves the side-stream repack of
        the conv weights to the caller (ops.repack_async), e.g. until no collective is in flight any more."""
        g = self.param_groups[0]
        scale = self.grad_scale if grad_scale is None else grad_scale
        self.steps = [t + 1 for t in self.steps]
        self.step_count = max(self.steps)
        # one launch per run of consecutive parameters that share a step counter: ONE launch unless a loaded state
        # carried different counters (a reference run that re-created its optimizer for part of the model)
        i = 0
        while i < len(self.params):
            j = i
            while j + 1 < len(self.params) and self.steps[j + 1] == self.steps[i]:
                j += 1
            lo = self.offsets[i]
            hi = self.offsets[j + 1] if j + 1 < len(self.params) else self.total
            ops.adam_step_flat(self.flat_param[lo:hi], self.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                               self.steps[i], g["lr"], g["betas"][0], g["betas"][1], g["eps"], scale)
            i = j + 1
        self._epoch[0] += 1
        if repack:
            ops.repack_async(self.params)

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                if self.steps[i] == 0:
                    continue                      # torch creates a parameter's state at its first step
                state[i] = {
                    "step": torch.tensor(float(self.steps[i])),
                    "exp_avg": self.exp_avg[o:o + p.numel()].view(p.shape).clone(),
                    "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view(p.shape).clone(),
                }
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """torch.optim.Adam's state_dict (ours or the reference's).  The parameter list must be the one the state was
        saved for: a reference checkpoint written after `configure_optimizers(decoder_only=True)` (Networks.py:307-313)
        holds the decoder's tensors only and loads into an optimizer configured the same way.  Per-parameter step
        counters are kept as they are (they may differ, and a parameter without state starts at step 0)."""
        groups = sd["param_groups"]
        n_saved = sum(len(g["params"]) for g in groups)
        if n_saved != len(self.params):
            raise ValueError(f"optimizer state holds {n_saved} parameter(s), this optimizer {len(self.params)}: the state was "
                             "saved for another parameter list (e.g. a reference Autoencoder run with "
                             "configure_optimizers(decoder_only=True) — configure this model the same way before loading)")
        if len(groups) != 1:
            hyper = [(g.get("lr"), tuple(g.get("betas", ())), g.get("eps"), g.get("weight_decay", 0)) for g in groups]
            if len(set(hyper)) != 1:
                raise ValueError("FusedAdam runs one hyper-parameter set over its flat buffer; the loaded state has "
                                 f"{len(groups)} param groups with different lr / betas / eps")
        if groups[0].get("weight_decay", 0) or groups[0].get("amsgrad", False) or groups[0].get("maximize", False):
            raise ValueError("FusedAdam implements plain Adam (no weight decay / amsgrad / maximize), as the reference uses it")
        for k in ("lr", "betas", "eps"):
            if k in groups[0]:
                self.param_groups[0][k] = tuple(groups[0][k]) if k == "betas" else groups[0][k]
        order = [i for g in groups for i in g["params"]]           # saved index of our i-th parameter
        steps = [0] * len(self.params)
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                st = sd["state"].get(order[i], sd["state"].get(str(order[i])))
                if st is None:
                    ops.fill_(self.exp_avg[o:o + p.numel()], 0.0)
                    ops.fill_(self.exp_avg_sq[o:o + p.numel()], 0.0)
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state of parameter {i} has shape {tuple(st['exp_avg'].shape)}, expected {tuple(p.shape)}")
                self.exp_avg[o:o + p.numel()].view(p.shape).copy_(st["exp_avg"])
                self.exp_avg_sq[o:o + p.numel()].view(p.shape).copy_(st["exp_avg_sq"])
                t = float(st["step"])
                if abs(t - round(t)) > 1e-3 or t < 0:
                    raise ValueError(f"optimizer state of parameter {i} has a non-integral step counter {t!r}")
                steps[i] = int(round(t))
        self.steps = steps
        self.step_count = max(steps) if steps else 0
