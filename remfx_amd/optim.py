"""Flat-buffer optimiser for the training path (models.py:185-206, cfg/config.yaml:119).

All parameters of the network are re-seated as views into ONE contiguous fp32 buffer and
their .grad tensors as views into ONE contiguous gradient buffer.  That makes
  * the AdamW step one HIP launch over 6..84 M elements (HBM-bound, 16 B/elt streams),
  * global grad-norm clipping one reduction + a device-side coefficient (no host sync),
  * the data-parallel gradient exchange a handful of large RCCL all-reduces over xGMI
    on slices of the same buffer (remfx_amd/ddp.py) -- no per-tensor collectives.
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import check
from .ops import _ptr, _stream


class FlatParams:
    def __init__(self, params, allow_cpu=False, layout=None):
        """`params` keeps its order as the optimiser's parameter list (torch.optim state_dict indices, checkpoints).  `layout`
        (optional): the same parameters in the order they should lie in MEMORY -- the order the forward pass uses them in, so that the
        flat gradient buffer fills from its high end downwards during backward and ddp.GradSync's buckets (contiguous slices, taken
        from the high end) complete one after the other instead of all at the end (a network whose registration order is not its
        execution order: Hybrid Demucs registers all frequency layers before the time layers they interleave with)."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda" and not allow_cpu:   # allow_cpu: collective-plumbing tests only (gloo)
            raise ValueError("FlatParams needs parameters on the GPU (no CPU fallback)")
        order = list(range(len(self.params)))
        if layout is not None:
            pos = {id(p): i for i, p in enumerate(self.params)}
            order = [pos[id(p)] for p in layout if id(p) in pos]
            if sorted(order) != list(range(len(self.params))):
                raise ValueError("FlatParams(layout=...): must name every trainable parameter exactly once")
        self.offsets, n = [0] * len(self.params), 0
        for i in order:
            self.offsets[i] = n
            n += (self.params[i].numel() + 3) // 4 * 4     # keep every view 16-byte aligned
        self.numel = n
        self.data = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            v = self.data[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            p.grad = self.grad[o:o + p.numel()].view_as(p)
        from . import ops
        import weakref
        # the pack cache pins the weight tensors it packed from (these views): drop its entries when this buffer goes away, or a
        # discarded model's parameters and packed weights stay resident
        weakref.finalize(self, ops.clear_pack_cache)
        self.sink = None                                                   # see ops.GradSink
        if dev.type == "cuda" and ops.GradSink.MODE != "off":
            self.sink = ops.GradSink(self, side_stream=ops.GradSink.MODE == "side")

    def zero_grad(self):
        """One fill of the flat gradient buffer; arms the gradient sink: until join() the backward kernels accumulate
        parameter gradients straight into this buffer (weight-gradient GEMMs on the sink's side stream)."""
        self.join()
        if self.grad.is_cuda:
            from . import ops
            ops.zero_(self.grad)
        else:
            self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # re-seat in case something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)
        if self.sink is not None:
            from . import ops
            self.sink.writes = [0] * len(self.params)
            ops.SINK = self.sink

    def join(self):
        """Make the current stream wait for every gradient the sink still has in flight and disarm it.  Called by the
        optimiser step and the gradient exchange; call it before reading `.grad` by hand after a backward pass."""
        if self.sink is not None:
            from . import ops
            self.sink.join()
            if ops.SINK is self.sink:
                ops.SINK = None


class FlatAdamW:
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction) on FlatParams."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.step_count = 0
        dev = flat.data.device
        self._sumsq = torch.zeros(1, device=dev, dtype=torch.float64)
        self._sumsq_ws = torch.empty(4096, device=dev, dtype=torch.float64)        # RFX_SUMSQ_SLOTS per-workgroup partials
        self._coef = torch.ones(1, device=dev, dtype=torch.float32)
        self.last_grad_norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.param_groups = [{"lr": lr}]                 # scheduler-facing, as torch optimisers

    def step(self, clip_norm=None, grad_prescale=1.0):
        """One update.  clip_norm: global L2 norm clip (Lightning gradient_clip_val);
        grad_prescale: factor already owed to the gradients (1/world_size after a SUM all-reduce)."""
        L, f = _lib.lib(), self.flat
        f.join()                                         # side-stream weight gradients land before the buffer is read
        self.step_count += 1
        gscale = None
        if clip_norm or grad_prescale != 1.0:
            check(L.rfx_sumsq(_ptr(f.grad), f.numel, _ptr(self._sumsq_ws), _ptr(self._sumsq), _stream()), "rfx_sumsq")
            check(L.rfx_clip_coef(_ptr(self._sumsq), float(clip_norm or 0.0), float(grad_prescale),
                                  _ptr(self._coef), _ptr(self.last_grad_norm), _stream()), "rfx_clip_coef")
            gscale = self._coef
        lr = self.param_groups[0]["lr"]
        check(L.rfx_adamw_step(_ptr(f.data), _ptr(f.grad), _ptr(self.m), _ptr(self.v), f.numel, lr,
                               self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count,
                               _ptr(gscale), _stream()), "rfx_adamw_step")
        ops.weights_changed()                            # the kernel wrote the parameters through a raw pointer: cached packs are stale

    def zero_grad(self):
        self.flat.zero_grad()

    def state_dict(self):
        """torch.optim.AdamW.state_dict() layout ({"state": {i: {step, exp_avg, exp_avg_sq}}, "param_groups": [...]}),
        the form Lightning stores under ckpt["optimizer_states"][0] (reference scripts/test.py:20-23 reads such files);
        the per-parameter tensors are views of the flat moment buffers."""
        f = self.flat
        state = {}
        for i, (p, o) in enumerate(zip(f.params, f.offsets)):
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.m[o:o + p.numel()].view_as(p), "exp_avg_sq": self.v[o:o + p.numel()].view_as(p)}
        group = {"lr": self.param_groups[0]["lr"], "betas": tuple(self.betas), "eps": self.eps,
                 "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(f.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "m" in sd:                                    # round-1 flat form
            self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
            self.step_count = int(sd["step"]); self.param_groups[0]["lr"] = float(sd["lr"])
            return
        f = self.flat
        # torch / Lightning key the state by the parameter's position in the optimiser's param list, which may hold parameters this
        # flat buffer does not (frozen ones: no state entry).  Entries are matched to the trainable parameters IN ORDER and by
        # SHAPE; anything that does not line up is an error, never a silent skip.
        ids = list(sd["param_groups"][0].get("params", range(len(f.params))))
        entries = [(i, sd["state"][i]) for i in ids if i in sd["state"]]
        if entries and len(entries) != len(f.params):
            raise ValueError(f"optimizer state holds {len(entries)} parameters with moments, the model has {len(f.params)} trainable ones")
        steps = set()
        for (i, st), (p, o) in zip(entries, zip(f.params, f.offsets)):
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state entry {i}: moment shape {tuple(st['exp_avg'].shape)} != parameter shape {tuple(p.shape)}")
            self.m[o:o + p.numel()].view_as(p).copy_(st["exp_avg"])
            self.v[o:o + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"optimizer state entries disagree on the step count: {sorted(steps)}")
        if steps:
            self.step_count = steps.pop()
        g = sd["param_groups"][0]
        self.param_groups[0]["lr"] = float(g["lr"])
        self.betas, self.eps, self.weight_decay = tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])


class MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma) as `.step()` evaluates it (the chainable
    form): the rate is multiplied by gamma ** count only on the step whose index EQUALS a milestone.  The reference
    passes the floats 0.8 * max_steps and 0.95 * max_steps (models.py:193-197): a non-integer milestone never fires,
    exactly as upstream.  Stepped per batch (interval "step")."""

    def __init__(self, optimizer, milestones, gamma=0.1):
        from collections import Counter
        self.opt, self.milestones, self.gamma = optimizer, Counter(milestones), gamma
        self.base_lr = optimizer.param_groups[0]["lr"]
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        if self.last_epoch in self.milestones:
            self.opt.param_groups[0]["lr"] *= self.gamma ** self.milestones[self.last_epoch]

    def state_dict(self):
        lr = self.opt.param_groups[0]["lr"]
        return {"milestones": self.milestones, "gamma": self.gamma, "base_lrs": [self.base_lr],
                "last_epoch": self.last_epoch, "_step_count": self.last_epoch + 1, "_last_lr": [lr]}

    def load_state_dict(self, sd):
        self.milestones, self.gamma = sd["milestones"], sd["gamma"]
        self.base_lr, self.last_epoch = sd["base_lrs"][0], int(sd["last_epoch"])
        if sd.get("_last_lr"):
            self.opt.param_groups[0]["lr"] = float(sd["_last_lr"][0])
