"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend
"nccl" == RCCL on ROCm) over xGMI.

The path shards over independent clips: the only data-path collective is
one SUM all-reduce of the flat gradient buffer per step.  xGMI is point-to-point
(7 links x ~153 GB/s per GPU), so instead of many small per-tensor collectives the
flat buffer (remfx_amd/optim.FlatParams) is cut into a few large buckets that are
reduced asynchronously as soon as backward has produced every gradient inside them
(deepest layers first), overlapping the remaining backward compute.  The 1/world_size
mean is folded into the optimiser's gradient scale (no extra pass over the buffer).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the launcher (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "RFX_FORCE_DEVICE" in os.environ:          # test hook: several ranks on one GPU (gloo only)
        local = int(os.environ["RFX_FORCE_DEVICE"])
    if world > 1 and not dist.is_initialized():
        backend = backend or os.environ.get("RFX_DIST_BACKEND")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def broadcast_parameters(flat_data, src=0):
    """Replicas start from rank 0's weights (and buffers passed in as extra tensors)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_data, src)
        from . import ops
        ops.weights_changed()


def broadcast_buffers(module, src=0):
    """Module buffers (BatchNorm running mean / var / counters of Cnn14, Open-Unmix, DCUNet) from rank 0: DDP
    broadcasts them at construction; parameters travel with the flat buffer above."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for b in module.buffers():
            if b.numel():
                dist.broadcast(b.data, src)


class GradSync:
    """Bucketed async all-reduce of a FlatParams gradient buffer, overlapped with backward."""

    def __init__(self, flat, bucket_mb=64.0, overlap=True):
        self.flat = flat
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.handles = []
        self._stall = []                # (event before the waits, event after) per step on the compute stream
        self.overlap = overlap and self.world > 1
        # buckets over the flat buffer in REVERSE parameter order (backward produces the last
        # layers' gradients first); boundaries fall on parameter boundaries
        limit = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets, self.bucket_of = [], {}
        hi = flat.numel
        cur_lo, cur_params = hi, []
        by_offset = sorted(range(len(flat.params)), key=lambda i: flat.offsets[i])      # memory order (FlatParams(layout=...)), high end first
        for k in range(len(by_offset) - 1, -1, -1):
            idx = by_offset[k]
            cur_lo = flat.offsets[idx]
            cur_params.append(idx)
            if hi - cur_lo >= limit or k == 0:
                b = len(self.buckets)
                self.buckets.append({"lo": cur_lo, "hi": hi, "n": len(cur_params), "ready": 0})
                for i in cur_params:
                    self.bucket_of[i] = b
                hi, cur_params = cur_lo, []
        # Gradients reach the flat buffer two ways: autograd's accumulation and the gradient sink (ops.GradSink: kernels accumulate in
        # place, once per USE of the parameter; the backward node then returns None).  The post-accumulate hook of a parameter runs
        # after ALL of its uses have been back-propagated -- on this torch also when every gradient handed to it was None (sink-only
        # parameters), which earlier versions did not promise.  So neither signal is trusted alone: the FIRST step only records, per
        # parameter, how many sink writes it received and whether its hook ran (no early launches, finish() reduces everything); from
        # then on a parameter is complete when it has been written that often AND (if its hook ran in the calibration step) the hook
        # has run, and a bucket is reduced as soon as all its parameters are complete.
        self.sink = getattr(flat, "sink", None)
        self.expected = None                        # sink writes per parameter in one step, learned in step 1
        self.hooked = None                          # did the parameter's post-accumulate hook run in step 1
        self._seen = [0] * len(flat.params)
        self._hook_seen = [False] * len(flat.params)
        self._counted = [False] * len(flat.params)  # a parameter enters its bucket's `ready` count once per step
        self._next = 0                              # first bucket whose all-reduce has not been issued in this step (_launch_in_order)
        self._late = False                          # a gradient write arrived after its bucket's all-reduce was queued
        self._late_info = []                        # (route, parameter index, expected sink writes, seen) of those writes, for the error text
        self.measure_stall = False                  # bench.py: time the compute stream's stall on the collectives (exposed_ms)
        if self.overlap:
            for idx, p in enumerate(flat.params):
                p.register_post_accumulate_grad_hook(self._make_hook(idx))
            if self.sink is not None:
                self.sink.on_write = self._sink_write

    def _precount(self):
        """Parameters the calibration step never touched (no sink write, no hook: unused in this model's forward) send no signal:
        they count as complete from the start of the step."""
        if self.expected is None:
            return
        for idx in range(len(self.flat.params)):
            if self.expected[idx] == 0 and not self.hooked[idx]:
                self._counted[idx] = True
                b = self.buckets[self.bucket_of[idx]]
                b["ready"] += 1
                if b["ready"] == b["n"]:
                    b["complete"] = True             # goes out when the buckets in front of it have (never at step start: the
                                                     # gradient buffer is zeroed after this bookkeeping)

    def _maybe_complete(self, idx):
        if self.expected is None or self._counted[idx]:
            return
        b = self.buckets[self.bucket_of[idx]]
        if b.get("hold") or self._seen[idx] < self.expected[idx] or (self.hooked[idx] and not self._hook_seen[idx]):
            return
        self._counted[idx] = True
        b["ready"] += 1
        if b["ready"] == b["n"]:
            b["complete"] = True
            self._launch_in_order()

    def _launch_in_order(self):
        """Collectives are issued in BUCKET INDEX order on every rank: a complete bucket goes out early only once all buckets in
        front of it have gone out, otherwise it waits (finish() sends the rest, in index order too).  A rank whose step deviates from
        the calibration step (a held or never-completed bucket) therefore still issues the same sequence of all-reduces over the same
        slices as the others -- it only issues them later."""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if not b.get("complete") or b.get("hold"):
                return
            self._launch(b)
            self._next += 1

    def _make_hook(self, idx):
        def hook(_p):
            self._hook_seen[idx] = True
            if self.sink is not None and self.flat.grad.is_cuda:
                self.sink.note_stream()              # autograd accumulated on the node's stream (the time branch has its own)
            if self.expected is None:
                return                               # calibration step: finish() launches everything
            b = self.buckets[self.bucket_of[idx]]
            if b.get("launched"):
                # a hook that did not run in the calibration step runs behind the queued all-reduce: autograd may have just
                # accumulated into the slice being reduced
                self._late = True
                self._late_info.append(("hook", idx, self.expected[idx], self._seen[idx]))
                return
            self._maybe_complete(idx)
        return hook

    def _sink_write(self, idx):
        self._seen[idx] += 1
        if self.expected is None:
            return                                   # calibration step
        b = self.buckets[self.bucket_of[idx]]
        if b.get("launched"):
            # the write lands behind the queued all-reduce and would stay a rank-local addition to the reduced slice
            self._late = True
            self._late_info.append(("sink", idx, self.expected[idx], self._seen[idx]))
        elif self._seen[idx] > self.expected[idx]:
            b["hold"] = True                         # more writes than the calibration step saw: this bucket waits for finish()
        else:
            self._maybe_complete(idx)

    def _launch(self, b):
        if b.get("launched"):
            return
        b["launched"] = True
        sl = self.flat.grad[b["lo"]:b["hi"]]
        side = self.sink.side if (self.sink is not None and sl.is_cuda) else None
        if side is not None:
            # the bucket's gradients come from the compute stream, the sink's side stream AND any other stream backward nodes ran on
            # (Hybrid Demucs' time branch): order the collective after all of them
            side.wait_stream(torch.cuda.current_stream())
            self.sink.order_after_writes(side)
            with torch.cuda.stream(side):
                self.handles.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True))
            self.sink.used_side = True
        else:
            self.handles.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Call after backward: launch whatever has not been reduced yet, wait, reset."""
        if self.world == 1:
            return 1.0
        for b in self.buckets:
            self._launch(b)           # whatever the hooks have not launched (parameters without a gradient never fire theirs), in index order
        self._next = 0
        timed = self.flat.grad.is_cuda and self.measure_stall
        if timed:                     # GPU time the compute stream spends stalled on the collectives = what backward did not hide
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in self.handles:
            h.wait()
        if timed:
            e1.record()
            self._stall.append((e0, e1))
        self.handles = []
        self.flat.join() if hasattr(self.flat, "join") else None
        for b in self.buckets:
            b["ready"] = 0
            b["launched"] = False
            b["hold"] = False
            b["complete"] = False
        self._counted = [False] * len(self.flat.params)
        late, self._late = self._late, False
        if self.overlap:
            if self.expected is None:
                self.expected, self.hooked = list(self._seen), list(self._hook_seen)
            elif self._seen != self.expected or any(h and not s for h, s in zip(self.hooked, self._hook_seen)):
                self.expected = self.hooked = None   # the graph changed: re-calibrate (one step without early launches)
        self._seen = [0] * len(self.flat.params)
        self._hook_seen = [False] * len(self.flat.params)
        self._precount()
        if late:
            # A gradient was written into a slice AFTER its all-reduce had been queued (a data-dependent branch used a parameter more
            # often than the calibration step did): that contribution is rank-local, the replicas would silently diverge.  No rank can
            # repair it alone, so the step must not be applied: stop here (the other ranks stop at their next collective).
            info, self._late_info = self._late_info[:8], []
            raise RuntimeError("GradSync: a parameter gradient was written after its bucket's all-reduce was launched; the reduced "
                               "gradients of this step are inconsistent across ranks (construct GradSync(overlap=False) for models "
                               f"whose parameter use varies from step to step).  (route, parameter, expected, seen): {info}")
        return 1.0 / self.world       # fold the mean into the optimiser's gradient scale


    def exposed_ms(self):
        """Total ms the compute stream waited on gradient all-reduces so far (synchronises; measurement only)."""
        if not self._stall:
            return 0.0
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in self._stall))


def barrier():
    """dist.barrier() when a process group is up (no-op in single-process runs)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_reduce_mean_scalar(t):
    """sync_dist=True logging (models.py:135,144,244,254): mean of a scalar over ranks."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= dist.get_world_size()
    return t
