"""torch.autograd glue over the C-ABI kernels (include/remfx_hip.h).

PyTorch is used here only for device memory, streams and the autograd tape; every
op below launches hand-written HIP kernels from libremfx_hip.so.  There is no CPU
or ATen fallback: tensors must be fp32 CUDA(HIP) tensors and the library must load.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, convplan
from ._lib import Epilogue, GemmDesc, check

# arithmetic of the MFMA gather-GEMMs: "f32" (exact fp32 MFMA), "bf16x3" (split-bf16 emulation of fp32 products) or
# "bf16" (operands rounded to bf16, fp32 accumulate: trainer.precision=bf16-mixed)
import os as _os
PREC_NAMES = {"f32": 0, "bf16x3": 1, "bf16": 2}
GEMM_PREC = PREC_NAMES[_os.environ.get("RFX_GEMM_PREC", "f32")]


BF16_STORE = True      # tests flip it to compare 16-bit storage with fp32 storage bit for bit


def bf16_storage():
    """bf16 mode keeps selected activations (conv outputs that are read only by GLU / norm kernels and GEMM operands, and their
    gradients) in 16 bits -- DESIGN.md section 3."""
    return GEMM_PREC == 2 and BF16_STORE


def set_gemm_precision(name):
    global GEMM_PREC
    GEMM_PREC = PREC_NAMES[name]


def gemm_precision():
    return {v: k for k, v in PREC_NAMES.items()}[GEMM_PREC]


class at_least_fp32_parity:
    """Scope in which the gather-GEMMs run at fp32 parity whatever the session's mode: `bf16` is lifted to `bf16x3`
    (f32 / bf16x3 stay).  Used by the effect detector: its thresholded labels must equal the fp32 reference's bit for bit
    (north_star; reference remfx/models.py:60-64), and 41 GFLOP per clip do not need the single-bf16 pipe."""

    def __enter__(self):
        global GEMM_PREC
        self.prev = GEMM_PREC
        if GEMM_PREC == 2:
            GEMM_PREC = 1
        return self

    def __exit__(self, *exc):
        global GEMM_PREC
        GEMM_PREC = self.prev
        return False


class GradSink:
    """Where the backward kernels put PARAMETER gradients when the parameters live in an optim.FlatParams buffer: straight
    into the parameter's slice of the flat gradient buffer (accumulating: the buffer is zeroed once per step), instead of
    returning a fresh tensor that autograd then adds into `.grad` with one small launch per parameter (322 launches per Demucs
    step).  Weight-gradient GEMMs additionally run on a SIDE stream: nothing in the backward chain depends on them, so they
    fill the machine under the latency-bound kernels of the main stream (LSTM recurrence, FFT, small layers); the optimiser
    (FlatAdamW.step) and the gradient exchange (ddp.GradSync) join the side stream before they read the buffer.

    Armed by FlatParams.zero_grad(), joined / disarmed by FlatParams.join().  A weight is recognised by its storage address:
    the tensor a backward node saved must start where a parameter's view of the flat buffer starts, be contiguous and have the
    parameter's size (`w.unsqueeze(2)` of a Conv1d weight qualifies, a transposed or concatenated weight does not and takes the
    ordinary autograd route)."""

    MODE = "side"      # "side": weight gradients on a side stream; "main": in place on the compute stream; "off": autograd accumulation

    def __init__(self, flat, side_stream=True):
        self.flat = flat
        self.base = flat.data.data_ptr()
        self.index = {self.base + 4 * o: i for i, o in enumerate(flat.offsets)}
        dev = flat.data.device
        self.side = torch.cuda.Stream(device=dev) if (side_stream and dev.type == "cuda") else None
        self.used_side = False
        self.writes = [0] * len(flat.params)         # sink writes per parameter in the current step
        self.on_write = None                           # ddp.GradSync: called with the parameter index after each write
        # every stream a gradient was written on in this step (raw handle -> torch stream): the time branch of Hybrid Demucs
        # back-propagates on its own stream, and a node that writes through the sink returns None, so no AccumulateGrad orders its
        # writes for the engine.  join() and ddp.GradSync order the optimiser step / a bucket's all-reduce behind all of them.
        self.write_streams = {}

    def lookup(self, w):
        """(index, 1-D gradient view) of the parameter `w` is, or None."""
        if w is None or not w.is_cuda:
            return None
        i = self.index.get(w.data_ptr())
        if i is None:
            return None
        p = self.flat.params[i]
        if w.numel() != p.numel() or not w.is_contiguous() or w.dtype != torch.float32:
            return None
        o = self.flat.offsets[i]
        return i, self.flat.grad[o:o + p.numel()]

    def note_stream(self):
        """Remember the current stream as one that carries gradient writes of this step."""
        raw = _raw_stream(_cur_device())
        if raw not in self.write_streams:
            self.write_streams[raw] = torch.cuda.current_stream()

    def order_after_writes(self, stream):
        """`stream` waits for everything enqueued so far on every stream that has written gradients in this step."""
        for raw, st in self.write_streams.items():
            if st != stream:
                stream.wait_stream(st)

    def wrote(self, i):
        self.writes[i] += 1
        if self.flat.grad.is_cuda:
            self.note_stream()
        if self.on_write is not None:
            self.on_write(i)

    def stream_for_wgrad(self, *operands):
        """The stream weight-gradient work goes to (ordered after everything already on the current stream).  The operands were
        allocated on the current stream: tell the caching allocator the side stream uses them too."""
        main = torch.cuda.current_stream()
        self.note_stream()
        if self.side is None:
            return main
        self.side.wait_stream(main)
        for t in operands:
            if t is not None:
                t.record_stream(self.side)
        self.used_side = True
        return self.side

    def join(self):
        """Current stream waits for every queued weight gradient: the side stream's, and whatever other streams (the time branch's)
        backward nodes wrote gradients on."""
        cur = torch.cuda.current_stream() if self.flat.grad.is_cuda else None
        if self.side is not None and self.used_side:
            cur.wait_stream(self.side)
            self.used_side = False
        if cur is not None and self.write_streams:
            self.order_after_writes(cur)
            self.write_streams = {}


class StepGC:
    """Python's cyclic collector out of the step loop.  A training step creates ~10^4 short-lived objects (autograd nodes, descriptors,
    views); the generational collector then fires at arbitrary points of the enqueue loop, and at 8 clips per GPU -- where the host
    needs 23 of the step's 31 ms to enqueue it -- its pauses make the step host-bound: 32 - 36 ms against a steady 31 ms without them
    (same box, three alternating pairs, round 6).  Inside the `with` block automatic collection is off, everything alive at entry
    (model, plans, caches: what makes a full collection take tens of ms) is frozen out of the collector's sight, and `tick()` (once
    per step) collects the YOUNG generations every `every` steps, between two steps -- cyclic garbage of the steps since, nothing
    else."""

    def __init__(self, every=50):
        self.every, self.n, self.was = every, 0, False

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.freeze()
        gc.disable()
        return self

    def tick(self):
        self.n += 1
        if self.n % self.every == 0:
            import gc
            gc.collect(1)

    def __exit__(self, *exc):
        import gc
        gc.unfreeze()
        if self.was:
            gc.enable()
        return False


_COMPUTE_STREAMS = {}


def enter_compute_stream(device):
    """Make a HIGH-priority stream the current stream of `device` (once per process and device; idempotent) and return it.
    The training loops (trainer.Trainer.fit, bench.py) call this before their first step: the GradSink's side stream keeps the
    normal priority, so the dispatcher serves the compute stream -- the critical path of the step -- first and the weight-gradient
    GEMMs fill what is left.  Same-box A/B of the Demucs step: 145.4 -> 144.1 ms and 145.65 -> 144.7 ms (the opposite assignment, side stream
    high: +0.9 ms; enqueueing a layer's weight gradient before its input gradient instead of after: no change).
    gfx950 / HIP exposes two levels (torch.cuda.Stream.priority_range() == (0, -1))."""
    device = torch.device(device)
    if device.type != "cuda":
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _COMPUTE_STREAMS.get(idx)
    if st is None:
        torch.cuda.synchronize(idx)                # everything enqueued so far (parameter init, H2D copies) is done
        st = torch.cuda.Stream(device=idx, priority=-1)
        _COMPUTE_STREAMS[idx] = st
    torch.cuda.set_stream(st)
    return st


SINK = None                   # the armed GradSink (optim.FlatParams.zero_grad arms, .join disarms)
TRACE_VARIANT = None          # bench.py sets this to a list: gemm_fwd appends rfx_gemm_fwd_variant() of every launch


ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2, "tanh": 3, "prelu": 4, "leaky": 5, "sigmoid": 6}


# torch.cuda.current_stream() builds a Stream object through four Python layers (~9 us); a training step asks ~1200 times (11 ms of
# the 34 ms a step costs the host at 8 clips, where the host is what bounds the step: scripts/host_time.py).  The raw handle is one
# C call.
_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def raw_stream():
    """hipStream_t of the current device's current stream, as an int."""
    return _raw_stream(_cur_device())


def _stream():
    return C.c_void_p(_raw_stream(_cur_device()))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def zeros(shape, device, dtype=torch.float32):
    """torch.zeros on the GPU through rfx_zero (one launch at HBM rate on the current stream)."""
    t = torch.empty(shape, device=device, dtype=dtype)
    if t.numel():
        check(_lib.lib().rfx_zero(_ptr(t), t.numel() * t.element_size(), _stream()), "rfx_zero")
    return t


def zero_(t):
    """t.zero_() for a contiguous GPU tensor of a 4- or 8-byte type."""
    if t.numel():
        check(_lib.lib().rfx_zero(_ptr(t), t.numel() * t.element_size(), _stream()), "rfx_zero")
    return t


def _req(t, name="tensor"):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise ValueError(f"{name}: remfx_amd ops need fp32 tensors on the GPU (got {t.dtype}, {t.device}); "
                         "there is no CPU fallback")


# ---- plan cache (tables live on the device next to the tensors) -------------------
class DevPlan:
    def __init__(self, plan, device):
        self.p = plan
        self.ktab = torch.from_numpy(np.ascontiguousarray(plan.ktab)).to(device)
        self.woff = torch.from_numpy(np.ascontiguousarray(plan.woff)).to(device)
        d = GemmDesc()
        for f, _ in GemmDesc._fields_:
            if f not in ("gpt2", "in_extent", "in_bf16", "out_bf16"):
                setattr(d, f, int(getattr(plan, f)))
        d.in_extent = int(plan.in_extent) * 4              # bytes
        self.desc = d
        self._desc16 = {}
        self.tap = plan.cin >= 8 and plan.R > 0              # tap-major tables exist: bf16x3 / bf16 run the gemm_tap kernels
        if self.tap:
            self.tap_tab = torch.from_numpy(np.ascontiguousarray(plan.tap_tab)).to(device)
            self.woff_t = torch.from_numpy(np.ascontiguousarray(plan.woff_t)).to(device)

    def fwd_prec(self):
        """Arithmetic the forward-family launch of this plan runs in: the library mode for tap-major plans, exact fp32 on
        the channel-major kernel otherwise (fewer than 8 input channels / thin M <= 8 layers: HBM-bound either way)."""
        return GEMM_PREC if self.tap else 0

    def desc_for(self, x, out):
        """The descriptor with the storage types of this call's operands (bf16 STORAGE of the gathered / written tensor: a
        copy of the plan's descriptor with the flag set and the operand's byte extent halved; fp32 + fp32 is the plan's own)."""
        in16, out16 = x.dtype == torch.bfloat16, out.dtype == torch.bfloat16
        if not (in16 or out16):
            return self.desc
        if (in16 and x.dtype != torch.bfloat16) or GEMM_PREC != 2:
            raise ValueError("bf16 tensor storage needs the bf16 arithmetic mode")
        key = (in16, out16)
        v = self._desc16.get(key)
        if v is None:
            v = GemmDesc.from_buffer_copy(self.desc)
            v.in_bf16, v.out_bf16 = int(in16), int(out16)
            v.in_extent = int(self.p.in_extent) * (2 if in16 else 4)
            self._desc16[key] = v
        return v

    def set_io(self, x=None, out=None):
        """Refresh strides for a new tensor with the same geometry key (cheap)."""
        return self


_PLANS = {}


def _plans(key, device, builder):
    k = (key, str(device))
    v = _PLANS.get(k)
    if v is None:
        built = builder()
        v = [DevPlan(p, device) for p in built] if isinstance(built, list) else DevPlan(built, device)
        _PLANS[k] = v
    return v


def pack_a(dp, w, out=None):
    """Gather the weight tensor into the packed [Kpad][Mpad] A matrix of a plan (out: an earlier result of the same plan to overwrite)."""
    p = dp.p
    prec = dp.fwd_prec()
    if prec == 0:
        apack = out if out is not None else torch.empty((p.Kpad + 64, p.Mpad), device=w.device, dtype=torch.float32)
        check(_lib.lib().rfx_pack_a(_ptr(w), _ptr(dp.woff), p.w_ms, p.M, p.extra["n_weight_rows"], p.Mpad, p.Kpad, 0,
                                    _ptr(apack), _stream()), "rfx_pack_a")
    else:                                       # tap-major order, bf16 hi / lo cells: 2 x [Kpad_t / 8 + 8][Mpad] x 16 bytes
        apack = out if out is not None else torch.empty((p.Kpad_t + 64, p.Mpad), device=w.device, dtype=torch.float32)
        check(_lib.lib().rfx_pack_a(_ptr(w), _ptr(dp.woff_t), p.w_ms, p.M, p.Kpad_t, p.Mpad, p.Kpad_t, prec,
                                    _ptr(apack), _stream()), "rfx_pack_a")
    return apack


# ---- packed weights, cached ----------------------------------------------------------
# A layer's packed A matrix changes only when its weights do, yet it used to be rebuilt by every call (218 pack launches + the torch
# reshapes of the merged / interleaved forms per Demucs step, on the compute stream in front of the GEMM that needs them).  An entry
# is valid while (a) the tensor it was packed from is held alive HERE (a strong reference: its storage cannot be freed and re-used
# under the same address), (b) torch's version counter of that storage is unchanged (every in-place torch op, load_state_dict, the
# optimisers of torch.optim bump it) and (c) no NATIVE writer has touched the weights since (FlatAdamW's kernel and the parameter
# broadcast write through raw pointers: they call weights_changed()).  Entries are per plan, arithmetic mode and stream; the cache
# is bounded by bytes (oldest entries go first).  NOT seen: writes through `p.data` (they bypass the version counter) -- call
# ops.weights_changed() after such a write, or run with RFX_PACK_CACHE=0.
# (Rebuilding all entries on a side stream right after the optimiser step, so that a training step finds them ready, was built and
# measured SLOWER: 137.4 -> 141.8-146.9 ms at 64 clips, 39.5 -> 41.5-42.4 ms at 8 -- the cache therefore pays in inference and
# wherever a weight is used more than once between updates; a training step still packs each weight once per use.  Re-packing all
# plain weights in ONE launch on the compute stream right after the optimiser step (device descriptor table, in place) was also
# built: 136.3 -> 138.7 ms at 64 clips -- the pack kernels' cost is their gather (a lane per output row: 64 cache lines per wave
# load), not their launches, so one big launch in front of the forward pass is slower than 218 small ones spread through the step.)
PACK_CACHE = _os.environ.get("RFX_PACK_CACHE", "1") != "0"
PACK_CACHE_BYTES = 6 << 30        # upper bound; _pack_cache_limit() also keeps it below 1/16 of the device's memory
_PACK_LIMIT = {}


def _pack_cache_limit(device):
    """Byte bound of the pack cache on `device`: PACK_CACHE_BYTES, at most 1/16 of the device's total memory (a fixed 6 GiB was a
    quarter of a small card); RFX_PACK_CACHE_BYTES overrides."""
    k = str(device)
    v = _PACK_LIMIT.get(k)
    if v is None:
        env = _os.environ.get("RFX_PACK_CACHE_BYTES")
        if env is not None:
            v = int(env)
        else:
            v = PACK_CACHE_BYTES
            try:
                v = min(v, torch.cuda.get_device_properties(device).total_memory // 16)
            except Exception:
                pass
        _PACK_LIMIT[k] = v
    return v
_PACKS = {}
_PACKS_BYTES = [0]
_WEIGHT_EPOCH = [0]


class _PackEntry:
    __slots__ = ("src", "dp", "ver", "epoch", "apack", "nbytes")


def weights_changed():
    """A native kernel wrote parameters in place (no torch version bump): every cached pack is stale."""
    _WEIGHT_EPOCH[0] += 1


def _derived(src, derive):
    return derive(src) if derive is not None else src.contiguous()


def pack_cached(dp, src, derive=None, tag=0):
    """Packed A of plan dp for the weight tensor src; derive(src) -> the tensor rfx_pack_a gathers from (default: src.contiguous())."""
    if PACK_CACHE:
        try:
            ver = src._version
        except RuntimeError:                       # inference tensors carry no version counter: no caching
            ver = None
        if ver is not None:
            prec = dp.fwd_prec()
            key = (id(dp), src.data_ptr(), tuple(src.stride()), tag, prec, raw_stream())
            e = _PACKS.get(key)
            if e is not None and e.dp is dp and e.ver == ver and e.epoch == _WEIGHT_EPOCH[0]:
                return e.apack
            apack = pack_a(dp, _derived(src, derive))
            if e is not None:
                _PACKS_BYTES[0] -= e.nbytes
                del _PACKS[key]                    # re-insert at the young end
            e = _PackEntry()
            e.src, e.dp, e.ver, e.epoch, e.apack = src, dp, ver, _WEIGHT_EPOCH[0], apack
            e.nbytes = apack.numel() * 4 + src.numel() * src.element_size()
            _PACKS[key] = e
            _PACKS_BYTES[0] += e.nbytes
            while _PACKS_BYTES[0] > _pack_cache_limit(src.device) and len(_PACKS) > 1:
                k0 = next(iter(_PACKS))
                _PACKS_BYTES[0] -= _PACKS.pop(k0).nbytes
            return apack
    return pack_a(dp, _derived(src, derive))


def clear_pack_cache():
    _PACKS.clear()
    _PACKS_BYTES[0] = 0


def gemm_fwd(dp, apack, x, out, bias=None, act=None, act_param=None, res=None, act2=None,
             dp2=None, apack2=None, in2=None, bwd=False, gparam=None, stat_sums=None, glu_out=None):
    e = Epilogue()
    e.bias = bias.data_ptr() if bias is not None else None
    e.act = ACT[act]
    e.act_param = act_param.data_ptr() if act_param is not None else None
    if res is not None:
        e.res = res.data_ptr()
        e.res_ns, e.res_cs, e.res_as, e.res_bs = res.stride()
    e.act2 = ACT[act2]
    e.bwd = 1 if bwd else 0
    e.gparam = gparam.data_ptr() if gparam is not None else None
    e.stat_sums = stat_sums.data_ptr() if stat_sums is not None else None
    e.stat_slots = stat_sums.shape[1] if (stat_sums is not None and stat_sums.dim() == 3) else 1
    if gparam is not None and gparam.dim() == 2:          # [slots][M] partial sums of the slope gradient
        e.stat_slots = gparam.shape[0]
    if glu_out is not None:                               # fused GLU store (rows interleaved by the caller)
        e.glu_out = glu_out.data_ptr()
        e.glu_ns = glu_out.stride(0)
    prec = dp.fwd_prec()
    if dp2 is not None and dp2.fwd_prec() != prec:
        raise ValueError("two-phase gather-GEMM: both phases must use the same table form")
    tab = dp.tap_tab if prec else dp.ktab
    if dp2 is not None and prec:
        a2, k2, K2, Kpad2 = _ptr(apack2), _ptr(dp2.tap_tab), dp2.p.ntaps, dp2.p.Kpad_t
        dp.desc.gpt2 = int(dp2.p.gpt)                      # the second table's own channel blocking
    elif dp2 is not None:
        a2, k2, K2, Kpad2 = _ptr(apack2), _ptr(dp2.ktab), dp2.p.K, dp2.p.Kpad
    else:
        a2, k2, K2, Kpad2 = None, None, 0, 0
    if x.dtype == torch.bfloat16 and not (prec == 2 and dp.p.R > 0):
        x = x.float()                # thin / channel-major plans (fewer than 8 input channels, M <= 8) read fp32: widen once
    desc = dp.desc_for(x, out)
    if dp2 is not None and prec:
        desc.gpt2 = int(dp2.p.gpt)
    if TRACE_VARIANT is not None:               # measurement hook (bench.py): which kernel instantiation this launch is
        TRACE_VARIANT.append(_lib.lib().rfx_gemm_fwd_variant(C.byref(desc), C.byref(e), int(dp2 is not None), prec))
    check(_lib.lib().rfx_gemm_fwd(C.byref(desc), _ptr(apack), _ptr(tab), _ptr(x), _ptr(out),
                                  C.byref(e), a2, k2, K2, Kpad2, _ptr(in2), prec, _stream()),
          "rfx_gemm_fwd")
    return out


class WgradOut:
    """What rfx_gemm_wgrad leaves behind: `splits` partial [M][Kpad] matrices, one per position split, to be added in order by an
    unpack entry point (fixed-order reduction: the weight gradients are deterministic, nothing is zero-filled)."""
    __slots__ = ("ws", "splits")

    def __init__(self, ws, splits):
        self.ws, self.splits = ws, splits


WGRAD_SPLITS = int(_os.environ.get("RFX_WGRAD_SPLITS", "256"))     # position splits of a gather-GEMM weight gradient (A/B)


def gemm_wgrad(dp, x, g):
    p = dp.p
    sl = p.M * p.Kpad
    # at most 256 position splits (one per CU): every split costs the fixed-order reduction in the unpack kernels a slice to read --
    # with up to 2048 slices those kernels took 8.3 ms per Demucs step (r05 kernel stats), 104 launches at 0.7 TB/s
    cap = sl * min(WGRAD_SPLITS, max(4, (1 << 25) // sl))
    ws = torch.empty(cap, device=x.device, dtype=torch.float32)
    ns = C.c_int32(0)
    if x.dtype == torch.bfloat16:                            # bf16 storage is implemented for the gradient operand only
        x = x.float()
    if g.dtype == torch.bfloat16:
        rc = _lib.lib().rfx_gemm_wgrad(C.byref(dp.desc_for(x, g)), _ptr(dp.ktab), _ptr(x), _ptr(g), _ptr(ws), cap, C.byref(ns),
                                       GEMM_PREC, _stream())
        if rc == 0:
            return WgradOut(ws, ns.value)
        if rc != -1:                                         # -1 = "plan not supported by the 16-bit path"; anything else is a real failure
            check(rc, "rfx_gemm_wgrad")
        g = g.float()                                        # a plan the wide-load kernel does not take: widen once
    check(_lib.lib().rfx_gemm_wgrad(C.byref(dp.desc), _ptr(dp.ktab), _ptr(x), _ptr(g), _ptr(ws), cap, C.byref(ns),
                                    GEMM_PREC, _stream()), "rfx_gemm_wgrad")
    return WgradOut(ws, ns.value)


def unpack_set(dp, wg, dw):
    """dw = sum of the split matrices, unpacked; only for plans whose rows cover every element of dw exactly once (dense conv plans)."""
    p = dp.p
    check(_lib.lib().rfx_unpack_set(_ptr(wg.ws), _ptr(dp.woff), p.w_ms, p.M, p.extra["n_weight_rows"],
                                    p.Kpad, _ptr(dw), wg.splits, _stream()), "rfx_unpack_set")


def unpack_add(dp, wg, dw):
    p = dp.p
    check(_lib.lib().rfx_unpack_add(_ptr(wg.ws), _ptr(dp.woff), p.w_ms, p.M, p.extra["n_weight_rows"],
                                    p.Kpad, _ptr(dw), wg.splits, _stream()), "rfx_unpack_add")


def unpack_col(dp, wg, col):
    """(M,) = column `col` of the summed split matrices (the bias-gradient column)."""
    p = dp.p
    out = torch.empty(p.M, device=wg.ws.device, dtype=torch.float32)
    check(_lib.lib().rfx_unpack_col(_ptr(wg.ws), p.M, p.Kpad, col, wg.splits, _ptr(out), _stream()), "rfx_unpack_col")
    return out


# ---- convolution (4-D view: N, C, A, B) ----------------------------------------------
def _key(*a):
    return tuple(tuple(x) if isinstance(x, (list, tuple, torch.Size)) else x for x in a)


def conv2d_forward(x, w, bias, stride, padding, dilation, act=None, act_param=None, out=None, stat_sums=None,
                   out_bf16=False):
    """out_bf16: the caller's only readers of the result are a GroupNorm kernel and GEMM operands -- in the bf16 arithmetic mode
    it is then STORED in 16 bits (what torch autocast stores for a conv output); ignored in the other modes."""
    _req(x, "x"); _req(w, "weight")
    N, Cin, IA, IB = x.shape
    Cout, _, KA, KB = w.shape
    OA = convplan._out_len(IA, KA, stride[0], padding[0], dilation[0])
    OB = convplan._out_len(IB, KB, stride[1], padding[1], dilation[1])
    if out is None:
        use16 = out_bf16 and bf16_storage() and Cout > 8 and (OA * OB) % 4 == 0 and act is None
        out = torch.empty((N, Cout, OA, OB), device=x.device, dtype=torch.bfloat16 if use16 else torch.float32)
    key = _key("cf", x.shape, x.stride(), w.shape, stride, padding, dilation, out.stride())
    dp = _plans(key, x.device, lambda: convplan.conv_fwd_plan(
        tuple(x.shape), x.stride(), tuple(w.shape), stride, padding, dilation, out.stride()))
    gemm_fwd(dp, pack_cached(dp, w), x, out, bias=bias, act=act, act_param=act_param, stat_sums=stat_sums)
    return out


def _merge_axis(ksize, stride, padding, dilation):
    """Axis along which the stride phases can be merged into one GEMM (convplan.merged_phase_plan), or None:
    exactly one strided axis with K % S == 0, S a power of two, no dilation; the other axis a plain 1-tap pass."""
    if tuple(dilation) != (1, 1):
        return None
    for ax in (0, 1):
        o = 1 - ax
        S, K = stride[ax], ksize[ax]
        if (2 <= S <= 256 and (S & (S - 1)) == 0 and K % S == 0 and stride[o] == 1 and ksize[o] == 1 and padding[o] == 0):
            return ax
    return None


def _merged_weight(w_oik, ax, S):
    """(O, I, KA, KB) correlation weight -> (O*S, I, J taps) rows ordered (o, phase), taps in gather order."""
    O, I = w_oik.shape[:2]
    J = w_oik.shape[2 + ax] // S
    wm = w_oik.reshape(O, I, J, S).flip(2).permute(0, 3, 1, 2).reshape(O * S, I, J)      # [o*S+q][i][t] = w[o][i][q + S*(J-1-t)]
    return (wm.unsqueeze(3) if ax == 0 else wm.unsqueeze(2)).contiguous(), J


def _merged_launch(inp, w_oik, ax, S, off, out, bias=None, res=None, w_src=None, tag=0):
    """w_oik: the (O, I, KA, KB) correlation weight (possibly a transposed view); w_src: the tensor the pack cache watches."""
    O, I = w_oik.shape[:2]
    J = w_oik.shape[2 + ax] // S
    wm_shape = (O * S, I, J, 1) if ax == 0 else (O * S, I, 1, J)
    key = _key("mg", inp.shape, inp.stride(), wm_shape, ax, S, off, out.shape, out.stride())
    dp = _plans(key, inp.device, lambda: convplan.merged_phase_plan(
        tuple(inp.shape), inp.stride(), wm_shape[0], ax, S, J, off, out.shape[2 + ax], out.stride()))
    apack = pack_cached(dp, w_src if w_src is not None else w_oik, derive=lambda _t: _merged_weight(w_oik, ax, S)[0], tag=tag)
    gemm_fwd(dp, apack, inp, out, bias=bias, res=res)
    return out


def conv2d_dgrad(g, w, xshape, xstrides, stride, padding, dilation, dx=None, res=None):
    """res: optional tensor of x's shape added to the result in the GEMM's store (a second gradient path into x that
    autograd would otherwise accumulate with a separate add pass); in the store for unit-stride plans, a plain add otherwise."""
    if dx is None:
        dx = torch.empty_strided(xshape, xstrides, device=g.device, dtype=torch.float32)
    ax = _merge_axis(w.shape[2:], stride, padding, dilation)
    if ax is not None and w.shape[1] * stride[ax] >= 4:
        # dx[ci][o] = sum g[co][i] w[co][ci][kk], o = S*i + kk - P: all S phases of o as rows (ci, q) of one GEMM over g
        # the residual (skip-connection gradient) rides in the merged-phase store: stride 4 along a unit-stride axis makes a
        # lane's four phases one 16-byte run, so the residual is ONE 16-byte load per lane and channel (round 1 measured the
        # strided 4-byte form as a loss and kept the torch add instead)
        # (time branch); along the frequency axis the four phases are four rows and the residual loads coalesce across lanes
        return _merged_launch(g, w.transpose(0, 1), ax, stride[ax], -padding[ax], dx, res=res, w_src=w, tag=1)
    if res is not None and tuple(stride) != (1, 1):       # per-phase plans: add afterwards
        return conv2d_dgrad(g, w, xshape, xstrides, stride, padding, dilation, dx).add_(res)
    key = _key("cd", xshape, xstrides, w.shape, stride, padding, dilation, g.shape, g.stride())
    dps = _plans(key, g.device, lambda: convplan.conv_dgrad_plans(
        tuple(xshape), tuple(xstrides), tuple(w.shape), stride, padding, dilation, tuple(g.shape), g.stride()))
    for dp in dps:
        gemm_fwd(dp, pack_cached(dp, w), g, dx, res=res)
    return dx


def conv2d_wgrad(x, g, wshape, stride, padding, dilation, need_bias, w=None, b=None):
    """w, b: the weight / bias tensors of the forward call (as the backward node saved them).  With a GradSink armed and both
    recognised as flat-buffer parameters the gradients are accumulated in place on the sink's side stream and (None, None) is
    returned; otherwise (dw, db) as fresh tensors for autograd."""
    key = _key("cw", x.shape, x.stride(), wshape, stride, padding, dilation, g.stride(), need_bias)
    dp = _plans(key, x.device, lambda: convplan.conv_fwd_plan(
        tuple(x.shape), x.stride(), tuple(wshape), stride, padding, dilation, g.stride(), bias_row=need_bias))
    p = dp.p
    sink = SINK
    tw = sink.lookup(w) if (sink is not None and w is not None) else None
    tb = sink.lookup(b) if (tw is not None and need_bias) else None
    if tw is not None and (tb is not None or not need_bias):
        with torch.cuda.stream(sink.stream_for_wgrad(x, g)):
            wg = gemm_wgrad(dp, x, g)
            if need_bias:       # weight + bias gradient out of the same matrices in one launch
                check(_lib.lib().rfx_unpack_add_bias(_ptr(wg.ws), _ptr(dp.woff), p.w_ms, p.M, p.extra["n_weight_rows"], p.Kpad,
                                                     _ptr(tw[1]), p.K - 1, _ptr(tb[1]), wg.splits, _stream()), "rfx_unpack_add_bias")
            else:
                unpack_add(dp, wg, tw[1])                  # conv_fwd_plan rows = every live (ci, ka, kb) of every output channel, once
        sink.wrote(tw[0])
        if need_bias:
            sink.wrote(tb[0])
        return None, None
    wg = gemm_wgrad(dp, x, g)
    if p.extra.get("dense", True):
        dw = torch.empty(wshape, device=x.device, dtype=torch.float32)
    else:                                                  # taps that only meet the padding have no row: their gradient is 0
        dw = zeros(tuple(wshape), x.device)
    unpack_set(dp, wg, dw)                                 # conv_fwd_plan rows = every live (ci, ka, kb) of every output channel, once
    db = unpack_col(dp, wg, p.K - 1) if need_bias else None
    return dw, db


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d / nn.Conv1d (A == 1) forward + backward on the gather-GEMM kernels."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, dilation, stat_sums=None, out_bf16=False):
        ctx.save_for_backward(x, w)
        ctx.bias = bias                                   # identity only (GradSink lookup); never read
        ctx.cfg = (stride, padding, dilation, bias is not None)
        return conv2d_forward(x, w, bias, stride, padding, dilation, stat_sums=stat_sums, out_bf16=out_bf16)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, padding, dilation, has_bias = ctx.cfg
        g = g if g.is_contiguous() else g.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_dgrad(g, w, tuple(x.shape), tuple(x.stride()), stride, padding, dilation)
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw, db = conv2d_wgrad(x, g, tuple(w.shape), stride, padding, dilation, has_bias, w, ctx.bias)
        return dx, dw, db, None, None, None, None, None


class ConvFork2dFn(torch.autograd.Function):
    """(conv(x), x): the convolution plus a pass-through alias of its input for a residual branch that re-joins later
    (HDemucs DConv: x + scale * branch(x)).  Routing the alias through this node hands BOTH gradients of x to one
    backward call, so the residual gradient is added in the input-gradient GEMM's store instead of a separate
    read-read-write accumulation pass over the activation."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, dilation, stat_sums=None, out_bf16=False):
        ctx.save_for_backward(x, w)
        ctx.bias = bias
        ctx.cfg = (stride, padding, dilation, bias is not None)
        return conv2d_forward(x, w, bias, stride, padding, dilation, stat_sums=stat_sums, out_bf16=out_bf16), x.view_as(x)

    @staticmethod
    def backward(ctx, g, gres):
        x, w = ctx.saved_tensors
        stride, padding, dilation, has_bias = ctx.cfg
        g = g if g.is_contiguous() else g.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_dgrad(g, w, tuple(x.shape), tuple(x.stride()), stride, padding, dilation, res=gres)
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw, db = conv2d_wgrad(x, g, tuple(w.shape), stride, padding, dilation, has_bias, w, ctx.bias)
        return dx, dw, db, None, None, None, None, None


class ConvGlu2dFn(torch.autograd.Function):
    """GLU(conv(x)) over the channel axis (HDemucs rewrite conv + GLU) with the GLU in the GEMM's store: the GEMM runs on
    the weight rows interleaved (c, C+c) so a lane holds both halves; the conv output is still written (the backward
    needs it) but never re-read in the forward pass."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, dilation):
        _req(x, "x"); _req(w, "weight")
        N, Cin, IA, IB = x.shape
        C2, _, KA, KB = w.shape
        Ch = C2 // 2
        OA = convplan._out_len(IA, KA, stride[0], padding[0], dilation[0])
        OB = convplan._out_len(IB, KB, stride[1], padding[1], dilation[1])
        y2 = torch.empty((N, C2, OA, OB), device=x.device, dtype=torch.float32)
        out = torch.empty((N, Ch, OA, OB), device=x.device, dtype=torch.float32)
        key = _key("cf", x.shape, x.stride(), w.shape, stride, padding, dilation, y2.stride())
        dp = _plans(key, x.device, lambda: convplan.conv_fwd_plan(
            tuple(x.shape), x.stride(), tuple(w.shape), stride, padding, dilation, y2.stride()))
        # bf16 mode: the conv output and its gradient never leave this node and are read only as GLU inputs / GEMM operands
        # (which round to bf16 anyway): keep both in 16 bits -- what torch autocast stores for a conv output
        if bf16_storage() and dp.tap and C2 >= 16 and Cin > 8 and (Ch * OA * OB) % 4 == 0:
            y2 = torch.empty((N, C2, OA, OB), device=x.device, dtype=torch.bfloat16)
        apack = pack_cached(dp, w, tag=2,                                            # rows (c, half)
                            derive=lambda t: t.reshape(2, Ch, Cin, KA, KB).transpose(0, 1).reshape(C2, Cin, KA, KB))
        gemm_fwd(dp, apack, x, y2, bias=bias, glu_out=out)
        ctx.save_for_backward(x, w, y2)
        ctx.bias = bias
        ctx.cfg = (stride, padding, dilation, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, y2 = ctx.saved_tensors
        stride, padding, dilation, has_bias = ctx.cfg
        N, C2 = y2.shape[0], y2.shape[1]
        S = y2.numel() // (N * C2)
        g2 = torch.empty_like(y2)
        if y2.dtype == torch.bfloat16:
            check(_lib.lib().rfx_glu_bwd_bf16(_ptr(y2), _ptr(g.contiguous()), _ptr(g2), N, C2, S, _stream()), "rfx_glu_bwd_bf16")
        else:
            check(_lib.lib().rfx_glu_bwd(_ptr(y2), _ptr(g.contiguous()), _ptr(g2), N, C2, S, _stream()), "rfx_glu_bwd")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_dgrad(g2, w, tuple(x.shape), tuple(x.stride()), stride, padding, dilation)
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw, db = conv2d_wgrad(x, g2, tuple(w.shape), stride, padding, dilation, has_bias, w, ctx.bias)
        return dx, dw, db, None, None, None


def conv2d_glu(x, w, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
    """GLU(conv2d(x)) over channels; falls back to conv2d + glu when the fused store does not apply."""
    if w.shape[0] % 2 == 0 and w.shape[0] > 8:
        return ConvGlu2dFn.apply(x, w, bias, tuple(stride), tuple(padding), tuple(dilation))
    from . import nnops
    return nnops.glu(conv2d(x, w, bias, stride, padding, dilation), 1)


def conv1d_glu(x, w, bias=None, stride=1, padding=0, dilation=1):
    return conv2d_glu(x.unsqueeze(2), w.unsqueeze(2), bias, (1, stride), (0, padding), (1, dilation)).squeeze(2)


def conv2d(x, w, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), stat_sums=None, out_bf16=False):
    return Conv2dFn.apply(x, w, bias, tuple(stride), tuple(padding), tuple(dilation), stat_sums, out_bf16)


def conv1d(x, w, bias=None, stride=1, padding=0, dilation=1, stat_sums=None, out_bf16=False):
    """stat_sums: optional zeroed fp64 (N, 2) tensor; the GEMM epilogue adds {sum, sum^2} of every sample's
    output into it (GroupNorm(1, C) statistics for free).  out_bf16: see conv2d_forward."""
    y = Conv2dFn.apply(x.unsqueeze(2), w.unsqueeze(2), bias, (1, stride), (0, padding), (1, dilation), stat_sums, out_bf16)
    return y.squeeze(2)


def conv2d_fork(x, w, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), stat_sums=None, out_bf16=False):
    """(conv2d(x), alias of x) -- see ConvFork2dFn."""
    return ConvFork2dFn.apply(x, w, bias, tuple(stride), tuple(padding), tuple(dilation), stat_sums, out_bf16)


def conv1d_fork(x, w, bias=None, stride=1, padding=0, dilation=1, stat_sums=None, out_bf16=False):
    """(conv1d(x), alias of x) -- see ConvFork2dFn; use the alias for the residual connection around the branch."""
    y, xr = ConvFork2dFn.apply(x.unsqueeze(2), w.unsqueeze(2), bias, (1, stride), (0, padding), (1, dilation), stat_sums,
                               out_bf16)
    return y.squeeze(2), xr.squeeze(2)


# ---- transposed convolution -----------------------------------------------------------
def convT2d_forward(x, w, bias, stride, dilation, crop_lo, out_len, act=None):
    _req(x, "x"); _req(w, "weight")
    N, Cin, IA, IB = x.shape
    _, Cout, KA, KB = w.shape
    out = torch.empty((N, Cout, out_len[0], out_len[1]), device=x.device, dtype=torch.float32)
    ax = _merge_axis(w.shape[2:], stride, (0, 0), dilation)
    if (ax is not None and act is None and Cout * stride[ax] >= 4 and crop_lo[1 - ax] == 0
            and out_len[1 - ax] == x.shape[3 - ax]):
        # y[co][o] = sum x[ci][i] w[ci][co][kk], o = S*i + kk: the S phases of o as rows (co, q) of one GEMM over x
        return _merged_launch(x, w.transpose(0, 1), ax, stride[ax], -crop_lo[ax], out, bias=bias, w_src=w, tag=3)
    key = _key("tf", x.shape, x.stride(), w.shape, stride, dilation, crop_lo, out_len, out.stride())
    dps = _plans(key, x.device, lambda: convplan.convT_fwd_plans(
        tuple(x.shape), x.stride(), tuple(w.shape), stride, dilation, crop_lo, out_len, out.stride()))
    for dp in dps:
        gemm_fwd(dp, pack_cached(dp, w), x, out, bias=bias, act=act)
    return out


class ConvT2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d / 1d followed by the crop [lo : lo+len] on each axis."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, dilation, crop_lo, out_len):
        ctx.save_for_backward(x, w)
        ctx.bias = bias                                   # identity only (GradSink lookup)
        ctx.cfg = (stride, dilation, crop_lo, out_len, bias is not None)
        return convT2d_forward(x, w, bias, stride, dilation, crop_lo, out_len)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, dilation, crop_lo, out_len, has_bias = ctx.cfg
        g = g if g.is_contiguous() else g.contiguous()
        key = _key("td", x.shape, x.stride(), w.shape, stride, dilation, crop_lo, g.shape, g.stride())
        dx = dw = db = None
        need_w = ctx.needs_input_grad[1]
        if ctx.needs_input_grad[0] or need_w:
            xs = tuple(x.stride())
            dp = _plans(key, x.device, lambda: convplan.convT_dgrad_plan(
                tuple(x.shape), xs, tuple(w.shape), stride, dilation, crop_lo, tuple(g.shape), g.stride()))
            if ctx.needs_input_grad[0]:
                dx = torch.empty_strided(tuple(x.shape), xs, device=x.device, dtype=torch.float32)
                gemm_fwd(dp, pack_cached(dp, w), g, dx)
            sink = SINK
            tw = sink.lookup(w) if (sink is not None and need_w) else None
            tb = sink.lookup(ctx.bias) if (tw is not None and has_bias) else None
            if tw is not None and (tb is not None or not has_bias):
                # in place on the sink's side stream (see conv2d_wgrad)
                with torch.cuda.stream(sink.stream_for_wgrad(x, g)):
                    unpack_add(dp, gemm_wgrad(dp, g, x), tw[1])
                    if has_bias:
                        tb[1].add_(channel_sum(g))
                sink.wrote(tw[0])
                if has_bias:
                    sink.wrote(tb[0])
                return dx, None, None, None, None, None, None
            if need_w:
                dw = zeros(tuple(w.shape), w.device)
                unpack_add(dp, gemm_wgrad(dp, g, x), dw)
        if has_bias and ctx.needs_input_grad[2]:
            db = channel_sum(g)
        return dx, dw, db, None, None, None, None


def conv_transpose2d(x, w, bias=None, stride=(1, 1), dilation=(1, 1), crop_lo=(0, 0), out_len=None):
    if out_len is None:
        out_len = tuple(convplan.convT_out_len(x.shape[2 + i], w.shape[2 + i], stride[i], dilation[i]) - crop_lo[i]
                        for i in range(2))
    return ConvT2dFn.apply(x, w, bias, tuple(stride), tuple(dilation), tuple(crop_lo), tuple(out_len))


def conv_transpose1d(x, w, bias=None, stride=1, dilation=1, crop_lo=0, out_len=None):
    if out_len is None:
        out_len = convplan.convT_out_len(x.shape[-1], w.shape[-1], stride, dilation) - crop_lo
    y = ConvT2dFn.apply(x.unsqueeze(2), w.unsqueeze(2), bias, (1, stride), (1, dilation), (0, crop_lo),
                        (1, out_len))
    return y.squeeze(2)


# ---- small reductions / elementwise ----------------------------------------------------
def channel_sum(g):
    """sum over (N, A, B) of a (N, C, A, B) tensor -> (C,)"""
    g4 = g if g.dim() == 4 else g.unsqueeze(2)
    out = torch.empty(g4.shape[1], device=g.device, dtype=torch.float32)
    N, Cc, A, B = g4.shape
    s = g4.stride()
    L = _lib.lib()
    ws = torch.empty(int(L.rfx_channel_sum_ws(_ptr(g4), N, Cc, A, B, s[0], s[1], s[2], s[3])), device=g.device, dtype=torch.float64)
    check(L.rfx_channel_sum(_ptr(g4), N, Cc, A, B, s[0], s[1], s[2], s[3], _ptr(ws), _ptr(out), _stream()), "rfx_channel_sum")
    return out


def _rows16(x, gy, out, act):
    """rfx_act_rows16 on a contiguous tensor whose storage is bf16 (ops.bf16_storage): (rows, T) view, T = last axis."""
    T = x.shape[-1]
    R = x.numel() // T
    check(_lib.lib().rfx_act_rows16(_ptr(x), 0, 0, T, _ptr(gy) if gy is not None else None, 0, 0, T, _ptr(out), 0, 0, T,
                                    1, 1, R, T, ACT[act], _stream()), "rfx_act_rows16")


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        if x.dtype == torch.bfloat16:           # a conv output stored in 16 bits (bf16 mode): fp32 result, 16-bit gradient
            x = x.contiguous()
            y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
            _rows16(x, None, y, act)
        else:
            _req(x)
            x = x.contiguous()
            y = torch.empty_like(x)
            check(_lib.lib().rfx_act_fwd(_ptr(x), _ptr(y), x.numel(), ACT[act], _stream()), "rfx_act_fwd")
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        if x.dtype == torch.bfloat16:
            _rows16(x, gy, gx, ctx.act)
            return gx, None
        check(_lib.lib().rfx_act_bwd(_ptr(x), _ptr(gy), _ptr(gx), x.numel(), ACT[ctx.act], _stream()),
              "rfx_act_bwd")
        return gx, None


def activation(x, act):
    return ActFn.apply(x, act)


class ActAddFn(torch.autograd.Function):
    """act(x) + res in one pass (rfx_act_add_fwd): the HDemucs decoder's GELU and the next layer's `x + skip`."""

    @staticmethod
    def forward(ctx, x, res, act):
        _req(x); _req(res, "res")
        x, res = x.contiguous(), res.contiguous()
        y = torch.empty_like(x)
        check(_lib.lib().rfx_act_add_fwd(_ptr(x), _ptr(res), _ptr(y), x.numel(), ACT[act], _stream()), "rfx_act_add_fwd")
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        check(_lib.lib().rfx_act_bwd(_ptr(x), _ptr(gy), _ptr(gx), x.numel(), ACT[ctx.act], _stream()),
              "rfx_act_bwd")
        return gx, gy, None


def activation_add(x, res, act):
    return ActAddFn.apply(x, res, act)


def _row_strides(t):
    """(D0, D1, D2, T) tensor whose last axis is contiguous -> its three row strides."""
    if t.dim() != 4 or t.stride(3) != 1:
        raise ValueError("activation_to: 4-D tensors with a contiguous last axis")
    return t.stride()[:3]


class ActToFn(torch.autograd.Function):
    """act(x) written with the memory order `perm` of x's first three axes (the result is x-shaped, with permuted
    strides): the consumer's `permute(perm).reshape(...)` is then a view.  The backward reads the incoming gradient in
    whatever row layout it arrives (csrc rfx_act_rows) -- no `.contiguous()` copy on either side."""

    @staticmethod
    def forward(ctx, x, act, perm):
        x16 = x.dtype == torch.bfloat16         # a conv output stored in 16 bits (bf16 mode): fp32 result, 16-bit gradient
        if not x16:
            _req(x)
        rows = _lib.lib().rfx_act_rows16 if x16 else _lib.lib().rfx_act_rows
        xs = _row_strides(x)
        D = x.shape
        y = torch.empty([D[p] for p in perm] + [D[3]], device=x.device, dtype=torch.float32)
        inv = [perm.index(i) for i in range(3)]
        y = y.permute(*inv, 3)                                  # x-shaped view of the permuted buffer
        ys = _row_strides(y)
        check(rows(_ptr(x), xs[0], xs[1], xs[2], None, 0, 0, 0, _ptr(y), ys[0], ys[1], ys[2],
                   D[0], D[1], D[2], D[3], ACT[act], _stream()), "rfx_act_rows")
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        if gy.stride(3) != 1:
            gy = gy.contiguous()
        xs, gs, D = _row_strides(x), _row_strides(gy), x.shape
        gx = torch.empty_strided(x.shape, x.stride(), device=x.device, dtype=x.dtype)
        rows = _lib.lib().rfx_act_rows16 if x.dtype == torch.bfloat16 else _lib.lib().rfx_act_rows
        check(rows(_ptr(x), xs[0], xs[1], xs[2], _ptr(gy), gs[0], gs[1], gs[2], _ptr(gx), xs[0], xs[1],
                   xs[2], D[0], D[1], D[2], D[3], ACT[ctx.act], _stream()), "rfx_act_rows")
        return gx, None, None


def activation_to(x, act, perm):
    return ActToFn.apply(x, act, tuple(perm))
