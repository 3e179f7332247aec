"""Dev tool: one Demucs training step with every gather-GEMM launch timed (synchronising) and attributed to its caller:
kernel variant, operand dtypes and shapes, and the hdemucs.py / nnops.py / ops.py frames that issued it."""
import sys, os, time, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from remfx_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
ops.GradSink.MODE = "main"
model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
data = bench.synthetic_batch(B, 0, dev)
rec = []
phase = ["fwd"]


def site():
    out = []
    for f in traceback.extract_stack()[:-2]:
        fn = os.path.basename(f.filename)
        if fn in ("hdemucs.py", "nnops.py", "ops.py", "stft.py", "losses.py", "lstm.py"):
            out.append(f"{fn[:-3]}:{f.name}:{f.lineno}")
    return " > ".join(out[-4:])


def wrap_fwd(fn):
    def f(dp, apack, x, out, *a, **k):
        ops.TRACE_VARIANT = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(dp, apack, x, out, *a, **k)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        v = ops.TRACE_VARIANT[-1] if ops.TRACE_VARIANT else -1
        ops.TRACE_VARIANT = None
        p = dp.p
        fl = 2.0 * p.M * p.K * x.shape[0] * out.numel() / max(1, out.shape[0] * out.shape[1])
        rec.append(("F", v, str(x.dtype)[6:], str(out.dtype)[6:], tuple(x.shape), tuple(out.shape), p.M, p.K, dt, fl, site()))
        return r
    return f


def wrap_wg(fn):
    def f(dp, x, g):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(dp, x, g)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        p = dp.p
        fl = 2.0 * p.M * p.K * g.numel() / max(1, g.shape[1])
        rec.append(("W", -1, str(x.dtype)[6:], str(g.dtype)[6:], tuple(x.shape), tuple(g.shape), p.M, p.K, dt, fl, site()))
        return r
    return f


for it in range(3):
    if it == 2:
        ops.gemm_fwd = wrap_fwd(ops.gemm_fwd)
        ops.gemm_wgrad = wrap_wg(ops.gemm_wgrad)
    opt.zero_grad()
    loss = model.training_step(data, 0)
    loss.backward()
    opt.step(clip_norm=10.0)
    torch.cuda.synchronize()
L = bench._lib.lib() if hasattr(bench, "_lib") else None
tot = sum(r[8] for r in rec)
print(f"{len(rec)} gemm launches, {tot:.1f} ms (sync-timed, B={B})")
agg = collections.OrderedDict()
for r in rec:
    k = (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[10])
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += r[8]; a[2] += r[9]
for k, (c, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:7.2f} ms x{c:2d} {k[0]} v={k[1]} in={k[2]} out/g={k[3]} x{k[4]} o{k[5]} M={k[6]} K={k[7]} {fl / t / 1e9:6.0f} TF/s | {k[8]}")
