"""Dev tool: one Demucs training step with EVERY C-ABI launch timed (synchronising, single stream) and printed in program
order with its geometry and the Python call site -- the sequence shows which layer a launch belongs to, so the step can be
budgeted per module (DConv depth-layers, rewrite convs, encoders / decoders, BLSTM, attention, loss, optimiser).
    python scripts/seq_profile.py [B] > seq.txt"""
import sys, os, time, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import _lib, ops
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
L = _lib.lib()
VARIANT = L.rfx_gemm_fwd_variant
ops.set_gemm_precision(os.environ.get("RFX_GEMM_PREC", "bf16"))
ops.GradSink.MODE = "main"
rec = []
FILES = ("hdemucs.py", "nnops.py", "ops.py", "stft.py", "losses.py", "lstm.py", "models.py", "optim.py", "clchain.py", "clast.py")


def site():
    out = []
    for f in traceback.extract_stack()[:-2]:
        fn = os.path.basename(f.filename)
        if fn in FILES:
            out.append(f"{fn[:-3]}:{f.name}:{f.lineno}")
    return ">".join(out[-5:])


def wrap(name):
    fn = getattr(L, name)

    def timed(*args):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = fn(*args)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        info = ""
        if name in ("rfx_gemm_fwd", "rfx_gemm_wgrad"):
            d = args[0]._obj
            info = f"N={d.N} M={d.M} K={d.K} P={d.OA}x{d.OB} S=({d.SA},{d.SB}) in16={d.in_bf16} out16={d.out_bf16}"
            if name == "rfx_gemm_fwd":
                a2 = args[6]
                pv = getattr(args[11], "value", args[11])
                v = VARIANT(args[0], args[5], int(bool(getattr(a2, "value", a2))), pv)
                info += f" v={v >> 4}/{v & 15}"
        elif name == "rfx_cl_conv":
            d = args[0]._obj
            info = f"N={d.N} M={d.M} K={d.NTR}x{d.NTC}x{d.NCH * 16 * d.KS} rows {d.IA}->{d.OA} BM={d.BM} mode={d.mode}"
        elif name == "rfx_cl_wgrad":
            d = args[0]._obj
            info = f"N={d.N} M={d.M} Cq={d.Cq} taps={d.NTR}x{d.NTC} rows {d.OA}/{d.IA} S={d.S} WK={d.WK}"
        elif name.startswith("rfx_groupnorm"):
            a = args[3:7] if name.endswith("fwd") or "fwd" in name else args[6:10]
            info = "N,C,S,G=" + ",".join(str(getattr(v, "value", v)) for v in a)
        elif name.startswith("rfx_fft") or name == "rfx_stft_pair_loss":
            d = args[0]._obj
            info = f"R={d.R} T={d.T} nfft={d.n_fft} hop={d.hop} mode={d.mode}"
        else:
            info = " ".join(str(getattr(v, "value", v)) for v in args if isinstance(getattr(v, "value", v), int) and not isinstance(v, bool) and abs(getattr(v, "value", v)) < (1 << 40))[:80]
        rec.append((name, info, dt, site()))
        return rc
    return timed


model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
data = bench.synthetic_batch(B, 0, dev)
for it in range(3):
    if it == 2:
        for name in _lib.SIGNATURES:
            if name not in ("rfx_abi_version", "rfx_gemm_pick_r", "rfx_gemm_fwd_variant", "rfx_dconv_layer_ok"):
                setattr(L, name, wrap(name))
        torch.cuda.synchronize(); T0 = time.perf_counter()
    opt.zero_grad()
    loss = model.training_step(data, 0)
    rec.append(("---- backward", "", 0.0, ""))
    loss.backward()
    rec.append(("---- optimiser", "", 0.0, ""))
    opt.step(clip_norm=10.0)
    torch.cuda.synchronize()
wall = (time.perf_counter() - T0) * 1e3
tot = sum(r[2] for r in rec)
print(f"# {len(rec)} native launches, {tot:.1f} ms inside launches, {wall:.1f} ms wall incl. torch glue and syncs (B={B})")
agg = collections.defaultdict(lambda: [0, 0.0])
for n, i, t, s in rec:
    agg[n][0] += 1; agg[n][1] += t
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"#  {n:26s} {c:5d} calls {t:9.2f} ms")
for k, (n, i, t, s) in enumerate(rec):
    print(f"{k:5d} {t:8.3f} {n:24s} {i} | {s}")
