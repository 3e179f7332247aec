"""Per-launch profile of one training step (dev tool): wraps every C-ABI entry point with
synchronising timers and prints the launches sorted by time, with their geometry."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import _lib
import bench

workload = sys.argv[1] if len(sys.argv) > 1 else "demucs"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
L = _lib.lib()
records = []


def wrap(name):
    fn = getattr(L, name)

    def timed(*args):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = fn(*args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        info = ""
        if name in ("rfx_gemm_fwd", "rfx_gemm_wgrad"):
            d = args[0]._obj
            info = f"N={d.N} M={d.M} K={d.K} P={d.OA}x{d.OB} S=({d.SA},{d.SB}) osa={d.out_sa}*{d.out_as} osb={d.out_sb}*{d.out_bs}"
            fl = 2.0 * d.N * d.M * d.K * d.OA * d.OB
            by = 4.0 * d.N * d.OA * d.OB * (d.M + d.K / max(1, (d.K // max(d.M, 1)) if False else 1))
            info += f" {fl / dt / 1e9:6.1f} TF/s  out {4.0 * d.N * d.M * d.OA * d.OB / 1e6:7.1f} MB"
        elif name.startswith("rfx_groupnorm"):
            if name.endswith("fwd"):
                info = f"N={args[3]} C={args[4]} S={args[5]} G={args[6]} mode={args[8]}"
            else:
                info = f"N={args[6]} C={args[7]} S={args[8]} G={args[9]} mode={args[10]}"
        elif name.startswith("rfx_fft"):
            d = args[0]._obj
            info = f"R={d.R} T={d.T} nfft={d.n_fft} hop={d.hop} frames={d.frames_out} mode={d.mode}"
        records.append((name, info, dt))
        return rc
    return timed


from remfx_amd import ops
ops.set_gemm_precision(os.environ.get("RFX_GEMM_PREC", "bf16x3"))
model = bench.build_model(workload, dev)
cfg = model.configure_optimizers()
opt = cfg["optimizer"]
data = bench.synthetic_batch(B, 0, dev)
for it in range(2):
    if it == 1:
        for name in _lib.SIGNATURES:
            if name not in ("rfx_abi_version", "rfx_gemm_pick_r"):
                setattr(L, name, wrap(name))
    opt.zero_grad()
    loss = model.training_step(data, 0)
    loss.backward()
    opt.step(clip_norm=10.0)
    torch.cuda.synchronize()
tot = sum(r[2] for r in records)
print(f"total native launch time {tot:.1f} ms over {len(records)} launches (B={B})")
agg = collections.defaultdict(lambda: [0, 0.0])
for n, i, t in records:
    agg[n][0] += 1; agg[n][1] += t
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:24s} {c:5d} calls {t:9.2f} ms {100 * t / tot:5.1f}%")
print("--- top launches")
for n, i, t in sorted(records, key=lambda r: -r[2])[:int(os.environ.get("RFX_TOP", "70"))]:
    print(f"{t:8.3f} ms  {n:20s} {i}")
