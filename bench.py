"""bench.py -- one "step" = one training step (forward + backward + clip + AdamW, plus the
no-grad metrics the reference logs every step, remfx/models.py:217-256) of a RemFX
removal network on a batch of synthetic white-noise clips (B, 1, 262144) @ 48 kHz.

    python bench.py --gpus N --steps K --warmup W [--workload demucs|tcn] [--batch B]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
one rank per GPU over RCCL; per-GPU work is fixed (weak scaling): every rank trains on its
own B clips and the flat gradient buffer is all-reduced over xGMI every step.

Prints ONE JSON line (rank 0).  metric = audio-seconds processed per second (whole job).
"""
import argparse
import json
import os

os.environ.setdefault("RFX_STRICT_NATIVE", "1")     # an op without a HIP kernel raises instead of running through torch-ROCm
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CLIP = 262144
SR = 48000
PEAK_F32_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA; the bf16x3 mode issues 3 bf16 MFMA flops per algorithmic flop
DTYPES = {"f32": "f32", "bf16x3": "f32 via bf16x3 split MFMA (fp32 accumulate)", "bf16": "bf16 (fp32 accumulate)"}
PEAK_HBM_GBS = 8000.0


def build_model(workload, device):
    from remfx_amd import models
    torch.manual_seed(12345)                                 # cfg/config.yaml:7
    if workload == "tcn":                                    # cfg/model/tcn.yaml
        net = models.TCNModel(sample_rate=SR, num_bins=1025, ninputs=1, noutputs=1, nblocks=20,
                              channel_growth=0, channel_width=256, kernel_size=7, stack_size=10,
                              dilation_growth=2, condition=False, latent_dim=2, norm_type="identity",
                              causal=False, estimate_loudness=False)
    elif workload == "demucs":                               # cfg/model/demucs.yaml
        net = models.DemucsModel(sample_rate=SR, sources=["mixture"], audio_channels=1, nfft=4096, channels=48)
    elif workload == "dcunet":                               # cfg/model/dcunet.yaml
        net = models.DCUNetModel(sample_rate=SR, num_bins=1025, architecture="Large-DCUNet-20",
                                 stft_kernel_size=512, fix_length_mode="pad")
    elif workload == "umx":                                  # cfg/model/umx.yaml
        net = models.OpenUnmixModel(n_fft=2048, hop_length=512, n_channels=1, alpha=0.3, sample_rate=SR)
    else:
        raise ValueError(workload)
    model = models.RemFX(lr=1e-4, lr_beta1=0.95, lr_beta2=0.999, lr_eps=1e-6, lr_weight_decay=1e-3,
                         sample_rate=SR, network=net)
    return model.to(device)


def synthetic_batch(batch, rank, device, clip=CLIP):
    g = torch.Generator().manual_seed(12345 + rank)
    x = torch.randn(batch, 1, clip, generator=g) * 0.1       # ~ -20 dB white noise
    y = torch.randn(batch, 1, clip, generator=g) * 0.1
    lab = torch.zeros(batch, 5)
    return tuple(t.to(device) for t in (x, y, lab, lab.clone()))


class KernelTimer:
    """HIP-event timing of the dominant kernel family (forward-family gather-GEMM launches: conv forward, input
    gradients, transposed convs, linear layers) on the stream they are launched on, with the algorithmic FLOPs
    (2*M*K*positions*N) and bytes (every distinct input element / weight read once, every output written once) of each
    launch, so that every launch can be priced against the roofline that binds IT (ridge = peak FLOP/s / peak B/s)."""

    def __init__(self):
        self.launches, self.enabled, self.desc, self.wgrad = [], False, [], []          # (start, end, flops, bytes); plan of each launch

    def install(self):
        from remfx_amd import ops
        orig = ops.gemm_fwd
        timer = self

        def timed(dp, apack, x, out, *a, **kw):
            if not timer.enabled:
                return orig(dp, apack, x, out, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(dp, apack, x, out, *a, **kw)
            e.record()
            p = dp.p
            k = p.extra["n_weight_rows"] + (kw["dp2"].p.extra["n_weight_rows"] if kw.get("dp2") is not None else 0)
            extra = sum(t.numel() for t in (kw.get("res"), kw.get("glu_out")) if t is not None)
            timer.launches.append((s, e, 2.0 * p.M * k * p.OA * p.OB * p.N, 4.0 * (x.numel() + out.numel() + p.M * k + extra)))
            timer.desc.append({"M": p.M, "K": k, "N": p.N, "OA": p.OA, "OB": p.OB, "R": p.R, "x": list(x.shape), "out": list(out.shape),
                               "two_phase": kw.get("dp2") is not None, "extra_elems": extra})
            return r
        ops.gemm_fwd = timed
        import remfx_amd.tcn as tcn_mod
        tcn_mod.ops.gemm_fwd = timed
        orig_w = ops.gemm_wgrad

        def timed_w(dp, x, g, dapack):                     # weight-gradient launches: only listed by --dump-launches
            if not timer.enabled:
                return orig_w(dp, x, g, dapack)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_w(dp, x, g, dapack)
            e.record()
            p = dp.p
            timer.wgrad.append((s, e, {"M": p.M, "K": p.K, "N": p.N, "OA": p.OA, "OB": p.OB, "x": list(x.shape), "g": list(g.shape),
                                       "x_dtype": str(x.dtype), "g_dtype": str(g.dtype),
                                       "flops": 2.0 * p.M * p.K * p.OA * p.OB * p.N,
                                       "bytes": float(x.numel() * x.element_size() + g.numel() * g.element_size())}))
            return r
        ops.gemm_wgrad = timed_w

    def result(self, peak_tflops, peak_gbs):
        """Totals + the split of the family into MFMA-bound and HBM-bound launches (by each launch's own arithmetic
        intensity against the ridge point of the mode's peaks)."""
        ridge = peak_tflops * 1e12 / (peak_gbs * 1e9)
        cls = {"mfma": [0.0, 0.0, 0.0, 0], "hbm": [0.0, 0.0, 0.0, 0]}        # ms, flops, bytes, launches
        for s, e, fl, by in self.launches:
            c = cls["mfma" if fl / by >= ridge else "hbm"]
            c[0] += s.elapsed_time(e); c[1] += fl; c[2] += by; c[3] += 1
        ms = cls["mfma"][0] + cls["hbm"][0]
        return ms, len(self.launches), cls, ridge


def _host_cpu():
    """(physical cores, model name) of the host from /proc/cpuinfo (falls back to os.cpu_count())."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return (len(cores) or (os.cpu_count() or 1)), model


def cpu_baseline(workload):
    """The CPU oracle (pure-torch restatement of the reference) timed on the host cores: forward + loss +
    backward of the removal network on ONE short clip (bounded: tens of seconds at most).  Threads are capped
    at 32: torch's CPU kernels get slower, not faster, when spread over every hardware thread of the host."""
    from oracle import ref_dcunet, ref_hdemucs, ref_losses, ref_tcn, ref_umx
    from oracle.ref_utils import causal_crop
    phys, cpu_model = _host_cpu()
    cores = min(phys, 32)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    torch.manual_seed(0)
    T = {"tcn": 16384, "demucs": CLIP, "dcunet": 32768, "umx": 65536}[workload]     # headline: one full 262144-sample clip
    x, y = torch.randn(1, 1, T, generator=g) * 0.1, torch.randn(1, 1, T, generator=g) * 0.1
    if workload == "tcn":
        sd = {k: v.requires_grad_(True) for k, v in ref_tcn.tcn_init_state_dict(1, 1, 20, 256, 7).items()}
        fwd = lambda: ref_tcn.tcn_forward(x, sd, 20)
        params = list(sd.values())
    elif workload == "demucs":
        net = ref_hdemucs.HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48)
        fwd = lambda: net(x).squeeze(1)
        params = list(net.parameters())
    elif workload == "dcunet":
        net = ref_dcunet.DCUNet(stft_kernel_size=512, fix_length_mode="pad")
        fwd = lambda: net(x.squeeze(1))
        params = list(net.parameters())
    else:
        net = ref_umx.OpenUnmix(nb_bins=1025, nb_channels=1)
        fwd = lambda: ref_umx.separator(net, x).squeeze(1)
        params = list(net.parameters())
    def one():
        for p in params:
            p.grad = None
        t0 = time.time()
        out = fwd()
        tgt = causal_crop(y, out.shape[-1]) if out.shape[-1] < y.shape[-1] else y
        ref_losses.removal_loss(out, tgt).backward()
        return time.time() - t0
    one()                                            # warm-up (allocator, MKL / oneDNN primitive caches)
    times = sorted(one() for _ in range(3))
    dt = times[1]                                    # median of three timed steps
    return {"value": round(T / SR / dt, 4), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
            "cpu_model": cpu_model, "physical_cores": phys,
            "sample": f"oracle {workload} forward + MRSTFT/L1 loss + backward, 1 clip x {T} samples, 1 warm-up + median of 3 "
                      f"timed steps ({times[0]:.1f} / {times[1]:.1f} / {times[2]:.1f} s), {cores} torch threads"}


def bench_chain(args, rank, world, device):
    """BASELINE config 5: RemFX-detect chain inference (Cnn14 detector + Demucs x2 + DCUNet x3 removal
    networks, cfg/exp/remfx_detect.yaml), inference only, 16 clips per GPU; random-init weights under the
    fixed seed (no checkpoints offline), every clip takes the chain its detected labels select."""
    from remfx_amd import models
    from remfx_amd.classifier import Cnn14
    torch.manual_seed(12345)
    batch = args.batch or 16
    mk_d = lambda: models.RemFX(1e-4, 0.95, 0.999, 1e-6, 1e-3, SR, models.DemucsModel(
        sample_rate=SR, sources=["mixture"], audio_channels=1, nfft=4096, channels=48)).to(device)
    mk_u = lambda: models.RemFX(1e-4, 0.95, 0.999, 1e-6, 1e-3, SR, models.DCUNetModel(
        sample_rate=SR, num_bins=1025, architecture="Large-DCUNet-20", stft_kernel_size=512,
        fix_length_mode="pad")).to(device).eval()
    nets = {"RandomPedalboardDistortion": mk_d(), "RandomPedalboardCompressor": mk_d(),
            "RandomPedalboardReverb": mk_u(), "RandomPedalboardChorus": mk_u(), "RandomPedalboardDelay": mk_u()}
    cls = models.FXClassifier(3e-4, 1e-3, SR, Cnn14(5, SR, SR, 2048, 512, 128)).to(device).eval()
    order = ["RandomPedalboardDistortion", "RandomPedalboardCompressor", "RandomPedalboardReverb",
             "RandomPedalboardChorus", "RandomPedalboardDelay"]
    chain = models.RemFXChainInference(nets, SR, 1025, order, classifier=cls).to(device).eval()
    data = synthetic_batch(batch, rank, device)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    with torch.no_grad():
        for i in range(args.warmup):
            chain.test_step(data, i)
        fence()
        t0 = time.time()
        for i in range(args.steps):
            chain.test_step(data, i)
        fence()
    dt = time.time() - t0
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    if rank != 0:
        return
    dt = float(t)
    napplied = float(chain.last_labels.sum())
    print(json.dumps({
        "metric": "audio-seconds/sec chain inference (whole job)", "value": round(world * batch * CLIP / SR * args.steps / dt, 3),
        "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPES[args.gemm],
        "data": "synthetic", "config": {"workload": "RemFX-detect chain inference (+exp=remfx_detect), inference only",
                                        "clips_per_gpu": batch, "clip_samples": CLIP, "sample_rate": SR,
                                        "removal_model_applications_per_step": napplied, "parallelism": f"dp{world}"}}))


def bench_demucs_fwd(args, rank, world, device):
    """The north_star's own sub-metric: STFT + Hybrid Demucs FORWARD (model.sample: _spec, both U-Net branches, _ispec) on
    64 x 262144-sample clips, no loss / backward.  Algorithmic work: 117.2 GFLOP and 396 MB of fp32
    layer-boundary traffic per clip + 334 MB of weights per pass -> both roofline fractions are reported."""
    batch = args.batch or 64
    model = build_model("demucs", device).eval()
    x = synthetic_batch(batch, rank, device)[0]

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    with torch.no_grad():
        for _ in range(args.warmup):
            model.model.sample(x)
        fence()
        t0 = time.time()
        for _ in range(args.steps):
            out = model.model.sample(x)
        fence()
    dt = time.time() - t0
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    if rank != 0:
        return
    dt = float(t) / args.steps
    flops = batch * (117.2e9 + 0.13e9)
    nbytes = batch * 396e6 + 334e6
    mfma_peak = {"f32": PEAK_F32_TFLOPS, "bf16x3": PEAK_BF16_TFLOPS / 3.0, "bf16": PEAK_BF16_TFLOPS}[args.gemm]
    f_hbm, f_mfma = nbytes / dt / 1e9 / PEAK_HBM_GBS, flops / dt / 1e12 / mfma_peak
    bound = "hbm" if f_hbm >= f_mfma else "mfma"
    print(json.dumps({
        "metric": "audio-seconds/sec STFT + Demucs forward (whole job)", "value": round(world * batch * CLIP / SR / dt, 3),
        "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPES[args.gemm], "data": "synthetic",
        "config": {"workload": "Hybrid Demucs (cfg/model/demucs.yaml) forward only: STFT -> U-Net -> iSTFT, DemucsModel.sample",
                   "clips_per_gpu": batch, "clip_samples": CLIP, "sample_rate": SR, "parallelism": f"dp{world}",
                   "output_rms": round(float(out.float().pow(2).mean().sqrt()), 6)},
        "roofline": {"bound": bound, "achieved": round(nbytes / dt / 1e9, 1) if bound == "hbm" else round(flops / dt / 1e12, 2),
                     "peak": PEAK_HBM_GBS if bound == "hbm" else mfma_peak, "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                     "frac": round(max(f_hbm, f_mfma), 4), "frac_hbm": round(f_hbm, 4), "frac_mfma": round(f_mfma, 4),
                     "algorithmic_bytes_per_pass": nbytes, "algorithmic_flops_per_pass": flops, "traffic": None}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("RFX_WORKLOAD", "demucs"),
                    choices=["demucs", "tcn", "dcunet", "umx", "chain", "demucs_fwd"])
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (0 = the BASELINE config's batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-launches", default="", help="write the per-launch plan / algorithmic work / event time list of the timed steps (JSON)")
    ap.add_argument("--union-ranks", type=int, default=0,
                    help="single process only: train on the concatenation of the synthetic batches ranks 0..N-1 would "
                         "get (N x --batch clips) -- the reference point of the data-parallel equivalence test")
    ap.add_argument("--gemm", default=os.environ.get("RFX_GEMM_PREC", "bf16"), choices=["bf16", "bf16x3", "f32"],
                    help="MFMA arithmetic of the gather-GEMMs.  bf16 (default): operands rounded to bf16, fp32 accumulation -- "
                         "trainer.precision=bf16-mixed, the precision BASELINE.json's headline config names; bf16x3: fp32 "
                         "operands split hi + lo, 3 MFMAs per product (HDemucs forward within 4e-6 RMS of the fp32 oracle); "
                         "f32: exact fp32 MFMA")
    args = ap.parse_args()
    if args.workload == "tcn" and args.warmup < 2:
        # 32 x 262144 TCN activations fill ~190 of the 288 GB: the caching allocator settles only after its one
        # "free everything and retry" event at the start of step 2 (a 5.5 s host stall that is not part of a step)
        args.warmup = 2

    from remfx_amd import ddp, ops
    ops.set_gemm_precision(args.gemm)
    rank, local, world = ddp.init_from_env()
    assert world == args.gpus or world == 1, (world, args.gpus)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # BASELINE.json configs: Demucs 64 clips/GPU (headline), TCN 32, DCUNet 32 over 8 GPUs = 4/GPU, UMX 4
    batch = args.batch or {"tcn": 32, "demucs": 64, "dcunet": 4, "umx": 4, "chain": 16, "demucs_fwd": 64}[args.workload]

    if args.workload == "chain":
        return bench_chain(args, rank, world, device)
    if args.workload == "demucs_fwd":
        return bench_demucs_fwd(args, rank, world, device)
    model = build_model(args.workload, device)
    cfg = model.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    ddp.broadcast_parameters(opt.flat.data)
    sync = ddp.GradSync(opt.flat)
    data = synthetic_batch(batch, rank, device)
    if args.union_ranks > 1:
        parts = [synthetic_batch(batch, r, device) for r in range(args.union_ranks)]
        data = tuple(torch.cat([p[i] for p in parts], 0) for i in range(4))
    timer = KernelTimer()
    timer.install()

    def step(i):
        opt.zero_grad()
        loss = model.training_step(data, i)
        loss.backward()
        pre = sync.finish()
        opt.step(clip_norm=10.0, grad_prescale=pre)           # cfg/config.yaml:119
        sched.step()
        return loss

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    timer.enabled = True
    t0 = time.time()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    fence()
    dt = time.time() - t0
    timer.enabled = False
    from remfx_amd import lstm as _lstm
    if _lstm.error_flag():                      # a bounded cluster-exchange spin timed out: results are invalid
        raise RuntimeError("LSTM recurrence kernel reported a spin time-out")
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t)
    if rank != 0:
        return
    audio_s = world * batch * CLIP / SR * args.steps
    # algorithmic (fp32-equivalent) FLOP/s; in bf16x3 mode the matrix pipe executes 3x that in bf16
    peak = {"f32": PEAK_F32_TFLOPS, "bf16x3": PEAK_BF16_TFLOPS / 3.0, "bf16": PEAK_BF16_TFLOPS}[args.gemm]
    kname = {"f32": "gemm_fwd_kernel<R> (gather-GEMM, v_mfma_f32_32x32x2_f32)",
             "bf16x3": "gemm_tap_kernel<R,1> + gemm_tap_stream_kernel<1,..> (tap-major gather-GEMM, 3 x v_mfma_f32_32x32x16_bf16 per K step)",
             "bf16": "gemm_tap_kernel<R,2> + gemm_tap_stream_kernel<2,..> (tap-major gather-GEMM, 1 x v_mfma_f32_32x32x16_bf16 per K step)"}[args.gemm]
    kms, klaunches, cls, ridge = timer.result(peak, PEAK_HBM_GBS)
    if args.dump_launches:         # per-launch plan + algorithmic work + event time, in launch order (scripts/join_launch_pmc.py)
        json.dump([dict(d, ms=s.elapsed_time(e), flops=fl, bytes=by) for d, (s, e, fl, by) in zip(timer.desc, timer.launches)],
                  open(args.dump_launches, "w"))
        json.dump([dict(d, ms=s.elapsed_time(e)) for s, e, d in timer.wgrad], open(args.dump_launches + ".wgrad.json", "w"))
    tot_fl = cls["mfma"][1] + cls["hbm"][1]
    tot_by = cls["mfma"][2] + cls["hbm"][2]
    f_mfma = tot_fl / (kms * 1e-3) / 1e12 / peak if kms > 0 else 0.0
    f_hbm = tot_by / (kms * 1e-3) / 1e9 / PEAK_HBM_GBS if kms > 0 else 0.0
    bound = "hbm" if f_hbm >= f_mfma else "mfma"

    def _cls(name):
        ms, fl, by, n = cls[name]
        if not n:
            return {"launches": 0}
        return {"launches": n, "ms_per_step": round(ms / args.steps, 3), "tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                "gbs": round(by / (ms * 1e-3) / 1e9, 1),
                "frac": round((fl / (ms * 1e-3) / 1e12 / peak) if name == "mfma" else (by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS), 4)}
    # roofline.traffic: HBM bytes per launch of the family from the committed PMC passes of this same command
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, scripts/collect_pmc.py)
    traffic = None
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_demucs_b64_pmc_traffic_{args.gemm}.json")))   # latest round / pass last
    pmc = cands[-1] if cands else ""
    if args.workload == "demucs" and batch == 64 and pmc:
        ks = json.load(open(pmc))["kernels"]
        fam = [v for k, v in ks.items() if k.startswith(("gemm_fwd_kernel", "gemm_tap_kernel", "gemm_tap_stream_kernel"))]
        n = sum(v["launches_per_step"] for v in fam)
        if n:
            traffic = round(sum(v["bytes_per_step"] for v in fam) / n)
    out = {
        "metric": "audio-seconds/sec fwd+bwd (whole job)", "value": round(audio_s / dt, 3),
        "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPES[args.gemm],
        "data": "synthetic",
        "config": {"workload": {"tcn": "TCN (cfg/model/tcn.yaml) train step, +exp=reverb model=tcn",
                                "demucs": "Hybrid Demucs (cfg/model/demucs.yaml) train step, +exp=chorus_aug model=demucs",
                                "dcunet": "DCUNet Large-DCUNet-20 (cfg/model/dcunet.yaml) train step, +exp=5-5_full model=dcunet",
                                "umx": "Open-Unmix (cfg/model/umx.yaml) train step, +exp=distortion model=umx"}[args.workload],
                   "clips_per_gpu": batch, "clip_samples": CLIP, "sample_rate": SR,
                   "step": "fwd + MRSTFT+100*L1 loss + bwd + clip 10 + AdamW + per-step metrics",
                   "parallelism": f"dp{world}", "final_loss": round(float(loss.detach()), 5),
                   # data-parallel bookkeeping: collective backend ("nccl" = RCCL over xGMI) and a checksum of the
                   # parameters after the timed steps (equal across replicas; tests compare it with a 1-rank run)
                   "dist_backend": torch.distributed.get_backend() if world > 1 else None, "ranks": world,
                   "param_abs_sum": float(opt.flat.data.double().abs().sum())},
        # the dominant kernel FAMILY (all forward-type gather-GEMM launches of the step).  `bound` is the roofline the family
        # as a whole sits closer to: with bf16 operands on fp32 storage almost every Demucs layer has an arithmetic intensity
        # below the ridge (2500 TF/s / 8 TB/s = 312 flop/B) and is priced against HBM; `by_bound` prices the launches on
        # either side of the ridge separately.
        "roofline": {"bound": bound, "kernel": kname,
                     "achieved": round(tot_by / (kms * 1e-3) / 1e9, 1) if bound == "hbm" else round(tot_fl / (kms * 1e-3) / 1e12, 2),
                     "peak": PEAK_HBM_GBS if bound == "hbm" else round(peak, 1), "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                     "frac": round(max(f_hbm, f_mfma), 4), "frac_mfma": round(f_mfma, 4), "frac_hbm": round(f_hbm, 4),
                     "ridge_flop_per_byte": round(ridge, 1), "by_bound": {"mfma": _cls("mfma"), "hbm": _cls("hbm")},
                     "traffic": traffic, "algorithmic_bytes_per_launch": round(tot_by / max(klaunches, 1)),
                     "algorithmic_flops_per_launch": round(tot_fl / max(klaunches, 1)),
                     "launches": klaunches, "avg_launch_ms": round(kms / max(klaunches, 1), 4),
                     "share_of_step": round(kms / (dt * 1e3), 3)},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
