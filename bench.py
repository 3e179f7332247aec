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

T_PROC0 = time.time()                                # phases_s of the JSON line are measured from here
PHASES = {}

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from remfx_amd import _lib as _lib_mod  # noqa: E402

CLIP = 262144
SR = 48000
PEAK_F32_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA; the bf16x3 mode issues 3 bf16 MFMA flops per algorithmic flop
DTYPES = {"f32": "f32", "bf16x3": "f32 via bf16x3 split MFMA (fp32 accumulate)", "bf16": "bf16 (fp32 accumulate)"}
PEAK_HBM_GBS = 8000.0


# ---- output: the FULL record goes to a file (and stderr on request); stdout carries ONE compact JSON line (< 4 KB) ---------------
ALG = {  # algorithmic work of ONE clip, forward; fwd+bwd = 3x flops, 2.5x bytes.  bytes: (fp32 count, 16-bit count)
    "demucs": {"flops_fwd": 117.2e9 + 0.13e9, "bytes_fwd": (396e6, 198e6), "weights": 83.63e6},
    "tcn": {"flops_fwd": 5.14e12, "bytes_fwd": (10.7e9, 10.7e9), "weights": 9.974e6},
    "dcunet": {"flops_fwd": 1705e9, "bytes_fwd": (None, None), "weights": 7.66e6},
    "umx": {"flops_fwd": 2 * 6.5e9, "bytes_fwd": (None, None), "weights": 6.3e6},
}


def step_roofline(workload, gemm, batch, sec_per_step, peak_tflops, pmc_total_bytes):
    """Whole-step roofline per SURVEY 8(d): achieved_mfma = flops_alg / (t * peak of the arithmetic mode), achieved_hbm =
    bytes_alg / (t * 8.0e12), algorithmic work of forward + backward = 3x the forward flops and 2.5x the forward layer-boundary
    bytes (in the width the mode stores activations in: 16-bit in the bf16 mode, fp32 otherwise) + the weights once per pass."""
    a = ALG[workload]
    fl = 3.0 * a["flops_fwd"] * batch
    b32, b16 = a["bytes_fwd"]
    bf = b16 if gemm == "bf16" else b32
    by = (2.5 * bf * batch + 3 * 4.0 * a["weights"]) if bf else None
    r = {"algorithmic_flops": round(fl), "algorithmic_bytes": round(by) if by else None,
         "frac_mfma": round(fl / sec_per_step / 1e12 / peak_tflops, 4),
         "frac_hbm": round(by / sec_per_step / 1e9 / PEAK_HBM_GBS, 4) if by else None,
         "traffic_bytes": pmc_total_bytes,
         "traffic_ratio": round(pmc_total_bytes / by, 2) if (by and pmc_total_bytes) else None}
    return r


def emit(out, tag):
    """Write the full record to gpurun_out/ (scratch on the GPU box; falls back to profiles/) and print the compact line."""
    full_path = None
    for d in (os.environ.get("RFX_BENCH_FULL_DIR"), os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")):
        if not d:
            continue
        try:
            os.makedirs(d, exist_ok=True)
            full_path = os.path.join(d, f"bench_full_{tag}.json")
            json.dump(out, open(full_path, "w"))
            break
        except OSError:
            full_path = None
    if os.environ.get("RFX_BENCH_FULL_STDERR"):
        print(json.dumps(out), file=sys.stderr)
    keep_roof = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_mfma", "frac_hbm", "traffic", "algorithmic_bytes_per_launch",
                 "algorithmic_flops_per_launch", "avg_launch_us", "launches_per_step", "share_of_step", "frac_source")
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data", "preheat_steps") if k in out}
    cfg = out.get("config", {})
    c["config"] = {k: cfg[k] for k in ("workload", "clips_per_gpu", "global_batch", "clip_samples", "parallelism", "final_loss",
                                       "rccl_ranks", "ranks", "dist_backend", "exposed_allreduce_ms", "param_abs_sum", "streams",
                                       "removal_model_applications_per_step") if k in cfg}
    roof = out.get("roofline")
    c["roofline"] = {k: roof[k] for k in keep_roof if k in roof} if roof else None
    if "step_roofline" in out:
        c["step_roofline"] = out["step_roofline"]
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {k: (v if k != "sample" else v[:200]) for k, v in cb.items() if k in ("value", "unit", "cores", "kind", "sample")}
    al = out.get("also")
    if al:
        ca = {}
        if "train_step_bf16x3" in al:
            ca["train_step_bf16x3_ms"] = al["train_step_bf16x3"]["ms_per_step"]
        if "train_step_f32" in al:
            ca["train_step_f32_ms"] = al["train_step_f32"]["ms_per_step"]
        f = al.get("demucs_fwd")
        if f:
            ca["demucs_fwd"] = {"ms_per_pass": f["ms_per_pass"], "frac_hbm": f["frac_hbm"], "frac_hbm_fp32_bytes": f.get("frac_hbm_fp32_bytes"),
                                "frac_mfma": f["frac_mfma"],
                                "stft_frac_hbm": f["stages"]["stft"]["frac_hbm"], "istft_frac_hbm": f["stages"]["istft"]["frac_hbm"]}
        for k in ("si_sdr_vs_cpu_oracle_db", "rms_vs_cpu_oracle", "parity_clip_samples"):
            if k in al:
                ca[k] = al[k]
        c["also"] = ca
    if "phases_s" in out:
        c["phases_s"] = out["phases_s"]
    c["full_record"] = os.path.relpath(full_path, ROOT) if full_path else None      # every per-kernel table of earlier rounds lives there
    line = json.dumps(c)
    if len(line) > 3900:                         # never let the driver see an unparseable tail again: drop the optional blocks
        for k in ("phases_s", "also"):
            c.pop(k, None)
        line = json.dumps(c)
    print(line, flush=True)


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


def variant_name(code, prec):
    """Kernel instantiation behind an rfx_gemm_fwd_variant() code (csrc/gemm.hip), spelled as rocprofv3 prints it."""
    kind, r = code >> 4, code & 15
    mode = {0: 0, 1: 1, 2: 2}[prec]
    return {0: "gemm_thin_fwd_kernel<M>", 1: f"gemm_fwd_kernel<{r}>", 2: f"gemm_tap_kernel<{r}, {mode}, 0>",
            3: f"gemm_tap_kernel<{r}, 2, 1>", 4: f"gemm_tap_stream_kernel<{mode}, 4, 1>",
            5: f"gemm_tap_stream_kernel<{mode}, 2, 4>", 6: f"gemm_halo_kernel<{r}, 9, 0>", 7: f"gemm_halo_kernel<{r}, 9, 1>",
            8: f"gemm_halo_kernel<{r}, 3, 0>", 9: f"gemm_halo_kernel<{r}, 3, 1>"}.get(kind, f"variant{code}")


def _pmc_key(name):
    """Normalised key of a kernel name from a rocprofv3 CSV / the PMC traffic JSON: template booleans and integers compare
    equal (`<4, 2, false>` == `<4, 2, 0>`)."""
    return name.replace(" ", "").replace("false", "0").replace("true", "1").replace(",0>", ">") if name else name


class KernelTimer:
    """HIP-event timing of every forward-family gather-GEMM launch (conv forward, input gradients, transposed convs, linear
    layers) on the stream it is launched on, with its algorithmic FLOPs (2*M*K*positions*N) and algorithmic BYTES priced by
    the STORED type of each tensor (every distinct gathered input element once, every written output element once, the packed
    weights once in the arithmetic mode's operand width, residual / GLU side tensors once), and the kernel instantiation the
    launcher picked (rfx_gemm_fwd_variant).  Launches are grouped per instantiation; each group is priced against both roofs."""

    def __init__(self, prec):
        self.launches, self.enabled, self.desc, self.wgrad = [], False, [], []   # (start, end, flops, bytes, variant); plan of each launch
        self.prec = prec

    def install(self):
        from remfx_amd import ops
        orig = ops.gemm_fwd
        timer = self
        wbytes = {0: 4.0, 1: 4.0, 2: 2.0}        # packed weight bytes per element: fp32 / bf16 hi + lo / bf16

        def timed(dp, apack, x, out, *a, **kw):
            if not timer.enabled:
                return orig(dp, apack, x, out, *a, **kw)
            ops.TRACE_VARIANT = []
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(dp, apack, x, out, *a, **kw)
            e.record()
            code = ops.TRACE_VARIANT[-1] if ops.TRACE_VARIANT else -1
            ops.TRACE_VARIANT = None
            p = dp.p
            prec = dp.fwd_prec()
            k = p.extra["n_weight_rows"] + (kw["dp2"].p.extra["n_weight_rows"] if kw.get("dp2") is not None else 0)
            extra = sum(t.numel() * t.element_size() for t in (kw.get("res"), kw.get("glu_out"), kw.get("in2")) if t is not None)
            written = p.N * p.M * p.OA * p.OB                       # output elements this launch stores
            nbytes = float(x.numel() * x.element_size() + written * out.element_size() + p.M * k * wbytes[prec] + extra)
            timer.launches.append((s, e, 2.0 * p.M * k * p.OA * p.OB * p.N, nbytes, variant_name(code, prec)))
            timer.desc.append({"M": p.M, "K": k, "N": p.N, "OA": p.OA, "OB": p.OB, "R": p.R, "x": list(x.shape), "out": list(out.shape),
                               "x_dtype": str(x.dtype), "out_dtype": str(out.dtype), "kernel": variant_name(code, prec),
                               "two_phase": kw.get("dp2") is not None, "extra_bytes": extra})
            return r
        ops.gemm_fwd = timed
        import remfx_amd.tcn as tcn_mod
        tcn_mod.ops.gemm_fwd = timed
        orig_w = ops.gemm_wgrad

        def timed_w(dp, x, g):                             # weight-gradient launches: only listed by --dump-launches
            if not timer.enabled:
                return orig_w(dp, x, g)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_w(dp, x, g)
            e.record()
            p = dp.p
            timer.wgrad.append((s, e, {"M": p.M, "K": p.K, "N": p.N, "OA": p.OA, "OB": p.OB, "x": list(x.shape), "g": list(g.shape),
                                       "x_dtype": str(x.dtype), "g_dtype": str(g.dtype),
                                       "flops": 2.0 * p.M * p.K * p.OA * p.OB * p.N,
                                       "bytes": float(x.numel() * x.element_size() + g.numel() * g.element_size())}))
            return r
        ops.gemm_wgrad = timed_w
        # channels-last family (round 5): the same event bracket around clast.conv / clast.wgrad and the fused DConv launches
        from remfx_amd import clast, cldconv

        def cl_conv_name(f, halo, mode):                  # the rocprof name: <epilogue mode, RW, NT, WM, NTC, KS, DA, DB, HALO>
            rw, nt, wm = {192: (3, 2, 2), 96: (3, 1, 1), 64: (2, 1, 1), 32: (1, 1, 1)}[f.BM]
            mt = rw * wm
            db = (4 if mt >= 6 else 6) if (f.NTC == 3 and f.KS == 1) else ((3 if mt >= 6 else 4) if f.KS == 2 else 6)
            return f"cl_conv_kernel<{clast.EPI[mode]}, {rw}, {nt}, {wm}, {f.NTC}, {f.KS}, 2, {db}, {'true' if halo else 'false'}>"
        orig_c, orig_cw = clast.conv, clast.wgrad

        def timed_c(form, apack, x, N, IA, IB, OA, mode, **kw):
            if not timer.enabled:
                return orig_c(form, apack, x, N, IA, IB, OA, mode, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_c(form, apack, x, N, IA, IB, OA, mode, **kw)
            e.record()
            fl = 2.0 * N * OA * IB * form.M * form.NTR * form.NTC * form.Cin
            by = 2.0 * (N * IA * IB * form.Cin + sum(t.numel() for t in (kw.get("out0"), kw.get("out1"), kw.get("aux0"), kw.get("res"))
                                                        if t is not None) + apack.numel())
            timer.launches.append((s, e, fl, by, cl_conv_name(form, form.NTC > 1 or form.db0 != 0, mode)))
            timer.desc.append({"M": form.M, "K": form.NTR * form.NTC * form.Cin, "N": N, "OA": OA, "OB": IB, "kernel": "cl_conv", "mode": mode})
            return r

        def timed_cw(form, p, q, N, OA, IA, B, dw, db=None, **kw):
            if not timer.enabled:
                return orig_cw(form, p, q, N, OA, IA, B, dw, db, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_cw(form, p, q, N, OA, IA, B, dw, db, **kw)
            e.record()
            fl = 2.0 * N * OA * B * form.M * form.NTR * form.NTC * form.Cq
            by = 2.0 * (N * OA * B * form.M + N * IA * B * form.Cq) + 4.0 * form.wn
            timer.launches.append((s, e, fl, by, f"cl_wgrad_kernel<{form.RW}, {form.WK}, {form.PW if B % form.PW == 0 else 64}>"))
            timer.desc.append({"M": form.M, "K": form.NTR * form.NTC * form.Cq, "N": N, "OA": OA, "OB": B, "kernel": "cl_wgrad"})
            return r
        clast.conv, clast.wgrad = timed_c, timed_cw
        L = _lib_mod.lib()
        for nm, bwd in (("rfx_cl_dconv_fwd", False), ("rfx_cl_dconv_bwd", True)):
            def mk(nm=nm, bwd=bwd):
                fn = getattr(L, nm)

                def timed_d(dref, *a):
                    if not timer.enabled:
                        return fn(dref, *a)
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    rc = fn(dref, *a)
                    e.record()
                    d = dref._obj
                    hp = -(-d.H // 16) * 16
                    pos = float(d.S) * 256
                    fl = 2.0 * pos * (3 * d.C * d.H + d.H * 2 * d.C) * (3 if bwd else 1)
                    by = pos * 2.0 * ((d.C + 2 * hp + d.C + 2 * d.C + hp) if bwd else (2 * d.C + (2 * hp if d.a else 0)))
                    passes = d.TPS > 1 or (bwd and d.C != 48)            # several kernels inside one bracket: no single rocprof name
                    nm_k = (f"cl_dconv_bwdp_kernel<{d.C}, {d.H}, 1+2> (+ means, dh)" if (bwd and passes) else
                            f"cl_dconv_bwd_kernel<{d.C}, {d.H}>" if bwd else
                            f"cl_dconv_fwd_kernel<{d.C}, {d.H}, 1+2+3> (+ stats)" if passes else f"cl_dconv_fwd_kernel<{d.C}, {d.H}, 0>")
                    timer.launches.append((s, e, fl, by, nm_k))
                    timer.desc.append({"M": 2 * d.C, "K": d.H, "N": d.S, "OA": 1, "OB": 256, "kernel": nm})
                    return rc
                return timed_d
            setattr(L, nm, mk())

    def result(self, peak_tflops, peak_gbs):
        """Totals + the split of the family into MFMA-bound and HBM-bound launches (by each launch's own arithmetic
        intensity against the ridge point of the mode's peaks) + per-instantiation groups."""
        ridge = peak_tflops * 1e12 / (peak_gbs * 1e9)
        cls = {"mfma": [0.0, 0.0, 0.0, 0], "hbm": [0.0, 0.0, 0.0, 0]}        # ms, flops, bytes, launches
        kern = {}
        for s, e, fl, by, name in self.launches:
            ms = s.elapsed_time(e)
            c = cls["mfma" if fl / by >= ridge else "hbm"]
            c[0] += ms; c[1] += fl; c[2] += by; c[3] += 1
            k = kern.setdefault(name, [0.0, 0.0, 0.0, 0])
            k[0] += ms; k[1] += fl; k[2] += by; k[3] += 1
        ms = cls["mfma"][0] + cls["hbm"][0]
        return ms, len(self.launches), cls, ridge, kern


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
    T = {"tcn": 16384, "demucs": CLIP, "dcunet": CLIP, "umx": 65536}[workload]     # Demucs / DCUNet (configs 3, 4): one full 262144-sample clip
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
    t_w = one()                                      # warm-up (allocator, MKL / oneDNN primitive caches)
    times = sorted(one() for _ in range(3 if t_w < 6.0 else 1))
    dt = times[len(times) // 2]                      # median of the timed steps (three, or one when a step takes more than 6 s)
    times = (times * 3)[:3]
    note = ""
    if phys > cores and dt <= 5.0:
        # BASELINE.md section 4 asks for the host's physical cores: time that too (one warm-up + one step) and report the FASTER of
        # the two thread counts -- torch's CPU kernels usually get slower, not faster, beyond ~32 threads on these hosts
        torch.set_num_threads(phys)
        dt_all = one()                               # ONE step (thread-pool spin-up included): a bound, not a tuned number
        note = f"; all {phys} physical cores: {dt_all:.1f} s per step"
        if dt_all < dt:
            dt, cores = dt_all, phys
        torch.set_num_threads(cores)
    return {"value": round(T / SR / dt, 4), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
            "cpu_model": cpu_model, "physical_cores": phys,
            "sample": f"oracle {workload} forward + MRSTFT/L1 loss + backward, 1 clip x {T} samples, 1 warm-up + median of 3 "
                      f"timed steps ({times[0]:.1f} / {times[1]:.1f} / {times[2]:.1f} s) on {min(phys, 32)} torch threads{note}"}


def cpu_baseline_chain(labels, order):
    """CPU oracle timing of the chain (BASELINE config 5), bounded: the detector (oracle Cnn14 on one full clip) and ONE inference
    forward each of the two removal architectures (oracle Hybrid Demucs / DCUNet, one full clip), combined with the number of removal
    applications per clip the GPU run's detected labels selected -- the chain's cost on the CPU is that linear combination."""
    from oracle import ref_cnn14, ref_dcunet, ref_hdemucs
    phys, cpu_model = _host_cpu()
    cores = min(phys, 32)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    x = torch.randn(1, 1, CLIP, generator=torch.Generator().manual_seed(0)) * 0.1
    hd = ref_hdemucs.HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48).eval()
    du = ref_dcunet.DCUNet(stft_kernel_size=512, fix_length_mode="pad").eval()
    sd = ref_cnn14.cnn14_init_state_dict()

    def t_of(fn):
        fn()
        ts = []
        for _ in range(2):
            t0 = time.time(); fn(); ts.append(time.time() - t0)
        return min(ts)
    with torch.no_grad():
        t_d = t_of(lambda: hd(x))
        t_u = t_of(lambda: du(x.squeeze(1)))
        t_c = t_of(lambda: ref_cnn14.cnn14_forward(x, sd, SR, 2048, 512, 128))
    B = labels.shape[0]
    n_d = float(labels[:, :2].sum()) / B             # distortion, compressor -> Demucs;  reverb, chorus, delay -> DCUNet (cfg/exp/remfx_detect.yaml)
    n_u = float(labels[:, 2:].sum()) / B
    per_clip = t_c + n_d * t_d + n_u * t_u
    return {"value": round(CLIP / SR / per_clip, 4), "unit": "audio-seconds/sec", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "physical_cores": phys,
            "sample": f"oracle inference forwards on one {CLIP}-sample clip, best of 2 after a warm-up: detector {t_c:.2f} s, Hybrid Demucs "
                      f"{t_d:.2f} s, DCUNet {t_u:.2f} s; combined with the {n_d:.2f} Demucs + {n_u:.2f} DCUNet applications per clip the GPU "
                      f"run's labels selected; {cores} torch threads"}


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
    from remfx_amd import ops
    timer = KernelTimer(ops.PREC_NAMES[args.gemm])
    timer.install()

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    with torch.no_grad():
        for i in range(args.warmup):
            chain.test_step(data, i)
        fence()
        timer.enabled = True
        t0 = time.time()
        for i in range(args.steps):
            chain.test_step(data, i)
        fence()
    dt = time.time() - t0
    timer.enabled = False
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    if rank != 0:
        return
    dt = float(t)
    napplied = float(chain.last_labels.sum())
    # roofline of the dominant gather-GEMM instantiation of the chain (detector at fp32 parity = bf16x3 inside at_least_fp32_parity,
    # removal networks in the session mode): algorithmic flops / bytes per launch over its event-timed duration
    import glob
    peak = {"f32": PEAK_F32_TFLOPS, "bf16x3": PEAK_BF16_TFLOPS / 3.0, "bf16": PEAK_BF16_TFLOPS}[args.gemm]
    kern = timer.result(peak, PEAK_HBM_GBS)[4]
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_chain_b{batch}_pmc_traffic_{args.gemm}.json")))
    pmc = {_pmc_key(k): v for k, v in json.load(open(cands[-1]))["kernels"].items()} if cands else {}
    roof = None
    if kern:
        name, (ms, fl, by, n) = max(kern.items(), key=lambda kv: kv[1][0])
        tf, gb = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9
        bound = "hbm" if gb / PEAK_HBM_GBS >= tf / peak else "mfma"
        t_ = pmc.get(_pmc_key(name))
        roof = {"bound": bound, "kernel": name, "achieved": round(gb if bound == "hbm" else tf, 2),
                "peak": PEAK_HBM_GBS if bound == "hbm" else round(peak, 1), "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                "frac": round(max(gb / PEAK_HBM_GBS, tf / peak), 4), "frac_mfma": round(tf / peak, 4), "frac_hbm": round(gb / PEAK_HBM_GBS, 4),
                "traffic": round(t_["bytes_per_step"] / max(t_["launches_per_step"], 1)) if t_ else None,
                "traffic_source": os.path.basename(cands[-1]) if cands else None,
                "algorithmic_flops_per_launch": round(fl / n), "algorithmic_bytes_per_launch": round(by / n),
                "launches_per_step": round(n / args.steps, 2), "avg_launch_us": round(ms / n * 1e3, 2),
                "ms_per_step": round(ms / args.steps, 3), "share_of_step": round(ms / args.steps / (dt / args.steps * 1e3), 3)}
    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline_chain(chain.last_labels.detach().cpu(), order)
    emit({"roofline": roof, "cpu_baseline": cpu,
        "metric": "audio-seconds/sec chain inference (whole job)", "value": round(world * batch * CLIP / SR * args.steps / dt, 3),
        "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": DTYPES[args.gemm],
        "data": "synthetic", "config": {"workload": "RemFX-detect chain inference (+exp=remfx_detect), inference only",
                                        "clips_per_gpu": batch, "clip_samples": CLIP, "sample_rate": SR,
                                        "removal_model_applications_per_step": napplied, "parallelism": f"dp{world}"}},
         f"chain_{args.gemm}_b{batch}_n{world}")


def measure_demucs_fwd(model, x, steps, warmup, gemm, world=1):
    """STFT + Hybrid Demucs FORWARD (DemucsModel.sample: _spec, both U-Net branches, _ispec) on the clips `x`, no loss /
    backward: seconds per pass (max over ranks), output rms, and HIP-event times of the two HBM-bound end stages."""
    from remfx_amd import stft as stft_mod
    ev = {"stft": [], "istft": []}
    orig = {"stft": stft_mod.stft, "istft": stft_mod.istft}
    state = {"on": False}

    def wrap(name):
        def f(*a, **kw):
            if not state["on"]:
                return orig[name](*a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig[name](*a, **kw)
            e.record()
            ev[name].append((s, e))
            return r
        return f
    stft_mod.stft, stft_mod.istft = wrap("stft"), wrap("istft")

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    was_training = model.training
    model.eval()
    from remfx_amd import hdemucs as _hd
    try:
        with torch.no_grad():
            for _ in range(warmup):
                model.model.sample(x)
            fence()
            t0 = time.time()
            for _ in range(steps):
                out = model.model.sample(x)
            fence()
            dt = time.time() - t0
            # the two end stages are priced like the dominant kernel: on passes with the time branch on the compute stream, so that an
            # event bracket holds the FFT launch alone (beside the time branch's kernels it reads 1.6x longer)
            two, _hd.TWO_STREAMS = _hd.TWO_STREAMS, False
            try:
                model.model.sample(x)
                fence()
                state["on"] = True
                for _ in range(3):
                    model.model.sample(x)
                fence()
                state["on"] = False
            finally:
                _hd.TWO_STREAMS = two
    finally:
        stft_mod.stft, stft_mod.istft = orig["stft"], orig["istft"]
        model.train(was_training)
    t = torch.tensor([dt], device=x.device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t) / steps
    batch = x.shape[0]
    flops = batch * (117.2e9 + 0.13e9)
    # layer-boundary bytes in the width the mode stores activations in (396 MB fp32 / 198 MB 16-bit per clip) + weights once
    nbytes32 = batch * 396e6 + 334e6
    nbytes = (batch * 198e6 + 334e6) if gemm == "bf16" else nbytes32
    mfma_peak = {"f32": PEAK_F32_TFLOPS, "bf16x3": PEAK_BF16_TFLOPS / 3.0, "bf16": PEAK_BF16_TFLOPS}[gemm]
    f_hbm, f_mfma = nbytes / dt / 1e9 / PEAK_HBM_GBS, flops / dt / 1e12 / mfma_peak
    f_hbm32 = nbytes32 / dt / 1e9 / PEAK_HBM_GBS
    # the HBM-bound end stages (SURVEY 8d "STFT-only sub-metric"): _spec reads B*T*4 and writes (B, 2, nfft/2, T/hop) fp32;
    # _ispec the reverse
    end_bytes = batch * CLIP * 4 + batch * 2 * 2048 * 256 * 4
    stages = {}
    for name in ("stft", "istft"):
        ms = sum(s.elapsed_time(e) for s, e in ev[name]) / max(len(ev[name]), 1)
        stages[name] = {"ms": round(ms, 4), "algorithmic_bytes": end_bytes,
                        "frac_hbm": round(end_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if ms > 0 else None}
    unet_ms = dt * 1e3 - stages["stft"]["ms"] - stages["istft"]["ms"]
    stages["unet"] = {"ms": round(unet_ms, 3), "frac_hbm": round((nbytes - 2 * end_bytes) / (unet_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                      "frac_mfma": round(batch * 117.2e9 / (unet_ms * 1e-3) / 1e12 / mfma_peak, 4)}
    stages["source"] = "event brackets on 3 one-stream passes after the timed ones"
    return {"seconds": dt, "flops": flops, "bytes": nbytes, "frac_hbm": f_hbm, "frac_hbm_fp32_bytes": f_hbm32, "frac_mfma": f_mfma, "mfma_peak": mfma_peak,
            "stages": stages, "output_rms": float(out.float().pow(2).mean().sqrt())}


def bench_demucs_fwd(args, rank, world, device):
    """The north_star's own sub-metric: STFT + Hybrid Demucs FORWARD on 64 x 262144-sample clips.  Algorithmic work
    (SURVEY 8d): 117.2 GFLOP and 396 MB of fp32 layer-boundary traffic per clip + 334 MB of weights per pass -> both roofline
    fractions are reported, plus the HBM fraction of each stage."""
    batch = args.batch or 64
    model = build_model("demucs", device).eval()
    x = synthetic_batch(batch, rank, device)[0]
    m = measure_demucs_fwd(model, x, args.steps, args.warmup, args.gemm, world)
    if rank != 0:
        return
    dt, f_hbm, f_mfma = m["seconds"], m["frac_hbm"], m["frac_mfma"]
    bound = "hbm" if f_hbm >= f_mfma else "mfma"
    emit({
        "metric": "audio-seconds/sec STFT + Demucs forward (whole job)", "value": round(world * batch * CLIP / SR / dt, 3),
        "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPES[args.gemm], "data": "synthetic",
        "config": {"workload": "Hybrid Demucs (cfg/model/demucs.yaml) forward only: STFT -> U-Net -> iSTFT, DemucsModel.sample",
                   "clips_per_gpu": batch, "clip_samples": CLIP, "sample_rate": SR, "parallelism": f"dp{world}",
                   "output_rms": round(m["output_rms"], 6)},
        "roofline": {"bound": bound, "achieved": round(m["bytes"] / dt / 1e9, 1) if bound == "hbm" else round(m["flops"] / dt / 1e12, 2),
                     "peak": PEAK_HBM_GBS if bound == "hbm" else m["mfma_peak"], "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                     "frac": round(max(f_hbm, f_mfma), 4), "frac_hbm": round(f_hbm, 4), "frac_mfma": round(f_mfma, 4),
                     "frac_hbm_fp32_bytes": round(m["frac_hbm_fp32_bytes"], 4),
                     "algorithmic_bytes_per_pass": m["bytes"], "algorithmic_flops_per_pass": m["flops"], "traffic": None,
                     "stages": m["stages"]}}, f"demucs_fwd_{args.gemm}_b{batch}_n{world}")


def also_block(args, model, opt, sched, sync, data, device, step):
    """Measured in the SAME process as the headline line (N = 1 only): (a) the training step in the fp32-parity arithmetic
    (bf16x3: the mode whose forward is within 1e-4 RMS of the fp32 oracle, tests/), so the driver-run record carries the
    1e-4-parity number next to the bf16 one; (b) the north_star's own sub-metric, STFT + Demucs forward, with per-stage HBM
    fractions."""
    from remfx_amd import ops
    res = {}
    if args.gemm != "bf16x3":
        prev = ops.gemm_precision()
        ops.set_gemm_precision("bf16x3")
        try:
            for i in range(2):
                step(10_000 + i)
            torch.cuda.synchronize()
            t0 = time.time()
            n = max(3, min(args.steps, 10))
            for i in range(n):
                step(10_002 + i)
            torch.cuda.synchronize()
            dtx = (time.time() - t0) / n
        finally:
            ops.set_gemm_precision(prev)
        res["train_step_bf16x3"] = {"ms_per_step": round(dtx * 1e3, 3), "audio_seconds_per_sec": round(data[0].shape[0] * CLIP / SR / dtx, 3),
                                    "dtype": DTYPES["bf16x3"], "steps": n,
                                    "parity": "forward within 1e-4 RMS of the fp32 oracle (tests/test_gpu_hdemucs.py)"}
    m = measure_demucs_fwd(model, data[0], max(3, min(args.steps, 10)), 2, args.gemm)
    res["demucs_fwd"] = {"metric": "audio-seconds/sec STFT + Demucs forward", "ms_per_pass": round(m["seconds"] * 1e3, 3),
                         "audio_seconds_per_sec": round(data[0].shape[0] * CLIP / SR / m["seconds"], 3), "dtype": DTYPES[args.gemm],
                         "frac_hbm": round(m["frac_hbm"], 4), "frac_hbm_fp32_bytes": round(m["frac_hbm_fp32_bytes"], 4),
                         "frac_mfma": round(m["frac_mfma"], 4),
                         "algorithmic_bytes_per_pass": m["bytes"], "algorithmic_flops_per_pass": m["flops"], "stages": m["stages"],
                         "target": "north_star: >= 0.5 of the HBM roofline on STFT + Demucs forward at 64 x 262144"}
    return res


def dcunet_quality(model, data, device):
    """BASELINE config 4's quality figure: SI-SDR (auraloss definition, negated) and RMS difference of the device output
    against the CPU oracle restatement on IDENTICAL input and weights -- one full 262144-sample clip, inference mode (running
    statistics), the session's arithmetic mode."""
    from oracle import ref_dcunet, ref_losses
    net = model.model.model                                  # RemFX -> DCUNetModel -> DCUNet
    ref = ref_dcunet.DCUNet(stft_kernel_size=512, fix_length_mode="pad")
    ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
    ref.eval()
    was = net.training
    net.eval()
    x = data[0][:1]
    try:
        with torch.no_grad():
            y_dev = net(x.squeeze(1)).float().cpu()
            torch.set_num_threads(min(_host_cpu()[0], 32))
            y_cpu = ref(x.squeeze(1).cpu())
    finally:
        net.train(was)
    rms = float((y_dev - y_cpu).pow(2).mean().sqrt())
    return {"si_sdr_vs_cpu_oracle_db": round(-float(ref_losses.sisdr_loss(y_dev, y_cpu)), 2),
            "rms_vs_cpu_oracle": float(f"{rms:.3e}"), "output_rms": float(f"{float(y_cpu.pow(2).mean().sqrt()):.3e}"),
            "parity_clip_samples": int(x.shape[-1]), "parity_mode": "eval (running statistics), identical weights and input"}


def _n_aux_streams(args):
    """Streams beside the compute stream and the weight-gradient stream in a Demucs step: time branch, Input_* metrics."""
    if args.workload != "demucs":
        return 0
    from remfx_amd import hdemucs as _hd, models as _md
    sink_on = ops_sink_mode() != "off"
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    return int(_hd.TWO_STREAMS and (sink_on or not multi)) + int(_md.METRIC_STREAM)


def ops_sink_mode():
    from remfx_amd import ops
    return ops.GradSink.MODE


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("RFX_WORKLOAD", "demucs"),
                    choices=["demucs", "tcn", "dcunet", "umx", "chain", "demucs_fwd"])
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (0 = the BASELINE config's batch)")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="strong: the BASELINE config's GLOBAL batch split over the ranks (SURVEY 8(e): config 3 = 64 clips -> 64 / N per GPU; "
                         "the default for the training workloads); weak: that batch on EVERY rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preheat", type=int, default=3,
                    help="minimum number of extra UNTIMED steps before the W warm-up steps (0 = no preheat at all): a freshly leased box "
                         "runs its first seconds of load ~5 %% slower (see --preheat-seconds and the loop in main)")
    ap.add_argument("--preheat-seconds", type=float, default=10.0, help="minimum wall time of the untimed preheat (see the loop in main)")
    ap.add_argument("--preheat-max", type=int, default=400, help="upper bound of the untimed preheat in steps")
    ap.add_argument("--no-priority-stream", action="store_true", help="A/B: run the step on the default stream (ops.enter_compute_stream off)")
    ap.add_argument("--sink", default="side", choices=["side", "main", "off"],
                    help="parameter-gradient sink (ops.GradSink) A/B: side stream (default) / compute stream / autograd accumulation")
    ap.add_argument("--one-stream", action="store_true",
                    help="the whole step on ONE stream (weight gradients, time branch and Input_* metrics on the compute stream): the configuration "
                         "roofline.exclusive is measured in; profiles/r05_demucs_b64_kernel_stats_bf16_onestream.csv is rocprofv3 of this command")
    ap.add_argument("--no-halo", action="store_true", help="A/B: stride-1 multi-tap convolutions on the tap-major kernels instead of gemm_halo_kernel (convplan.HALO)")
    ap.add_argument("--fused-dconv-bwd", action="store_true", help="A/B: the one-launch DConv backward (nnops.DCONV_FUSED_BWD; correct but slower, see nnops.py)")
    ap.add_argument("--no-fused-dconv", action="store_true", help="A/B: layer-by-layer DConv instead of the fused kernels (csrc/dconv.hip)")
    ap.add_argument("--no-enc-z16-time", action="store_true", help="A/B: time-branch encoder conv outputs stored as fp32 (hdemucs.ENC_Z16_TIME)")
    ap.add_argument("--no-enc-z16", action="store_true", help="A/B: encoder conv outputs stored as fp32 (hdemucs.ENC_Z16)")
    ap.add_argument("--no-also", action="store_true", help="skip the `also` block (bf16x3 step + Demucs forward sub-metric)")
    ap.add_argument("--no-exclusive", action="store_true",
                    help="skip the 3 instrumented and the 3 one-stream steps behind the timed region (the per-kernel table and the dominant "
                         "kernel's price): profiler runs then contain exactly preheat + W + K steps")
    ap.add_argument("--dump-launches", default="", help="write the per-launch plan / algorithmic work / event time list of the timed steps (JSON)")
    ap.add_argument("--union-ranks", type=int, default=0,
                    help="single process only: train on the concatenation of the synthetic batches ranks 0..N-1 would "
                         "get (N x --batch clips) -- the reference point of the data-parallel equivalence test")
    ap.add_argument("--gemm", default=os.environ.get("RFX_GEMM_PREC"), choices=["bf16", "bf16x3", "f32"],
                    help="MFMA arithmetic of the gather-GEMMs.  Default = what the BASELINE config of the workload names: bf16 "
                         "(operands rounded to bf16, fp32 accumulation = trainer.precision=bf16-mixed) for the Demucs headline "
                         "(config 3); bf16x3 (fp32 operands split hi + lo, 3 MFMAs per product: fp32 parity, forward within 1e-4 RMS "
                         "of the fp32 oracle) for the fp32 configs (TCN config 2, DCUNet config 4, Open-Unmix, chain inference); "
                         "f32: exact fp32 MFMA")
    args = ap.parse_args()
    if args.gemm is None:
        args.gemm = "bf16" if args.workload in ("demucs", "demucs_fwd") else "bf16x3"
    if args.workload == "tcn" and args.warmup < 2:
        # 32 x 262144 TCN activations fill ~190 of the 288 GB: the caching allocator settles only after its one
        # "free everything and retry" event at the start of step 2 (a 5.5 s host stall that is not part of a step)
        args.warmup = 2
    if args.workload == "tcn":
        args.preheat = 0                        # 2.5 s per step: the warm-up steps are warm-up enough

    from remfx_amd import ddp, ops
    ops.set_gemm_precision(args.gemm)
    if args.one_stream:
        from remfx_amd import hdemucs as _hd, models as _md
        args.sink = "main"
        _hd.TWO_STREAMS = False
        _md.METRIC_STREAM = False
    ops.GradSink.MODE = args.sink
    if args.no_halo:
        from remfx_amd import convplan
        convplan.HALO = False
    if args.fused_dconv_bwd:
        from remfx_amd import nnops
        nnops.DCONV_FUSED_BWD = True
    if args.no_fused_dconv:
        from remfx_amd import nnops
        nnops.DCONV_FUSED = False
    if args.no_enc_z16_time:
        from remfx_amd import hdemucs
        hdemucs.ENC_Z16_TIME = False
    if args.no_enc_z16:
        from remfx_amd import hdemucs
        hdemucs.ENC_Z16 = False
    rank, local, world = ddp.init_from_env()
    assert world == args.gpus or world == 1, (world, args.gpus)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # BASELINE.json configs: Demucs 64 clips/GPU (headline), TCN 32, DCUNet 32 over 8 GPUs = 4/GPU, UMX 4
    batch = args.batch or {"tcn": 32, "demucs": 64, "dcunet": 4, "umx": 4, "chain": 16, "demucs_fwd": 64}[args.workload]
    # Strong scaling (SURVEY 8(e) "Partitioning": batch B split evenly, B / N clips per GPU): the Demucs / TCN configs name a global
    # batch (64 / 32 clips), so `--gpus N` splits it; DCUNet / Open-Unmix / chain are quoted per GPU already (32 over 8 GPUs = 4, 128 over
    # 8 = 16) and stay weak.  `--batch` fixes the per-GPU count and implies weak.
    if args.scaling is None:
        args.scaling = "strong" if (args.workload in ("demucs", "tcn") and not args.batch) else "weak"
    if args.scaling == "strong" and world > 1:
        if batch % world:
            raise SystemExit(f"--scaling strong: the global batch {batch} does not split over {world} ranks")
        batch //= world

    if args.workload == "chain":
        return bench_chain(args, rank, world, device)
    if args.workload == "demucs_fwd":
        return bench_demucs_fwd(args, rank, world, device)
    model = build_model(args.workload, device)
    cfg = model.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    torch.cuda.synchronize()
    PHASES["import_and_model_build"] = round(time.time() - T_PROC0, 2)
    ddp.broadcast_parameters(opt.flat.data)
    sync = ddp.GradSync(opt.flat)
    sync.measure_stall = True                     # exposed_allreduce_ms below (two timing events per step; off in Trainer.fit)
    data = synthetic_batch(batch, rank, device)
    if args.union_ranks > 1:
        parts = [synthetic_batch(batch, r, device) for r in range(args.union_ranks)]
        data = tuple(torch.cat([p[i] for p in parts], 0) for i in range(4))
    if not args.no_priority_stream:
        ops.enter_compute_stream(device)             # compute stream high priority, weight-gradient side stream normal (as Trainer.fit)
    timer = KernelTimer(ops.PREC_NAMES[args.gemm])
    timer.install()

    # as Trainer.fit: Python's cyclic collector stays out of the enqueue loop (ops.StepGC: a full collection every 50 steps, between
    # two steps; RFX_STEP_GC=0 is the A/B -- 32 - 36 ms against a steady 31 ms at 8 clips)
    step_gc = ops.StepGC()
    if os.environ.get("RFX_STEP_GC", "1") != "0":
        step_gc.__enter__()

    def step(i):
        opt.zero_grad()
        loss = model.training_step(data, i)
        loss.backward()
        pre = sync.finish()
        opt.step(clip_norm=10.0, grad_prescale=pre)           # cfg/config.yaml:119
        sched.step()
        step_gc.tick()
        return loss

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Untimed preheat.  A freshly leased box runs its first ~5-9 s of sustained load ~5 % slower, as a PLATEAU (three bench processes
    # back to back on one fresh box: 151.7 / 144.1 / 143.7 ms; with 60 untimed steps in front, the first process reads 144.8 ms) --
    # too long for W = 5 warm-up steps and invisible to a "step time has stopped falling" test.  So: at least `--preheat` steps AND
    # `--preheat-seconds` of wall time, then until the last three synchronised steps are within 1.5 % of the fastest one seen (bounded
    # by `--preheat-max` steps).  `--preheat 0` disables all of it.  The count is reported as `preheat_steps`; none of it is timed.
    pre_t, n_pre, pre_t0 = [], 0, time.time()
    n_max = args.preheat_max if args.preheat > 0 else 0
    while True:
        if n_pre >= args.preheat:
            stable = len(pre_t) >= 4 and max(pre_t[-3:]) <= 1.015 * min(pre_t[1:])
            more = torch.tensor([float(n_pre < n_max and (time.time() - pre_t0 < args.preheat_seconds or not stable))], device=device)
            if world > 1:                            # every rank takes the same decision
                torch.distributed.all_reduce(more, op=torch.distributed.ReduceOp.MAX)
            if float(more) == 0.0:
                break
        torch.cuda.synchronize()
        t0 = time.time()
        step(n_pre)
        torch.cuda.synchronize()
        pre_t.append(time.time() - t0)
        n_pre += 1
    args.preheat = n_pre
    PHASES["preheat"] = round(time.time() - pre_t0, 2)
    t_w0 = time.time()
    for i in range(args.warmup):
        step(i)
    fence()
    PHASES["warmup"] = round(time.time() - t_w0, 2)
    # The K timed steps run WITHOUT the kernel timer: its two events per GEMM launch (~800 event records per step) are host work, and
    # at 8 clips per GPU the host needs as long to enqueue a step as the GPU to run it -- instrumented, the 8-clip step read 31 - 36 ms,
    # bare a steady 29.6 - 31 (scripts/probes/step_jitter.py).  The per-kernel table comes from OBS instrumented steps afterwards (every
    # rank runs them: they contain the gradient exchange).
    t0 = time.time()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    fence()
    dt = time.time() - t0
    PHASES["timed"] = round(dt, 2)
    OBS = 0 if args.no_exclusive else 3       # profiler runs (--no-exclusive) execute exactly preheat + W + K steps
    timer.enabled = True
    for i in range(OBS):
        step(args.warmup + args.steps + i)
    fence()
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
    fam_name = {"f32": "gemm_fwd_kernel<R> (gather-GEMM, v_mfma_f32_32x32x2_f32)",
                "bf16x3": "gemm_tap_kernel<R,1> + gemm_tap_stream_kernel<1,..> (tap-major gather-GEMM, 3 x v_mfma_f32_32x32x16_bf16 per K step)",
                "bf16": "gemm_tap_kernel<R,2[,IN16]> + gemm_tap_stream_kernel<2,..> + gemm_halo_kernel<R,taps,IN16> (tap-major gather-GEMM / halo-tile implicit GEMM, 1 x v_mfma_f32_32x32x16_bf16 per K step)"}[args.gemm]
    kms, klaunches, cls, ridge, kern = timer.result(peak, PEAK_HBM_GBS)
    # The timed region runs weight-gradient GEMMs on a second stream (ops.GradSink), so a forward-family launch shares the machine
    # with them and its event-timed duration is a CONCURRENT one.  For the kernel's own efficiency, time a few extra steps with
    # everything on the compute stream (not part of `value`): roofline.exclusive.
    param_abs_sum = float(opt.flat.data.double().abs().sum())     # state after exactly W + K + OBS (+ preheat) steps: the 2-rank test compares it
    excl = None
    t_x0 = time.time()
    sink = getattr(opt.flat, "sink", None)
    if rank == 0 and world == 1 and sink is not None and sink.side is not None and not args.no_exclusive:
        timer2 = KernelTimer(ops.PREC_NAMES[args.gemm])
        timer2.install()
        from remfx_amd import hdemucs as _hd, models as _md
        side, sink.side = sink.side, None
        flags = (_hd.TWO_STREAMS, _md.METRIC_STREAM)
        _hd.TWO_STREAMS = _md.METRIC_STREAM = False          # the time branch and the Input_* metrics back on the compute stream too
        try:
            step(20_000)
            torch.cuda.synchronize()
            timer2.enabled = True
            for i in range(3):
                step(20_001 + i)
            torch.cuda.synchronize()
            timer2.enabled = False
        finally:
            sink.side = side
            _hd.TWO_STREAMS, _md.METRIC_STREAM = flags
        excl = timer2.result(peak, PEAK_HBM_GBS)[4]
    PHASES["exclusive"] = round(time.time() - t_x0, 2)
    if args.dump_launches:         # per-launch plan + algorithmic work + event time, in launch order (scripts/join_launch_pmc.py)
        json.dump([dict(d, ms=s.elapsed_time(e), flops=fl, bytes=by) for d, (s, e, fl, by, _) in zip(timer.desc, timer.launches)],
                  open(args.dump_launches, "w"))
        json.dump([dict(d, ms=s.elapsed_time(e)) for s, e, d in timer.wgrad], open(args.dump_launches + ".wgrad.json", "w"))
    tot_fl = cls["mfma"][1] + cls["hbm"][1]
    tot_by = cls["mfma"][2] + cls["hbm"][2]

    def _price(ms, fl, by, n):
        """One group of launches against both roofs."""
        if not n or ms <= 0:
            return {"launches": 0}
        tf, gb = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9
        fm, fh = tf / peak, gb / PEAK_HBM_GBS
        return {"launches_per_step": round(n / max(OBS, 1), 2), "ms_per_step": round(ms / max(OBS, 1), 3), "avg_launch_us": round(ms / n * 1e3, 2),
                "algorithmic_flops_per_launch": round(fl / n), "algorithmic_bytes_per_launch": round(by / n),
                "tflops": round(tf, 2), "gbs": round(gb, 1), "frac_mfma": round(fm, 4), "frac_hbm": round(fh, 4),
                "bound": "hbm" if fh >= fm else "mfma", "frac": round(max(fh, fm), 4)}
    # HBM bytes per launch from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate counter-only runs, scripts/collect_pmc.py + measure_round.sh), joined per kernel instantiation
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{args.workload}_b{batch}_pmc_traffic_{args.gemm}.json")))   # latest round / pass last
    pmc = {}
    pmc_file = None
    if cands:
        pmc_file = os.path.basename(cands[-1])
        pmc = {_pmc_key(k): v for k, v in json.load(open(cands[-1]))["kernels"].items()}
    by_kernel = []
    for name, (ms, fl, by, n) in sorted(kern.items(), key=lambda kv: -kv[1][0]):
        row = dict(kernel=name, **_price(ms, fl, by, n))
        t = pmc.get(_pmc_key(name))
        if t:
            row["traffic"] = round(t["bytes_per_step"] / max(t["launches_per_step"], 1))
            row["traffic_launches_per_step"] = t["launches_per_step"]
            # a lower bound cannot exceed what was moved: flag it instead of hiding it (tolerance: PMC sampling + launches
            # whose tensors sit in the 256 MB Infinity Cache)
            row["traffic_ge_algorithmic"] = bool(row["traffic"] >= 0.9 * row["algorithmic_bytes_per_launch"])
        by_kernel.append(row)
    dom = by_kernel[0] if by_kernel else {"kernel": None, "bound": "hbm", "frac": 0.0}
    exclusive = None
    in_step = None
    if excl:
        # The timed step runs on up to four streams (compute, time branch, weight gradients, Input_* metrics): a launch's event-timed
        # duration there is a CONCURRENT one -- it shares the machine, and on the low-priority weight-gradient stream it also contains
        # the time the launch waited for CUs (cl_wgrad_kernel<3, 1, 64>: 879 us between its events, 591 us in rocprofv3's kernel trace
        # of the same step).  A kernel's own efficiency is what it does with the machine to itself: the dominant kernel is therefore
        # picked by, and priced on, the one-stream pass (3 extra steps, `bench.py --one-stream` is the same configuration; its
        # rocprofv3 stats are profiles/r05_demucs_b64_kernel_stats_bf16_onestream.csv); the concurrent figures stay in `in_step`.
        name_x = max(excl, key=lambda k: excl[k][0])
        ms_, fl_, by_, n_ = excl[name_x]
        row_x = dict(kernel=name_x, **{k: v for k, v in _price(ms_ * OBS / 3.0, fl_ * OBS / 3.0, by_ * OBS / 3.0, n_ * OBS / 3.0).items()})
        row_c = next((r for r in by_kernel if r["kernel"] == name_x), None)
        if row_c is not None:
            in_step = {k: row_c.get(k) for k in ("launches_per_step", "ms_per_step", "avg_launch_us", "tflops", "gbs", "frac_mfma", "frac_hbm",
                                                  "frac")}
            for k in ("traffic", "traffic_launches_per_step", "traffic_ge_algorithmic"):
                if k in row_c:
                    row_x[k] = row_c[k]
        dom = row_x
        exclusive = {"note": "the figures of this line: 3 extra steps with the whole step on one stream (no concurrent launch)",
                     "avg_launch_us": row_x["avg_launch_us"], "tflops": row_x["tflops"], "gbs": row_x["gbs"],
                     "frac_mfma": row_x["frac_mfma"], "frac_hbm": row_x["frac_hbm"], "frac": row_x["frac"]}
    fam = _price(kms, tot_fl, tot_by, klaunches)
    fam_traffic = None
    if pmc:
        rows = [v for k, v in pmc.items() if k.startswith(("gemm_fwd_kernel", "gemm_tap_kernel", "gemm_tap_stream_kernel", "gemm_halo_kernel"))]
        n = sum(v["launches_per_step"] for v in rows)
        if n:
            fam_traffic = round(sum(v["bytes_per_step"] for v in rows) / n)
    out = {
        "metric": "audio-seconds/sec fwd+bwd (whole job)", "value": round(audio_s / dt, 3),
        "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": DTYPES[args.gemm],
        "data": "synthetic", "preheat_steps": args.preheat,
        "config": {"workload": {"tcn": "TCN (cfg/model/tcn.yaml) train step, +exp=reverb model=tcn",
                                "demucs": "Hybrid Demucs (cfg/model/demucs.yaml) train step, +exp=chorus_aug model=demucs",
                                "dcunet": "DCUNet Large-DCUNet-20 (cfg/model/dcunet.yaml) train step, +exp=5-5_full model=dcunet",
                                "umx": "Open-Unmix (cfg/model/umx.yaml) train step, +exp=distortion model=umx"}[args.workload],
                   "clips_per_gpu": batch, "global_batch": batch * world, "clip_samples": CLIP, "sample_rate": SR,
                   "step": "fwd + MRSTFT+100*L1 loss + bwd + clip 10 + AdamW + per-step metrics",
                   "parallelism": f"dp{world}", "final_loss": round(float(loss.detach()), 5),
                   # data-parallel bookkeeping: collective backend ("nccl" = RCCL over xGMI), the all-reduce time the step
                   # could NOT hide behind backward (GradSync.finish wait, ms per step), and a checksum of the parameters
                   # after the timed steps (equal across replicas; tests compare it with a 1-rank run)
                   "dist_backend": torch.distributed.get_backend() if world > 1 else None, "ranks": world,
                   "rccl_ranks": world if (world > 1 and torch.distributed.get_backend() == "nccl") else 0,
                   "exposed_allreduce_ms": round(sync.exposed_ms() / max(args.steps + args.warmup, 1), 3) if hasattr(sync, "exposed_ms") else None,
                   # streams of the step: compute + weight-gradient (sink side) + time branch + Input_* metrics; the same for every N
                   "streams": 1 + int(sink is not None and sink.side is not None) + _n_aux_streams(args),
                   "param_abs_sum": param_abs_sum},
        # roofline: the SINGLE dominant kernel instantiation of the step's dominant family (most event-timed ms): achieved =
        # its algorithmic bytes (or flops) per launch / its average launch duration; `bound` = the roof it sits closer to.
        # `by_kernel` lists every instantiation of the family the same way, `family` the whole family, `by_bound` the
        # family split at the ridge (peak FLOP/s / peak B/s) by each launch's own arithmetic intensity.
        "roofline": {"bound": dom["bound"], "kernel": dom["kernel"],
                     "achieved": (dom.get("gbs") if dom["bound"] == "hbm" else dom.get("tflops")),
                     "peak": PEAK_HBM_GBS if dom["bound"] == "hbm" else round(peak, 1),
                     "unit": "GB/s" if dom["bound"] == "hbm" else "TFLOP/s", "frac": dom["frac"],
                     "frac_mfma": dom.get("frac_mfma"), "frac_hbm": dom.get("frac_hbm"),
                     "traffic": dom.get("traffic"), "traffic_source": pmc_file,
                     "algorithmic_bytes_per_launch": dom.get("algorithmic_bytes_per_launch"),
                     "algorithmic_flops_per_launch": dom.get("algorithmic_flops_per_launch"),
                     "launches_per_step": dom.get("launches_per_step"), "avg_launch_us": dom.get("avg_launch_us"),
                     "ms_per_step": dom.get("ms_per_step"), "share_of_step": round(dom.get("ms_per_step", 0.0) / (dt / args.steps * 1e3), 3),
                     "concurrent_streams": (1 + int(sink is not None and sink.side is not None) + _n_aux_streams(args)), "exclusive": exclusive,
                     "frac_source": "one-stream pass" if exclusive else "timed region", "in_step": in_step,
                     "ridge_flop_per_byte": round(ridge, 1), "by_kernel": by_kernel,
                     "family": dict(fam, kernel=fam_name, traffic=fam_traffic, share_of_step=round((kms / max(OBS, 1)) / (dt / args.steps * 1e3), 3)),
                     "by_bound": {"mfma": _price(*cls["mfma"]), "hbm": _price(*cls["hbm"])}},
    }
    if args.workload == "demucs" and world == 1 and not args.no_also:
        t_a0 = time.time()
        out["also"] = also_block(args, model, opt, sched, sync, data, device, step)
        PHASES["also"] = round(time.time() - t_a0, 2)
    if args.workload == "dcunet" and world == 1 and not args.no_also:
        t_a0 = time.time()
        out["also"] = dcunet_quality(model, data, device)
        PHASES["also"] = round(time.time() - t_a0, 2)
    if args.workload == "tcn" and world == 1 and not args.no_also and args.gemm != "f32":
        # BASELINE config 2 is quoted in fp32: the exact-fp32 MFMA step (v_mfma_f32_32x32x2_f32) next to the bf16x3 headline
        t_a0 = time.time()
        prev = ops.gemm_precision()
        ops.set_gemm_precision("f32")
        try:
            step(30_000)
            torch.cuda.synchronize()
            t1 = time.time()
            step(30_001)
            torch.cuda.synchronize()
            dt32 = time.time() - t1
        finally:
            ops.set_gemm_precision(prev)
        out["also"] = {"train_step_f32": {"ms_per_step": round(dt32 * 1e3, 1), "audio_seconds_per_sec": round(batch * CLIP / SR / dt32, 3),
                                          "dtype": DTYPES["f32"], "steps": 1,
                                          "frac_mfma_f32_peak": round(3 * ALG["tcn"]["flops_fwd"] * batch / dt32 / 1e12 / PEAK_F32_TFLOPS, 4)}}
        PHASES["also"] = round(time.time() - t_a0, 2)
    if not args.no_cpu_baseline:
        t_c0 = time.time()
        out["cpu_baseline"] = cpu_baseline(args.workload)
        PHASES["cpu_baseline"] = round(time.time() - t_c0, 2)
    # where the wall time of this process went (seconds; `timed` is the K steps `value` is computed from, everything else is untimed)
    PHASES["total"] = round(time.time() - T_PROC0, 2)
    out["phases_s"] = PHASES
    tot_traffic = None
    if cands:
        tot_traffic = json.load(open(cands[-1])).get("total_bytes_per_step")
    out["step_roofline"] = dict(step_roofline(args.workload, args.gemm, batch, dt / args.steps, peak, tot_traffic), traffic_source=pmc_file)
    emit(out, f"{args.workload}_{args.gemm}_b{batch}_n{world}")


if __name__ == "__main__":
    main()
