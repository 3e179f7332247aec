"""Full-length (262144-sample) fixtures for BASELINE configs 4 (DCUNet, reference remfx/models.py:356-367) and 1 (Open-Unmix,
remfx/models.py:294-304: the 513-step 3-layer BiLSTM): the CPU oracles (oracle/ref_dcunet.py, oracle/ref_umx.py) run ONCE on one
seeded clip -- eval forward, train-mode forward + backward -- and the expected output slices, gradient norms and strided gradient
slices go to tests/golden/dcunet_full.npz / umx_full.npz.  Weights come from the seeded recipes of tests/test_gpu_dcunet.py::_pair
and tests/test_gpu_umx.py::_pair, so the fixtures hold seeds' consequences (inputs + expected outputs) only.  The DCUNet train-mode
gradients are taken from an fp64 run of the oracle (its batch-statistic 2x2 whitening is ill-conditioned in fp32: the fp32 run's
own distance from fp64 is stored next to them as the premise of the test's bound).
    python oracle/gen_full_length_golden.py [dcunet|umx]"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
T = 262144


def _slice(t, n=512):
    t = t.detach().reshape(-1)
    step = max(1, t.numel() // n)
    return t[::step][:n]


def dcunet_pair_oracle(train):
    """tests/test_gpu_dcunet.py::_pair, oracle half."""
    from oracle import ref_dcunet
    torch.manual_seed(0)
    ref = ref_dcunet.DCUNet(stft_kernel_size=512, fix_length_mode="pad")
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for n, b in ref.named_buffers():
            if n.endswith(("RMr", "RMi")):
                b.copy_(torch.randn(b.shape, generator=g) * 0.05)
            elif n.endswith(("RVrr", "RVii")):
                b.copy_(torch.rand(b.shape, generator=g) * 0.5 + 0.75)
            elif n.endswith("RVri"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.05)
        for n, p in ref.named_parameters():
            if n.endswith((".Br", ".Bi")):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    ref.train(train)
    return ref


def full_inputs(seed, shape_y=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, T, generator=g) * 0.3
    return x, g


def gen_dcunet():
    out = {}
    ref = dcunet_pair_oracle(False)
    x, g = full_inputs(21)
    with torch.no_grad():
        y = ref(x)
    out["eval_y_slice"] = _slice(y, 2048).numpy()
    out["eval_y_norm"] = np.float64(y.double().norm())
    out["eval_y_absmax"] = np.float64(y.abs().max())
    print("dcunet eval forward done", tuple(y.shape), flush=True)
    ref = dcunet_pair_oracle(True)
    ref64 = copy.deepcopy(ref).double()
    gy = torch.randn(y.shape, generator=g)
    y32 = ref(x)
    y32.backward(gy)
    print("dcunet train fp32 done", flush=True)
    y64 = ref64(x.double())
    y64.backward(gy.double())
    print("dcunet train fp64 done", flush=True)
    out["train_y_slice"] = _slice(y64, 2048).float().numpy()
    out["train_y_absmax"] = np.float64(y64.detach().abs().max())
    g32 = {n: p.grad for n, p in ref.named_parameters()}
    g64 = {n: p.grad for n, p in ref64.named_parameters()}
    num = sum(float((g32[n].double() - g64[n]).pow(2).sum()) for n in g64)
    den = sum(float(g64[n].pow(2).sum()) for n in g64)
    out["cpu_fp32_vs_fp64_global_rel"] = np.float64((num / den) ** 0.5)
    out["grad_global_norm"] = np.float64(den ** 0.5)
    names = [n for n, _ in ref.named_parameters()]
    pick = [names[i] for i in np.linspace(0, len(names) - 1, 10).round().astype(int)]
    out["names"] = np.array(pick)
    for i, n in enumerate(pick):
        out[f"g{i}_slice"] = _slice(g64[n]).float().numpy()
        out[f"g{i}_norm"] = np.float64(g64[n].norm())
    # running statistics after the one train-mode step (momentum lerp), first / last normalisation layer
    rb = dict(ref64.named_buffers())
    rm = [n for n in rb if n.endswith("RMr")]
    out["rm_names"] = np.array([rm[0], rm[-1]])
    out["rm0"], out["rm1"] = rb[rm[0]].float().numpy(), rb[rm[-1]].float().numpy()
    path = os.path.join(GOLDEN, "dcunet_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "cpu fp32 vs fp64", out["cpu_fp32_vs_fp64_global_rel"], flush=True)


def umx_pair_oracle():
    """tests/test_gpu_umx.py::_pair, oracle half."""
    from oracle import ref_umx
    torch.manual_seed(0)
    ref = ref_umx.OpenUnmix(nb_bins=1025, nb_channels=1)
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for n, p in ref.named_parameters():
            if n in ("input_mean", "input_scale", "output_scale", "output_mean"):
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
        for n, b in ref.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.05)
            elif n.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) * 0.5 + 0.75)
    return ref


def gen_umx():
    from oracle import ref_umx
    out = {}
    g = torch.Generator().manual_seed(22)
    x = torch.randn(2, 1, T, generator=g) * 0.3           # 2 clips: train-mode BatchNorm1d statistics over (batch, frames)
    ref = umx_pair_oracle()
    ref.eval()
    with torch.no_grad():
        y = ref_umx.separator(ref, x)
    out["eval_y_slice"] = _slice(y, 2048).numpy()
    out["eval_y_norm"] = np.float64(y.double().norm())
    out["eval_y_absmax"] = np.float64(y.abs().max())
    print("umx eval done", tuple(y.shape), flush=True)
    ref = umx_pair_oracle()
    ref.train()
    ref.lstm.dropout = 0.0
    gy = torch.randn(y.shape, generator=g)
    yt = ref_umx.separator(ref, x)
    yt.backward(gy)
    out["train_y_slice"] = _slice(yt, 2048).numpy()
    out["train_y_absmax"] = np.float64(yt.detach().abs().max())
    names = [n for n, p in ref.named_parameters() if p.grad is not None]
    tot = sum(float(dict(ref.named_parameters())[n].grad.double().pow(2).sum()) for n in names)
    out["grad_global_norm"] = np.float64(tot ** 0.5)
    params = dict(ref.named_parameters())
    out["names"] = np.array(names)
    for i, n in enumerate(names):
        out[f"g{i}_slice"] = _slice(params[n].grad).numpy()
        out[f"g{i}_norm"] = np.float64(params[n].grad.double().norm())
    path = os.path.join(GOLDEN, "umx_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "frames", (T // 512) + 1, "global grad norm", out["grad_global_norm"], flush=True)


if __name__ == "__main__":
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    which = sys.argv[1:] or ["dcunet", "umx"]
    if "umx" in which:
        gen_umx()
    if "dcunet" in which:
        gen_dcunet()
