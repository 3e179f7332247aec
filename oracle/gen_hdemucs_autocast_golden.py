"""Full-config Hybrid Demucs gradient fixture for the bf16 arithmetic mode: the CPU oracle
(oracle/ref_hdemucs.py, cfg/model/demucs.yaml geometry, 83.6 M parameters) run twice on one seeded 262144-sample clip -- in fp32
and under torch.autocast("cpu", bfloat16), what Lightning's bf16-mixed precision does to the reference -- forward + backward.
For EVERY parameter tensor a strided slice (<= 256 values) of both gradients is stored, plus output slices:
tests/golden/hdemucs_full_grad_autocast.npz.  The GPU test bounds the HIP bf16 mode's error against fp32 by the autocast
oracle's own error.  Same seeds / initialiser as oracle/gen_hdemucs_grad_golden.py.
    python oracle/gen_hdemucs_autocast_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle.gen_hdemucs_grad_golden import build, inputs  # noqa: E402

NSL = 256


def slices(ref):
    out = {}
    for n, p in ref.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().reshape(-1)
        step = max(1, g.numel() // NSL)
        out[n] = g[::step][:NSL].float().numpy().copy()
    return out


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = build()
    x, gy = inputs()
    y = ref(x)
    y.backward(gy)
    s32 = slices(ref)
    y32 = y.detach().reshape(-1)[::4099].numpy().copy()
    ref.zero_grad(set_to_none=True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ya = ref(x)
    ya.float().backward(gy)
    sa = slices(ref)
    names = sorted(s32)
    out = {"names": np.array(names), "y32": y32, "yauto": ya.detach().float().reshape(-1)[::4099].numpy()}
    num = den = 0.0
    for i, n in enumerate(names):
        out[f"f{i}"] = s32[n]
        out[f"a{i}"] = sa[n]
        num += float(((sa[n].astype(np.float64) - s32[n]) ** 2).sum())
        den += float((s32[n].astype(np.float64) ** 2).sum())
    out["auto_rel"] = np.float64((num / den) ** 0.5)
    path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "hdemucs_full_grad_autocast.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "tensors", len(names), "autocast relative gradient error (sliced)", out["auto_rel"],
          "output rel err", float(np.sqrt(((out["yauto"] - y32) ** 2).sum() / (y32 ** 2).sum())))


if __name__ == "__main__":
    main()
