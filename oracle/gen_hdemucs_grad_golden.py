"""Full-config Hybrid Demucs GRADIENT fixture: runs the CPU oracle (oracle/ref_hdemucs.py, 83.6 M
parameters, cfg/model/demucs.yaml geometry) forward + backward ONCE on one seeded 262144-sample clip and stores, for a
spread of parameters, the gradient's norm and a strided slice, plus output slices -> tests/golden/hdemucs_full_grad.npz.
Weights come from the seeded initialiser (tests/test_gpu_hdemucs.py::_pair uses the same recipe), so the fixture holds
inputs' seeds + expected outputs only.      python oracle/gen_hdemucs_grad_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import ref_hdemucs  # noqa: E402

PARAMS = ["freq_encoder.0.conv.weight", "freq_encoder.0.dconv.layers.0.3.weight", "freq_encoder.1.dconv.layers.1.0.weight",
          "freq_encoder.2.rewrite.weight", "freq_encoder.4.dconv.layers.0.3.lstm.weight_hh_l0",
          "freq_encoder.5.dconv.layers.0.4.content.weight", "time_encoder.1.conv.weight", "time_encoder.3.dconv.layers.1.6.scale",
          "freq_decoder.0.conv_tr.weight", "freq_decoder.2.rewrite.bias", "freq_decoder.5.conv_tr.weight",
          "time_decoder.3.conv_tr.weight", "freq_emb.embedding.weight"]


def build(seed=3):
    torch.manual_seed(seed)
    ref = ref_hdemucs.HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith(".scale"):
                p.fill_(0.3)
    return ref


def inputs():
    g = torch.Generator().manual_seed(12)
    x = torch.randn(1, 1, 262144, generator=g) * 0.1
    gy = torch.randn(1, 1, 1, 262144, generator=g)
    return x, gy


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = build()
    names = dict(ref.named_parameters())
    missing = [n for n in PARAMS if n not in names]
    assert not missing, (missing, [n for n in names if "lstm" in n or "content" in n or "emb" in n][:12])
    x, gy = inputs()
    y = ref(x)
    y.backward(gy)
    out = {"y_slice": y.detach().reshape(-1)[::4099].numpy(), "y_norm": np.float64(y.detach().double().norm()),
           "names": np.array(PARAMS)}
    tot = 0.0
    for n, p in ref.named_parameters():
        if p.grad is not None:               # parameters the single-source forward never touches have no gradient
            tot += float(p.grad.double().pow(2).sum())
    out["grad_global_norm"] = np.float64(tot ** 0.5)
    for i, n in enumerate(PARAMS):
        gr = names[n].grad.detach().reshape(-1)
        step = max(1, gr.numel() // 512)
        out[f"g{i}_slice"] = gr[::step][:512].numpy()
        out[f"g{i}_norm"] = np.float64(gr.double().norm())
    path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "hdemucs_full_grad.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "global grad norm", out["grad_global_norm"])


if __name__ == "__main__":
    main()
