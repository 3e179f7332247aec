"""Dev tool: time the HDemucs DConv BLSTM (2 layers, bidirectional) forward + backward alone, bf16 mode.
usage: python scripts/perf_lstm.py [H] [T] [Bn]      (RFX_LSTM_LOCAL=0: cluster form only)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from remfx_amd import lstm, ops

H = int(sys.argv[1]) if len(sys.argv) > 1 else 192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
Bn = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ops.set_gemm_precision("bf16")
mod = nn.LSTM(H, H, num_layers=2, bidirectional=True).cuda()
x = torch.randn(1, H, T * Bn, device="cuda", requires_grad=True)
g = torch.randn(1, 2 * H, T * Bn, device="cuda")


def step():
    y = lstm.blstm(mod, x, T, Bn)
    y.backward(g)
    return y


for _ in range(3):
    y = step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    y = step()
e1.record()
torch.cuda.synchronize()
print(f"BLSTM H={H} T={T} Bn={Bn} local={os.environ.get('RFX_LSTM_LOCAL', '1')}: {e0.elapsed_time(e1) / 10:.3f} ms fwd + bwd "
      f"(2 layers), |y| {float(y.abs().sum()):.6e}, |dx| {float(x.grad.abs().sum()):.6e}, err {lstm.error_flag()}")
