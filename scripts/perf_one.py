"""One kernel family at one shape, few launches (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import ops, stft
from remfx_amd.tcn import tcn_block_forward
dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "tcn"
if what == "tcn":
    B, C, L, d = 4, 256, 262144, 16
    x = torch.randn(B, C, L, device=dev); w1 = torch.randn(C, C, 7, device=dev) * 0.02
    b1 = torch.randn(C, device=dev); sl = torch.full((C,), 0.25, device=dev); wr = torch.randn(C, C, 1, device=dev) * 0.05
    for _ in range(3):
        tcn_block_forward(x, w1, b1, sl, wr, d, False)
elif what == "wgrad":
    B, C, L, d = 4, 256, 262144, 16
    x = torch.randn(B, C, 1, L, device=dev); g = torch.randn(B, C, 1, L - 6 * d, device=dev)
    for _ in range(3):
        ops.conv2d_wgrad(x, g, (C, C, 1, 7), (1, 1), (0, 0), (1, d), True)
elif what == "stft":
    x = torch.randn(64, 262144, device=dev)
    for _ in range(3):
        stft.stft(x, 4096, 1024, mode="cac", normalized=True, bins=2048, frame0=2, frames_out=256, extra_pad=(1536, 1536))
        stft.stft(x, 1024, 120, 600, mode="complex")
        stft.stft(x, 2048, 240, 1200, mode="complex")
        stft.stft(x, 512, 50, 240, mode="complex")
torch.cuda.synchronize()
