"""Known-byte-count launches for calibrating FETCH_SIZE / WRITE_SIZE on gfx950 (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import ops, losses, nnops
dev = torch.device("cuda:0")
n = 1 << 28                                   # 2^28 floats = 1 GiB
x = torch.randn(n, device=dev); y = torch.randn(n, device=dev)
for _ in range(2):
    ops.activation(x, "relu")                 # act_fwd_kernel: dwordx4 loads, reads 1 GiB, writes 1 GiB
    losses.L1Loss()(x, y)                     # l1_sum_kernel: dword loads, reads 2 GiB, writes ~0
    nnops.group_norm(x.view(64, 64, 65536), 1, torch.ones(64, device=dev), torch.zeros(64, device=dev))
    # gn_stats: dword loads 1 GiB; gn_apply: dword loads 1 GiB (+stats), writes 1 GiB
torch.cuda.synchronize()
