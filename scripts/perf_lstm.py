"""Time the LSTM recurrence kernels alone (HDemucs DConv shapes at the BASELINE batch)."""
import sys
import torch
sys.path.insert(0, ".")
from remfx_amd import _lib, lstm
from remfx_amd.ops import _ptr, _stream

L = _lib.lib()
import os
PREC = int(os.environ.get("LSTM_PREC", "2"))     # 2 = bf16 mode (single bf16 fragments), 1 = bf16x3
shapes = [(192, 200, 192), (384, 128, 64), (192, 200, 24), (384, 128, 8)]     # HDemucs B = 64 and B = 8
for H, T, Bn in shapes:
    P = T * Bn
    w = torch.randn(4 * H, H, device="cuda") * 0.05
    pack = lstm._pack_whh(w, w)
    xp = torch.randn(2, 4 * H, P, device="cuda")
    out = torch.empty(2 * H, P, device="cuda")
    gates = torch.empty(2, 4 * H, P, device="cuda")
    cst = torch.empty(2, H, P, device="cuda")
    dG = torch.empty(2, 4 * H, P, device="cuda")
    g = torch.randn(2 * H, P, device="cuda")
    ws = lstm._workspace(g.device, H)
    def fwd():
        L.rfx_lstm_fwd(_ptr(xp), _ptr(pack), T, Bn, H, _ptr(out), _ptr(gates), _ptr(cst), _ptr(ws), PREC, _stream())
    def inf():
        L.rfx_lstm_fwd(_ptr(xp), _ptr(pack), T, Bn, H, _ptr(out), None, None, _ptr(ws), PREC, _stream())
    def bwd():
        L.rfx_lstm_bwd(_ptr(g), _ptr(pack), _ptr(gates), _ptr(cst), T, Bn, H, _ptr(dG), _ptr(ws), PREC, _stream())
    for name, fn in (("fwd", fwd), ("inf", inf), ("bwd", bwd)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"H={H} T={T} Bn={Bn} {name}: {ms:.3f} ms  {1e3 * ms / T:.2f} us/step", flush=True)
print("error flag:", lstm.error_flag())
