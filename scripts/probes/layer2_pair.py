"""Dev probe: the two halves of encoder layer 2 of the headline HDemucs that the hazard hunt points at, side by side
on two streams: the frequency branch's channel-major DConv (C = 192) on the main stream, the time branch's stride-4 node (EncMidFn,
folded view) on a side stream -- and the reverse pairing -- against serial references."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import ops, clchain
from remfx_amd.hdemucs import HDemucs

DEV = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
torch.manual_seed(11)
net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48).to(DEV).eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith(".scale"):
            p.fill_(0.3)
g = torch.Generator().manual_seed(3)
side = torch.cuda.Stream(priority=-1)
fe, te = net.freq_encoder, net.time_encoder
tot = 0
for B in (8, 1, 8, 1):
    xf = (torch.randn(B * 32, 192, 256, generator=g) * 0.5).to(DEV)          # frequency layer 2: (B Fr, C, T) samples
    xt = (torch.randn(B, 192, 4096, generator=g) * 0.5).to(DEV)              # time layer 2: (B, C, L)
    fns = {
        "freq dconv": lambda: fe[2].dconv(xf),
        "time dconv": lambda: te[2].dconv(xt),
        "time enc_mid": lambda: clchain.enc_mid(xt, te[2].rewrite, te[3].conv, None, B, y_cl=False, fold=True),
        "freq enc_mid": lambda: clchain.enc_mid(xf, fe[2].rewrite, fe[3].conv, None, B, y_cl=False),
    }
    with torch.no_grad():
        ref = {k: [t.clone() for t in (f() if isinstance(f(), tuple) else (f(),))] for k, f in fns.items()}
    torch.cuda.synchronize()
    for a, b in (("freq dconv", "time enc_mid"), ("freq enc_mid", "time dconv"), ("freq enc_mid", "time enc_mid"), ("freq dconv", "time dconv")):
        bad = 0
        for it in range(60):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.no_grad():
                with torch.cuda.stream(side):
                    rb = fns[b]()
                ra = fns[a]()
            main.wait_stream(side)
            torch.cuda.synchronize()
            for r, k in ((ra, a), (rb, b)):
                outs = r if isinstance(r, tuple) else (r,)
                if any(float((o.float() - q.float()).abs().max()) > 1e-4 for o, q in zip(outs, ref[k])):
                    bad += 1
        tot += bad
        print(f"B={B}: main {a:13s} | side {b:13s}: {bad} / 60 bad", flush=True)
print("total bad", tot)
