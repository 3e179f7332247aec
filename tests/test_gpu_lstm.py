"""BiLSTM recurrence kernels (csrc/lstm.hip) vs torch.nn.LSTM evaluated on the CPU in fp64.

Reference call sites: torchaudio HDemucs `_BLSTM` (remfx/models.py:319) and Open-Unmix (models.py:297-298)."""
import pytest
import torch

from tests.conftest import check, mode, tol
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _run(H, Cin, T, Bn, layers, seed=0):
    from remfx_amd import lstm
    torch.manual_seed(seed)
    ref = nn.LSTM(Cin, H, num_layers=layers, bidirectional=True).double()
    x = torch.randn(T, Bn, Cin, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(T, Bn, 2 * H, dtype=torch.float64)
    y = ref(x)[0]
    (y * gy).sum().backward()

    dev = nn.LSTM(Cin, H, num_layers=layers, bidirectional=True).cuda()
    dev.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    xc = x.detach().float().permute(2, 0, 1).reshape(1, Cin, T * Bn).contiguous().cuda().requires_grad_(True)
    out = lstm.blstm(dev, xc, T, Bn)
    gyc = gy.float().permute(2, 0, 1).reshape(1, 2 * H, T * Bn).contiguous().cuda()
    (out * gyc).sum().backward()

    def rel(a, b):
        return ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()

    yc = y.detach().permute(2, 0, 1).reshape(1, 2 * H, T * Bn)
    check(rel(out.detach(), yc), 1e-4)
    check(rel(xc.grad, x.grad.permute(2, 0, 1).reshape(1, Cin, T * Bn)), 2e-4)
    for (n, p), (_, q) in zip(dev.named_parameters(), ref.named_parameters()):
        check(rel(p.grad, q.grad), 3e-4, what=n)
    assert not lstm.error_flag()            # no bounded spin of the cluster exchange timed out


@pytest.mark.parametrize("H,Cin,T,Bn,layers", [
    (32, 32, 7, 5, 1),          # one wave, ragged sequence tile
    (64, 48, 20, 33, 2),        # two batch tiles, two layers
    (192, 192, 50, 40, 2),      # HDemucs DConv BLSTM width (layer 4); bf16 mode: forward in the single-workgroup form (csrc/lstm.hip)
    (192, 192, 30, 12, 1),      # ... and at <= 16 sequences the backward sweep too (ragged 16-sequence tile)
    (384, 384, 12, 20, 1),      # HDemucs layer 5 width: 16-sequence backward tiles
    (256, 512, 9, 4, 3),        # Open-Unmix (hidden 512 -> 256 per direction, 3 layers)
    (64, 16, 3, 8200, 1),       # more sequence tiles than one co-resident launch holds (chunked launches)
])
def test_blstm_matches_torch(H, Cin, T, Bn, layers):
    _run(H, Cin, T, Bn, layers)


def test_blstm_inference_no_saved_state():
    from remfx_amd import lstm
    torch.manual_seed(1)
    ref = nn.LSTM(64, 64, num_layers=2, bidirectional=True)
    x = torch.randn(11, 3, 64)
    with torch.no_grad():
        y = ref(x)[0]
        dev = nn.LSTM(64, 64, num_layers=2, bidirectional=True).cuda()
        dev.load_state_dict(ref.state_dict())
        out = lstm.blstm(dev, x.permute(2, 0, 1).reshape(1, 64, 33).contiguous().cuda(), 11, 3)
    check(float((out.cpu() - y.permute(2, 0, 1).reshape(1, 128, 33)).abs().max()), 2e-5)


def test_blstm_gradients_through_the_sink():
    """With a GradSink armed (parameters in an optim.FlatParams buffer) the eight parameter gradients of a layer land in the flat
    gradient buffer -- dW_hh by the weight-gradient unpack, the six others by ONE rfx_lstm_grad_scatter launch (both biases of a
    direction receive the same gradient) -- with the values autograd gets without a sink; a second pass accumulates (2x)."""
    from remfx_amd import lstm
    from remfx_amd.optim import FlatParams
    torch.manual_seed(5)
    H, Cin, T, Bn = 64, 48, 9, 7
    mod = nn.LSTM(Cin, H, num_layers=2, bidirectional=True).cuda()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, Cin, T * Bn, generator=g).cuda()
    gy = torch.randn(1, 2 * H, T * Bn, generator=g).cuda()
    xr = x.clone().requires_grad_(True)
    (lstm.blstm(mod, xr, T, Bn) * gy).sum().backward()
    ref = {n: p.grad.detach().clone() for n, p in mod.named_parameters()}
    for p in mod.parameters():
        p.grad = None
    flat = FlatParams(list(mod.parameters()))
    assert flat.sink is not None
    flat.zero_grad()
    for rep in range(2):
        xs = x.clone().requires_grad_(True)
        (lstm.blstm(mod, xs, T, Bn) * gy).sum().backward()
    nsunk = sum(1 for w in flat.sink.writes if w)
    flat.join()
    torch.cuda.synchronize()
    assert nsunk == len(flat.params), (nsunk, len(flat.params))
    for n, p in mod.named_parameters():
        scale = float(ref[n].abs().max())
        assert float((p.grad - 2 * ref[n]).abs().max()) <= 2e-6 * max(scale, 1.0), n
    assert not lstm.error_flag()
