"""Oracle (test infrastructure): micro-TCN.

Restates /root/reference/remfx/tcn.py:11-59 (TCNBlock) and tcn.py:62-138 (TCN)
with functional torch CPU ops.  PINNED by tests/golden/tcn_small.npz (outputs
of the imported reference module on the same weights).

Parameter names match the reference state_dict:
  process_blocks.{n}.conv1.{weight,bias}, .res.weight, .relu.weight,
  output.{weight,bias}
"""
import torch
import torch.nn.functional as F

from .ref_utils import causal_crop, center_crop


def tcn_dilations(nblocks, stack_size, dilation_growth):
    # tcn.py:108
    return [dilation_growth ** (n % stack_size) for n in range(nblocks)]


def tcn_receptive_field(nblocks, kernel_size, stack_size, dilation_growth):
    # tcn.py:132-138
    rf = kernel_size
    for n in range(1, nblocks):
        rf += (kernel_size - 1) * dilation_growth ** (n % stack_size)
    return rf


def tcn_forward(x, sd, nblocks, stack_size=10, dilation_growth=2, causal=False):
    """x: (B, ninputs, T); sd: dict of tensors with reference key names."""
    crop = causal_crop if causal else center_crop       # tcn.py:94-97
    dil = tcn_dilations(nblocks, stack_size, dilation_growth)
    for n in range(nblocks):
        p = f"process_blocks.{n}."
        y = F.conv1d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"],
                     dilation=dil[n])                   # tcn.py:50
        y = F.prelu(y, sd[p + "relu.weight"])           # tcn.py:51
        r = F.conv1d(x, sd[p + "res.weight"])           # tcn.py:54
        x = y + crop(r, y.shape[-1])                    # tcn.py:57
    return torch.tanh(F.conv1d(x, sd["output.weight"], sd["output.bias"]))  # tcn.py:129


def tcn_init_state_dict(ninputs=1, noutputs=1, nblocks=4, channel_width=32,
                        kernel_size=13, seed=0):
    """Deterministic weights with torch-default-like scale (kaiming-uniform
    bound 1/sqrt(fan_in)); NOT the reference's RNG stream -- parity tests load
    explicit weights on both sides."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    cin = ninputs
    for n in range(nblocks):
        p = f"process_blocks.{n}."
        b = 1.0 / (cin * kernel_size) ** 0.5
        sd[p + "conv1.weight"] = (torch.rand(channel_width, cin, kernel_size, generator=g) * 2 - 1) * b
        sd[p + "conv1.bias"] = (torch.rand(channel_width, generator=g) * 2 - 1) * b
        b = 1.0 / cin ** 0.5
        sd[p + "res.weight"] = (torch.rand(channel_width, cin, 1, generator=g) * 2 - 1) * b
        sd[p + "relu.weight"] = torch.full((channel_width,), 0.25)
        cin = channel_width
    b = 1.0 / cin ** 0.5
    sd["output.weight"] = (torch.rand(noutputs, cin, 1, generator=g) * 2 - 1) * b
    sd["output.bias"] = (torch.rand(noutputs, generator=g) * 2 - 1) * b
    return sd
