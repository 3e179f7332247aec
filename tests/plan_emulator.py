"""Test helper: CPU interpreter of the gather-GEMM descriptor semantics documented
in include/remfx_hip.h.  Validates the host-side planner (remfx_amd/convplan.py)
against torch's own conv ops without a GPU.  Never used by the product path."""
import numpy as np
import torch


def _gather(plan, x_flat, n):
    """-> B matrix [K, P] of gathered inputs (zeros where out of bounds)."""
    kt = torch.from_numpy(plan.ktab[:plan.K].astype(np.int64))
    P = plan.OA * plan.OB
    j = torch.arange(P)
    a, b = j // plan.OB, j % plan.OB
    ia0, ib0 = a * plan.SA, b * plan.SB
    pos = ia0 * plan.in_as + ib0 * plan.in_bs + n * plan.in_ns
    idx = kt[:, 0:1] + pos[None, :]
    ok = ((ia0[None] + kt[:, 1:2]) >= 0) & ((ia0[None] + kt[:, 1:2]) < plan.IA) & \
         ((ib0[None] + kt[:, 2:3]) >= 0) & ((ib0[None] + kt[:, 2:3]) < plan.IB)
    ones = (kt[:, 3:4] & 1).bool()
    vals = x_flat[idx.clamp(0, x_flat.numel() - 1)] * ok
    return torch.where(ones.expand_as(vals), torch.ones_like(vals), vals)


def _out_index(plan, n):
    P = plan.OA * plan.OB
    j = torch.arange(P)
    a, b = j // plan.OB, j % plan.OB
    return n * plan.out_ns + (a * plan.out_sa + plan.out_a0) * plan.out_as + \
        (b * plan.out_sb + plan.out_b0) * plan.out_bs


def emulate_fwd(plan, w_flat, x_flat, out_flat, bias=None):
    nrows = plan.extra["n_weight_rows"]
    m = torch.arange(plan.M)
    A = w_flat[(m[None, :] * plan.w_ms + torch.from_numpy(plan.woff.astype(np.int64))[:, None])]  # [K, M]
    for n in range(plan.N):
        B = _gather(plan, x_flat, n)[:nrows]
        o = A.t().double() @ B.double()
        if bias is not None:
            o = o + bias[m >> getattr(plan, "mg_log", 0)][:, None].double()
        if getattr(plan, "mg_log", 0):
            # phase-merged store (rfx_gemm_desc.mg_*): row m = channel*G + phase, position i -> axis index i*G + phase + off
            G = 1 << plan.mg_log
            P = plan.OA * plan.OB
            j = torch.arange(P)
            a, b = j // plan.OB, j % plan.OB
            pos = b if plan.mg_axis else a
            idx = pos[None, :] * G + (m[:, None] & (G - 1)) + plan.mg_off                      # [M, P]
            ok = (idx >= 0) & (idx < plan.mg_len)
            other = ((a * plan.out_sa + plan.out_a0) * plan.out_as) if plan.mg_axis else ((b * plan.out_sb + plan.out_b0) * plan.out_bs)
            st = plan.out_bs if plan.mg_axis else plan.out_as
            addr = n * plan.out_ns + (m[:, None] >> plan.mg_log) * plan.out_cs + other[None, :] + idx * st
            out_flat[addr[ok]] = o.float()[ok]
            continue
        oi = _out_index(plan, n)
        out_flat[(m[:, None] * plan.out_cs + oi[None, :])] = o.float()
    return out_flat


def emulate_wgrad(plan, x_flat, g_flat):
    """-> dapack [M, K] (the layout rfx_gemm_wgrad writes, include/remfx_hip.h)"""
    m = torch.arange(plan.M)
    acc = torch.zeros(plan.K, plan.M, dtype=torch.float64)
    for n in range(plan.N):
        B = _gather(plan, x_flat, n)
        oi = _out_index(plan, n)
        G = g_flat[(m[:, None] * plan.out_cs + oi[None, :])]
        acc += B.double() @ G.double().t()
    return acc.float().t().contiguous()


def scatter_weights(plan, dapack, wshape):
    dw = torch.zeros(int(np.prod(wshape)))
    nrows = plan.extra["n_weight_rows"]
    m = torch.arange(plan.M)
    idx = m[None, :] * plan.w_ms + torch.from_numpy(plan.woff.astype(np.int64))[:, None]
    dw.index_put_((idx.reshape(-1),), dapack[:, :nrows].t().reshape(-1), accumulate=True)
    return dw.view(wshape)
