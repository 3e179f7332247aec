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


def _gather_tap(plan, x_flat, n):
    """B matrix [Kpad_t, P] as the tap-major kernels (csrc/gemm_tap.h) gather it: row k' = g*8 + i of group
    g = t*gpt + c8 reads channel 8*c8 + i of tap t; offsets at / beyond the sample's extent read 0 (buffer range
    check), which is how the padded channels of the last group vanish."""
    tab = torch.from_numpy(plan.tap_tab.astype(np.int64))
    G = plan.ntaps * plan.gpt
    P = plan.OA * plan.OB
    j = torch.arange(P)
    a, b = j // plan.OB, j % plan.OB
    ia0, ib0 = a * plan.SA, b * plan.SB
    pos = ia0 * plan.in_as + ib0 * plan.in_bs
    rows = []
    for kk in range(plan.Kpad_t):
        g, i = kk // 8, kk % 8
        t, c8 = (g // plan.gpt, g % plan.gpt) if g < G else (plan.ntaps, 0)       # tail groups: invalid table rows
        off, da, db = int(tab[t, 0]), int(tab[t, 1]), int(tab[t, 2])
        ok = ((ia0 + da) >= 0) & ((ia0 + da) < plan.IA) & ((ib0 + db) >= 0) & ((ib0 + db) < plan.IB)
        rel = pos + off + (8 * c8 + i) * plan.in_cs
        ok = ok & (rel >= 0) & (rel < plan.in_extent)
        idx = (rel + n * plan.in_ns).clamp(0, x_flat.numel() - 1)
        rows.append(x_flat[idx] * ok)
    return torch.stack(rows, 0)


def emulate_fwd(plan, w_flat, x_flat, out_flat, bias=None):
    nrows = plan.extra["n_weight_rows"]
    m = torch.arange(plan.M)
    A = w_flat[(m[None, :] * plan.w_ms + torch.from_numpy(plan.woff.astype(np.int64))[:, None])]  # [K, M]
    if getattr(plan, "cin", 0) >= 8:
        wt = torch.from_numpy(plan.woff_t.astype(np.int64))
        At = w_flat[(m[None, :] * plan.w_ms + wt.clamp(min=0)[:, None])] * (wt >= 0)[:, None]       # [Kpad_t, M]
    for n in range(plan.N):
        B = _gather(plan, x_flat, n)[:nrows]
        o = A.t().double() @ B.double()
        if getattr(plan, "cin", 0) >= 8:          # the tap-major form of the same plan must give the same product
            ot = At.t().double() @ _gather_tap(plan, x_flat, n).double()
            torch.testing.assert_close(ot, o, rtol=1e-9, atol=1e-9)
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
