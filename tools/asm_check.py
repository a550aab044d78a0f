"""Development check of the hand-scheduled forward (fa_fwd_asm.hip) against an fp32 torch reference on the GPU.
Prints max |dO|, max |dLSE| and, on a mismatch, which rows / columns are off (the pattern names the broken piece).
Usage: asm_check.py [quick]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import flash_attn


def ref(q, k, v, causal, window, scale):
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    qf = q.float().permute(0, 2, 1, 3)
    kf = k.float().permute(0, 2, 1, 3).repeat_interleave(H // Hk, dim=1)
    vf = v.float().permute(0, 2, 1, 3).repeat_interleave(H // Hk, dim=1)
    s = torch.matmul(qf, kf.transpose(-1, -2)) * scale
    i = torch.arange(Sq, device=q.device)[:, None]
    j = torch.arange(Sk, device=q.device)[None, :]
    off = Sk - Sq
    wl, wr = window
    if causal:
        wr = 0
    vis = torch.ones(Sq, Sk, dtype=torch.bool, device=q.device)
    if wr >= 0:
        vis &= j <= i + off + wr
    if wl >= 0:
        vis &= j >= i + off - wl
    s = s.masked_fill(~vis, float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None])
    p = torch.nan_to_num(p, nan=0.0)
    o = torch.matmul(p, vf)
    return o.permute(0, 2, 1, 3), lse


def run(B, Sq, Sk, H, Hk, causal, window=(-1, -1), dt=torch.bfloat16, spike=False):
    torch.manual_seed(421)
    D = 128
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=dt)
    k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt)
    v = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt)
    if spike:                       # force late rescales of the running maximum
        for t in range(3, Sk // 64, 5):
            k[:, 64 * t + 7] *= 6.0
    scale = D ** -0.5
    o, lse, _ = flash_attn.flash_attn_func(q, k, v, causal=causal, window_size=window, return_attn_probs=True)
    torch.cuda.synchronize()
    o_ref, lse_ref = ref(q, k, v, causal, window, scale)
    eo = (o.float() - o_ref).abs()
    fin = torch.isfinite(lse_ref)
    el = torch.where(fin, (lse - lse_ref).abs(), torch.zeros_like(lse_ref))
    el = torch.nan_to_num(el, nan=1e9)
    same_inf = bool(((~fin) == (~torch.isfinite(lse))).all())
    tol = 2e-2 if dt == torch.bfloat16 else 4e-3
    mo, ml = float(torch.nan_to_num(eo, nan=1e9).max()), float(el.max())
    ok = mo < tol and ml < 2e-3 and same_inf
    print(f"B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} causal={causal} win={window} {str(dt)[6:]} spike={spike}: max|dO|={mo:.3e} max|dLSE|={ml:.3e} "
          f"inf-pattern={'ok' if same_inf else 'BAD'} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        bad = (torch.nan_to_num(eo, nan=1e9) > tol)
        rows = bad.any(dim=3).any(dim=2).any(dim=0).nonzero().flatten()
        cols = bad.any(dim=1).any(dim=1).any(dim=0).nonzero().flatten()
        print("   bad rows:", rows[:24].tolist(), "... n =", rows.numel(), " bad cols:", cols[:24].tolist(), "n =", cols.numel())
        badl = (el > 2e-3).any(dim=1).any(dim=0).nonzero().flatten()
        print("   bad lse rows:", badl[:24].tolist(), "n =", badl.numel())
        r = int(rows[0]) if rows.numel() else 0
        print("   o[0,r,0,:8]   =", o[0, r, 0, :8].float().tolist())
        print("   ref[0,r,0,:8] =", o_ref[0, r, 0, :8].tolist())
        print("   lse[0,0,r], ref =", float(lse[0, 0, r]), float(lse_ref[0, 0, r]))
    return ok


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    cases = [
        dict(B=1, Sq=256, Sk=256, H=1, Hk=1, causal=False),
        dict(B=1, Sq=256, Sk=256, H=1, Hk=1, causal=True),
        dict(B=1, Sq=512, Sk=512, H=2, Hk=2, causal=True),
        dict(B=2, Sq=1024, Sk=1024, H=4, Hk=2, causal=False),
        dict(B=2, Sq=1024, Sk=1024, H=4, Hk=2, causal=True),
    ]
    if not quick:
        cases += [
            dict(B=1, Sq=2048, Sk=2048, H=2, Hk=1, causal=True, dt=torch.float16),
            dict(B=1, Sq=300, Sk=300, H=2, Hk=2, causal=True),
            dict(B=1, Sq=333, Sk=777, H=2, Hk=2, causal=True),
            dict(B=1, Sq=777, Sk=333, H=2, Hk=2, causal=True),
            dict(B=1, Sq=1000, Sk=1000, H=2, Hk=2, causal=False),
            dict(B=1, Sq=1024, Sk=1024, H=2, Hk=2, causal=False, window=(200, 0)),
            dict(B=1, Sq=1024, Sk=1500, H=2, Hk=2, causal=False, window=(100, 50)),
            dict(B=1, Sq=2048, Sk=2048, H=2, Hk=2, causal=True, spike=True),
            dict(B=1, Sq=2048, Sk=2048, H=2, Hk=2, causal=False, spike=True),
            dict(B=2, Sq=4096, Sk=4096, H=4, Hk=4, causal=True),
        ]
    allok = True
    for c in cases:
        allok &= run(**c)
    print("ALL OK" if allok else "SOME MISMATCH")




def diag():
    """error maps of the smallest cases: max |dO| per (32-row block, 32-column block), |dLSE| per 32-row block"""
    torch.manual_seed(421)
    D = 128
    for (Sq, Sk, causal) in ((256, 64, False), (256, 128, False), (256, 256, False), (256, 256, True)):
        q = torch.randn(1, Sq, 1, D, device="cuda", dtype=torch.bfloat16)
        k = torch.randn(1, Sk, 1, D, device="cuda", dtype=torch.bfloat16)
        v = torch.randn(1, Sk, 1, D, device="cuda", dtype=torch.bfloat16)
        o, lse, _ = flash_attn.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
        o_ref, lse_ref = ref(q, k, v, causal, (-1, -1), D ** -0.5)
        eo = torch.nan_to_num((o.float() - o_ref).abs(), nan=99.0)[0, :, 0, :]
        el = torch.nan_to_num((lse - lse_ref).abs(), nan=99.0)[0, 0]
        print(f"--- Sq{Sq} Sk{Sk} causal={causal}")
        for rb in range(Sq // 32):
            row = " ".join(f"{float(eo[32 * rb:32 * rb + 32, 32 * cb:32 * cb + 32].max()):8.1e}" for cb in range(4))
            print(f"rows {32 * rb:3d}+: dO by col block: {row}   dLSE {float(el[32 * rb:32 * rb + 32].max()):8.1e}")
        # finer: inside row block 0, per row
        r = eo[:32]
        print("row block 0, per-row max:", " ".join(f"{float(x):.0e}" for x in r.max(dim=1).values))
        print("row block 0, per-col max (cols 0..31):", " ".join(f"{float(x):.0e}" for x in r.max(dim=0).values[:32]))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "diag":
    diag()
elif __name__ == "__main__":
    main()
