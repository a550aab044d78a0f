"""BASELINE config 3 backward, per kernel (dK/dV(+pre), dQ, both), evented median + back-to-back mean, for A/B of library
variants (FA_MI355_LIB=tools/variants/libfa_<name>.so); prints gradient checksums so that a variant that changes results shows.
  python tools/cfg3_bwd.py [dense]     (dense: B32 S2048 H32 D64 causal and window as well)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn
from _bwdsel import bwd_call
B, H, D, W = 64, 32, 64, 512
g = torch.Generator().manual_seed(421)
lens = torch.randint(64, 2049, (B,), generator=g); lens[0] = 2048
cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0); T = int(cu[-1]); cu = cu.cuda()
gq = torch.Generator().manual_seed(422)
def pairs(L, W): return L * (L + 1) // 2 if (W < 0 or L <= W + 1) else (W + 1) * (W + 2) // 2 + (L - W - 1) * (W + 1)
def ev(fn, n=20, warm=5):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2], ts[0]
def sustained(fn, n=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n
def run(name, fwd, q, k, v, flops):
    do = torch.randn(q.shape, generator=gq).to(q.dtype).cuda()
    out = []
    for nm, fl in (("dkdv", 2 * flops), ("dq", 0.5 * flops), ("all", 2.5 * flops)):
        f = bwd_call(fwd, q, k, v, do, nm)
        med, mn = ev(f); sus = sustained(f)
        out.append(f"{nm} {med:.4f} (min {mn:.4f}, b2b {sus:.4f}) {fl / sus / 1e9:4.0f} TF")
    gr = bwd_call(fwd, q, k, v, do, "all")()
    cs = " ".join(f"{float(x.float().abs().sum()):.6e}" for x in gr)
    print(f"lib={os.path.basename(os.environ.get('FA_MI355_LIB', 'product'))} {name}: " + " | ".join(out) + f" | sums {cs}", flush=True)
def mk(*s): return torch.randn(*s, generator=gq).to(torch.float16).cuda().requires_grad_(True)
q, k, v = mk(T, H, D), mk(T, H, D), mk(T, H, D)
run("cfg3", lambda q, k, v: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True, window_size=(W, 0)), q, k, v,
    4.0 * D * H * sum(pairs(int(L), W) for L in lens))
if "dense" in sys.argv:
    Bd, S = 32, 2048
    q, k, v = mk(Bd, S, H, D), mk(Bd, S, H, D), mk(Bd, S, H, D)
    run("dense causal B32 S2048 D64", lambda q, k, v: flash_attn.flash_attn_func(q, k, v, causal=True), q, k, v, 4.0 * D * H * Bd * S * S / 2)
    run("dense window512 B32 S2048 D64", lambda q, k, v: flash_attn.flash_attn_func(q, k, v, causal=True, window_size=(W, 0)), q, k, v, 4.0 * D * H * Bd * pairs(S, W))
