"""Per-kernel times of BASELINE config 3 (varlen, window (512,0), D 64) + dense equivalents for comparison."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn
from flash_attn_mi355 import _lib, flash_attn_interface as _fi
from bench_configs import timeit
from _bwdsel import bwd_call
g = torch.Generator().manual_seed(421)
B, H, D, W = 64, 32, 64, 512
lens = torch.randint(64, 2049, (B,), generator=g); lens[0] = 2048
cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0); T = int(cu[-1]); cu = cu.cuda()
torch.manual_seed(421)
def mk(*s): return torch.randn(*s, device="cuda", dtype=torch.float16, requires_grad=True)
def pairs(L, W): return L * (L + 1) // 2 if (W < 0 or L <= W + 1) else (W + 1) * (W + 2) // 2 + (L - W - 1) * (W + 1)
def run(name, fwd, flops):
    with torch.no_grad():
        tf = timeit(lambda: fwd(q, k, v))
    do = torch.randn_like(fwd(q, k, v))
    res = {}
    for nm in ("dkdv", "dq", "all"):       # gradients the op has to produce -> kernels launched (tools/_bwdsel.py)
        res[nm] = timeit(bwd_call(fwd, q, k, v, do, nm))
    print(f"{name:34s} fwd {tf:.3f} ms ({flops/tf/1e9:6.0f} TF) | dkdv(+pre) {res['dkdv']:.3f} ({2*flops/res['dkdv']/1e9:5.0f} TF) dq {res['dq']:.3f} all {res['all']:.3f} ({2.5*flops/res['all']/1e9:5.0f} TF)", flush=True)
q, k, v = mk(T, H, D), mk(T, H, D), mk(T, H, D); INS = (q, k, v)
fl = 4.0 * D * H * sum(pairs(int(L), W) for L in lens)
run("cfg3 varlen window(512,0) D64", lambda q, k, v: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True, window_size=(W, 0)), fl)
fl = 4.0 * D * H * sum(pairs(int(L), -1) for L in lens)
run("varlen causal (no window) D64", lambda q, k, v: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True), fl)
Bd, S = 32, 2048
q, k, v = mk(Bd, S, H, D), mk(Bd, S, H, D), mk(Bd, S, H, D); INS = (q, k, v)
run("dense causal B32 S2048 D64", lambda q, k, v: flash_attn.flash_attn_func(q, k, v, causal=True), 4.0 * D * H * Bd * S * S / 2)
run("dense window(512,0) B32 S2048 D64", lambda q, k, v: flash_attn.flash_attn_func(q, k, v, causal=True, window_size=(W, 0)), 4.0 * D * H * Bd * pairs(S, W))
q, k, v = mk(Bd, S, 16, 128), mk(Bd, S, 16, 128), mk(Bd, S, 16, 128); INS = (q, k, v)
run("dense causal B32 S2048 H16 D128", lambda q, k, v: flash_attn.flash_attn_func(q, k, v, causal=True), 4.0 * 128 * 16 * Bd * S * S / 2)
