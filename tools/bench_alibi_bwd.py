"""Causal ALiBi vs no bias, fwd and bwd kernels (bf16 B8 H16 S4096 D128)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn
from _bwdsel import bwd_call
from bench_configs import timeit
B, S, H, D = 8, 4096, 16, 128
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
do = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
slopes = (2.0 ** (-8.0 * (torch.arange(H) + 1) / H)).float().cuda()
for name, sl in (("no bias", None), ("alibi", slopes)):
    with torch.no_grad():
        tf = timeit(lambda: flash_attn.flash_attn_func(q, k, v, causal=True, alibi_slopes=sl))
    o = flash_attn.flash_attn_func(q, k, v, causal=True, alibi_slopes=sl)
    res = {}
    for nm in ("dkdv", "dq", "all"):
        res[nm] = timeit(bwd_call(lambda a, b, c: flash_attn.flash_attn_func(a, b, c, causal=True, alibi_slopes=sl), q, k, v, do, nm), iters=5)
    print(f"{name:8s}: fwd {tf:.3f} | dkdv {res['dkdv']:.3f} dq {res['dq']:.3f} bwd {res['all']:.3f} ms", flush=True)
