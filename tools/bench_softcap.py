"""softcap (Gemma-2 style) vs plain causal: forward and backward kernels (bf16 B8 S4096; H16 D128, or `python tools/bench_softcap.py 256`: H8 D256 - Gemma-2 9B's head dim)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn
from _bwdsel import bwd_call
from bench_configs import timeit
B, S, H, D = 8, 4096, 16, 128
if len(sys.argv) > 1 and sys.argv[1] == "256":
    H, D = 8, 256
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
do = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
for cap in (0.0, 0.0, 50.0):
    with torch.no_grad():
        tf = timeit(lambda: flash_attn.flash_attn_func(q, k, v, causal=True, softcap=cap))
    o = flash_attn.flash_attn_func(q, k, v, causal=True, softcap=cap)
    res = {}
    for nm in ("dkdv", "dq", "all"):
        res[nm] = timeit(bwd_call(lambda a, b, c: flash_attn.flash_attn_func(a, b, c, causal=True, softcap=cap), q, k, v, do, nm), iters=5)
    print(f"softcap={cap}: fwd {tf:.3f} | dkdv {res['dkdv']:.3f} dq {res['dq']:.3f} bwd {res['all']:.3f} ms", flush=True)
