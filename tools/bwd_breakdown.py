"""Per-kernel backward times for a dense shape: python tools/bwd_breakdown.py B S H D [causal]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn
from _bwdsel import bwd_call
from bench_configs import timeit
B, S, H, D = (int(x) for x in sys.argv[1:5])
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
do = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
fl = 4.0 * B * H * S * S * D / 2
with torch.no_grad():
    tf = timeit(lambda: flash_attn.flash_attn_func(q, k, v, causal=True))
o = flash_attn.flash_attn_func(q, k, v, causal=True)
res = {}
for nm in ("dkdv", "dq", "all"):
    res[nm] = timeit(bwd_call(lambda a, b, c: flash_attn.flash_attn_func(a, b, c, causal=True), q, k, v, do, nm), iters=5)
print(f"B{B} S{S} H{H} D{D}: fwd {tf:.3f} ms ({fl/tf/1e9:.0f} TF) | dkdv(+pre) {res['dkdv']:.3f} ({2*fl/res['dkdv']/1e9:.0f} TF) dq {res['dq']:.3f} ({0.5*fl/res['dq']/1e9:.0f} TF alg) all {res['all']:.3f}")
