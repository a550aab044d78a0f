"""fwd / fwd+bwd TFLOP/s across head dims (bf16 causal, 16K tokens per batch x heads fixed).
  python tools/bench_headdims.py [D ...]      (default: all)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn
from bench_configs import timeit
ONLY = [int(x) for x in sys.argv[1:]]
for D, H in ((64, 32), (96, 16), (128, 16), (192, 8), (256, 8)):
    if ONLY and D not in ONLY:
        continue
    B, S = 8, 4096
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    do = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    fl = 4.0 * B * H * S * S * D / 2
    with torch.no_grad():
        tf = timeit(lambda: flash_attn.flash_attn_func(q, k, v, causal=True))
    def fb():
        o = flash_attn.flash_attn_func(q, k, v, causal=True); o.backward(do); q.grad = k.grad = v.grad = None
    tfb = timeit(fb, iters=5)
    print(f"D{D:3d} H{H:2d}: fwd {tf:.3f} ms {fl/tf/1e9:6.0f} TF | fwd+bwd {tfb:.3f} ms {3.5*fl/tfb/1e9:6.0f} TF", flush=True)
