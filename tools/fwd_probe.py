import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import flash_attn
from quick_perf import timeit
q, k, v = (torch.randn(8, 4096, 16, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
for causal in (False, True):
    for _ in range(2):
        med, mn = timeit(lambda: flash_attn.flash_attn_func(q, k, v, causal=causal), warm=10, it=20)
    print(f"causal={causal}: {med:.3f} ms (min {mn:.3f})", flush=True)
