"""How much of config 3's forward time is empty / unbalanced workgroups?  Same total tokens, three length mixes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch
import flash_attn
from bench_configs import timeit

def run(lens, tag):
    B, H, D, W = len(lens), 32, 64, 512
    cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = torch.tensor(lens).cumsum(0)
    T = int(cu[-1]); cu = cu.cuda(); mx = max(lens)
    q, k, v = (torch.randn(T, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
    def pairs(L): return L * (L + 1) // 2 if L <= W + 1 else (W + 1) * (W + 2) // 2 + (L - W - 1) * (W + 1)
    flops = 4.0 * D * H * sum(pairs(int(L)) for L in lens)
    with torch.no_grad():
        t = timeit(lambda: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, mx, mx, causal=True, window_size=(W, 0)))
    print(f"{tag:28s} tokens {T:6d} max {mx:5d}: fwd {t:.3f} ms {flops / t / 1e9:6.1f} TF", flush=True)

g = torch.Generator().manual_seed(421)
lens = torch.randint(64, 2049, (64,), generator=g); lens[0] = 2048
lens = [int(x) for x in lens]
run(lens, "mixed (config 3)")
run(sorted(lens, reverse=True), "mixed, longest first")
avg = sum(lens) // len(lens)
run([avg] * 64, "uniform (same tokens)")
run([2048] * 33, "uniform 2048 (same tokens)")
run(lens, "mixed (config 3) again")
run(sorted(lens), "mixed, shortest first")
