"""Wall time of each of the first N fwd+bwd steps of the bench workload (is the warm-up long enough?)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
B, S, H, D = 8, 4096, 16, 128
g = torch.Generator(device="cpu").manual_seed(421)
mk = lambda: torch.randn(B, S, H, D, generator=g).to(torch.bfloat16).cuda()
q, k, v, do = mk(), mk(), mk(), mk()
for t in (q, k, v): t.requires_grad_(True)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = flash_attn.flash_attn_func(q, k, v, causal=True); o.backward(do); q.grad = k.grad = v.grad = None
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.2f}" for t in ts))
