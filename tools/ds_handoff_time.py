"""config-2 backward with / without the dS hand-off (FA_BWD_DS=0/1), for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import flash_attn
B, S, H, D = 8, 4096, 16, 128
g = torch.Generator(device="cuda").manual_seed(1)
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, generator=g).requires_grad_(True) for _ in range(3))
do = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, generator=g)
o = flash_attn.flash_attn_func(q, k, v, causal=True)
f = lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True)
for _ in range(5): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(n): f()
e.record(); torch.cuda.synchronize()
print("FA_BWD_DS =", os.environ.get("FA_BWD_DS", "default"), f"backward {s.elapsed_time(e) / n:.4f} ms")
