"""Run a few fwd+bwd steps of the bench workload (for rocprofv3)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn
torch.manual_seed(421)
B, S, H, D = 8, 4096, 16, 128
q, k, v, do = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n):
    o = flash_attn.flash_attn_func(q, k, v, causal=True)
    o.backward(do)
    q.grad = k.grad = v.grad = None
torch.cuda.synchronize()
