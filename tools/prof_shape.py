"""A few launches of one kernel at a given shape (for rocprofv3 PMC passes): prof_shape.py fwd|dq|dkdv B S [H] [D]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
torch.manual_seed(421)
which, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
H = int(sys.argv[4]) if len(sys.argv) > 4 else 16
D = int(sys.argv[5]) if len(sys.argv) > 5 else 128
q, k, v, do = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
if which == "fwd":
    with torch.no_grad():
        for _ in range(6):
            flash_attn.flash_attn_func(q, k, v, causal=True)
else:
    q.requires_grad_(which == "dq"); k.requires_grad_(which == "dkdv"); v.requires_grad_(which == "dkdv")
    o = flash_attn.flash_attn_func(q, k, v, causal=True)
    ins = (q,) if which == "dq" else (k, v)
    for _ in range(6):
        torch.autograd.grad(o, ins, do, retain_graph=True)
torch.cuda.synchronize()
