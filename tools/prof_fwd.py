"""Forward-only loop of the bench workload (for rocprofv3 PMC passes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
torch.manual_seed(421)
causal = "--nc" not in sys.argv
q, k, v = (torch.randn(8, 4096, 16, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
with torch.no_grad():
    for _ in range(4):
        flash_attn.flash_attn_func(q, k, v, causal=causal)
torch.cuda.synchronize()
