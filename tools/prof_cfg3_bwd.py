"""BASELINE config 3 backward loop (for rocprofv3 PMC / trace passes): prof_cfg3_bwd.py [n]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
B, H, D, W = 64, 32, 64, 512
g = torch.Generator().manual_seed(421)
lens = torch.randint(64, 2049, (B,), generator=g); lens[0] = 2048
cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0); T = int(cu[-1]); cu = cu.cuda()
gq = torch.Generator().manual_seed(422)
q, k, v, do = (torch.randn(T, H, D, generator=gq).to(torch.float16).cuda() for _ in range(4))
q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
o = flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True, window_size=(W, 0))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    torch.autograd.grad(o, (q, k, v), do, retain_graph=True)
torch.cuda.synchronize()
