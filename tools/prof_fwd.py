"""Forward-only loop of the bench workload (for rocprofv3 PMC passes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
torch.manual_seed(421)
causal = "--nc" not in sys.argv
if "--cfg3" in sys.argv:          # BASELINE config 3: varlen fp16, H32 D64, window (512, 0)
    g = torch.Generator().manual_seed(421)
    B, H, D, W = 64, 32, 64, 512
    lens = torch.randint(64, 2049, (B,), generator=g); lens[0] = 2048
    cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0); T = int(cu[-1]); cu = cu.cuda()
    q, k, v = (torch.randn(T, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
    with torch.no_grad():
        for _ in range(4):
            flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True, window_size=(W, 0))
else:
    q, k, v = (torch.randn(8, 4096, 16, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    with torch.no_grad():
        for _ in range(4):
            flash_attn.flash_attn_func(q, k, v, causal=causal)
torch.cuda.synchronize()
