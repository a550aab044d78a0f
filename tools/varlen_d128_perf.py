"""flash_attn_varlen_func forward / backward at D = 128 (packed training batches): the asm kernels vs FA_FWD_ASM=0 FA_BWD_ASM=0."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
g = torch.Generator().manual_seed(421)
for (B, lo, hi, H, Hk) in ((32, 512, 4097, 16, 16), (64, 256, 2049, 32, 8)):
    lens = torch.randint(lo, hi, (B,), generator=g)
    cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0); T = int(cu[-1]); cu = cu.cuda()
    q = torch.randn(T, H, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(T, Hk, 128, device="cuda", dtype=torch.bfloat16); v = torch.randn(T, Hk, 128, device="cuda", dtype=torch.bfloat16)
    fn = lambda: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, int(lens.max()), int(lens.max()), causal=True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    fl = 4.0 * 128 * H * sum(int(L) * (int(L) + 1) // 2 for L in lens)
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    o = fn(); do = torch.randn_like(o)
    bw = lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True)
    for _ in range(3): bw()
    torch.cuda.synchronize()
    s.record()
    for _ in range(10): bw()
    e.record(); torch.cuda.synchronize()
    msb = s.elapsed_time(e) / 10
    print(f"B{B} lens {lo}..{hi - 1} ({T} tokens) H{H}/{Hk} causal: fwd {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s | bwd {msb:.3f} ms  {2.5 * fl / msb / 1e9:.0f} TFLOP/s", flush=True)
