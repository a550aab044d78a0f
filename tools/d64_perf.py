"""Forward at D = 64 across shapes: the asm bodies (FA_FWD_ASM64=1) vs the compiler kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
torch.manual_seed(421)
for (B, S, H, causal) in ((8, 4096, 32, True), (8, 4096, 32, False), (4, 8192, 32, True), (2, 16384, 32, True), (2, 16384, 32, False)):
    q, k, v = (torch.randn(B, S, H, 64, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    fn = lambda: flash_attn.flash_attn_func(q, k, v, causal=causal)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    fl = 4.0 * B * H * S * S * 64 * (0.5 if causal else 1.0)
    print(f"B{B} S{S} H{H} D64 causal={causal}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
