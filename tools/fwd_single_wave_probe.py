"""Forward on causal shapes whose 256-row blocks all get a CU at once (one wave of unpaired workgroups: the heaviest block sets
the time): the hand-scheduled 256-row kernel against the 128-row compiler kernel (FA_FWD_ASM=0), which has twice the blocks to
balance.   python tools/fwd_single_wave_probe.py   (run once per setting of FA_FWD_ASM)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn as fa

def t_ms(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

SHAPES = [(1, 8192, 8, 1), (1, 4096, 8, 1), (2, 4096, 8, 1), (1, 2048, 32, 8), (1, 4096, 8, 8), (1, 4096, 16, 16), (1, 4096, 32, 8), (1, 1024, 32, 8),
          (2, 2048, 32, 8), (1, 8192, 32, 8), (4, 1024, 16, 16)]
print("FA_FWD_ASM =", os.environ.get("FA_FWD_ASM", "(default)"))
for (B, S, H, Hk) in SHAPES:
    q = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, S, Hk, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, S, Hk, 128, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        t = t_ms(lambda: fa.flash_attn_func(q, k, v, causal=True))
    fl = 4.0 * B * H * 128 * S * (S + 1) / 2
    print(f"B{B} S{S} H{H}/{Hk}: 256-row blocks {B * H * ((S + 255) // 256):5d} | fwd {t * 1e3:8.1f} us {fl / t / 1e9:6.0f} TF", flush=True)
