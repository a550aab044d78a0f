"""Dense forward / forward + backward over a grid of batch x heads x sequence length x head dim x mask, to spot launch-shape
outliers (a row far below its neighbours).   python tools/shape_sweep.py [D ...]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn as fa

def t_ms(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

DS = [int(x) for x in sys.argv[1:]] or [128, 64]
HEADS = [(8, 8), (32, 8), (8, 1), (12, 12), (28, 4), (16, 2)]
for D in DS:
    for causal in (True, False):
        print(f"== D {D} {'causal' if causal else 'full'}: fwd TF / fwd+bwd TF (back-to-back means)")
        print(f"{'':14s}" + "".join(f"{f'S {S}':>16s}" for S in (512, 1024, 2048, 4096, 8192)))
        for (H, Hk) in HEADS:
            for B in (1, 2, 3, 4, 8):
                cells = []
                for S in (512, 1024, 2048, 4096, 8192):
                    if B * H * S > 32 * 8192 * 2:
                        cells.append(f"{'-':>16s}"); continue
                    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
                    k = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
                    v = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
                    do = torch.randn_like(q)
                    fl = 4.0 * B * H * D * (S * (S + 1) / 2 if causal else S * S)
                    with torch.no_grad():
                        tf = t_ms(lambda: fa.flash_attn_func(q, k, v, causal=causal))
                    tfb = t_ms(lambda: torch.autograd.grad(fa.flash_attn_func(q, k, v, causal=causal), (q, k, v), do), n=6)
                    cells.append(f"{fl / tf / 1e9:7.0f}/{3.5 * fl / tfb / 1e9:<7.0f} ")
                print(f"B{B} H{H}/{Hk}".ljust(14) + "".join(cells), flush=True)
