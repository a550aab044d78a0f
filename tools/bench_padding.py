"""HBM rate of the row gather / scatter kernels behind flash_attn.bert_padding (unpad_input / pad_input) against the
plain torch indexing they replace.  Algorithmic bytes: gather = 2 x moved rows (+ 8 B index per row);
scatter = moved rows read + every destination row written once."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from flash_attn import bert_padding as bp
from bench_configs import timeit

PEAK = 8000.0
for (B, S, H, D, fill) in [(64, 2048, 32, 64, 0.5), (64, 2048, 32, 128, 0.5), (16, 8192, 32, 128, 0.9)]:
    x = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    lens = (torch.rand(B) * 2 * fill * S).clamp(1, S).long()
    lens[0] = S
    mask = (torch.arange(S)[None, :] < lens[:, None]).cuda()
    packed, idx, cu, mx, _ = bp.unpad_input(x, mask)
    flat = x.reshape(B * S, H, D)
    rb = H * D * 2
    n = idx.numel()
    t_g = timeit(lambda: bp.index_first_axis(flat, idx))
    t_g0 = timeit(lambda: flat.index_select(0, idx))
    t_s = timeit(lambda: bp.pad_input(packed, idx, B, S))
    idx_plain = idx.clone()                                   # no sorted marker: memset + scatter
    t_s2 = timeit(lambda: bp.pad_input(packed, idx_plain, B, S))
    def torch_pad():
        out = packed.new_zeros((B * S, H, D)); out.index_copy_(0, idx, packed); return out
    t_s0 = timeit(torch_pad)
    gb_g = (2 * n * rb + 8 * n) / 1e9
    gb_s = (n * rb + B * S * rb + 8 * n) / 1e9
    print(f"B{B} S{S} H{H} D{D} fill {n / (B * S):.2f} ({n * rb / 1e9:.2f} GB of rows): "
          f"gather {t_g:.3f} ms {gb_g / t_g * 1e3:6.0f} GB/s ({gb_g / t_g * 1e3 / PEAK:.0%}) [torch {t_g0:.3f} ms] | "
          f"pad {t_s:.3f} ms {gb_s / t_s * 1e3:6.0f} GB/s ({gb_s / t_s * 1e3 / PEAK:.0%}) [memset+scatter {t_s2:.3f} ms, torch {t_s0:.3f} ms]", flush=True)
