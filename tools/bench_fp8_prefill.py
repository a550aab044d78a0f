"""Chunked prefill over a KV cache: T_q new tokens attend to L cached ones (+ themselves, causal), 16-bit vs fp8-e4m3 cache.
fp8 caches with T_q x group > 32 run fa_decode_kernel in 32-row blocks (round 3); 16-bit caches run fa_fwd_kernel."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn
from bench_configs import timeit
torch.manual_seed(421)
for (B, L, Tq, H, Hk, D) in ((8, 8192, 512, 32, 8, 128), (8, 8192, 128, 32, 32, 128), (16, 4096, 64, 32, 8, 128), (8, 8192, 16, 32, 8, 128)):
    S = L + Tq
    q = torch.randn(B, Tq, H, D, device="cuda", dtype=torch.bfloat16)
    kc = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16); vc = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16)
    k8 = (kc.float() / 0.05).to(torch.float8_e4m3fn); v8 = (vc.float() / 0.05).to(torch.float8_e4m3fn)
    sl = torch.full((B,), S, dtype=torch.int32, device="cuda")
    fl = 4.0 * B * H * Tq * (L + Tq / 2) * D
    t16 = timeit(lambda: flash_attn.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=sl, causal=True), iters=5)
    t8 = timeit(lambda: flash_attn.flash_attn_with_kvcache(q, k8, v8, cache_seqlens=sl, causal=True, k_descale=0.05, v_descale=0.05), iters=5)
    gb16 = 2.0 * B * S * Hk * D * 2 / 1e9
    print(f"B{B} L{L} Tq{Tq} H{H}/{Hk}: 16-bit cache {t16:.3f} ms ({fl / t16 / 1e9:.0f} TFLOP/s, {gb16 / t16:.2f} TB/s of KV)   "
          f"fp8 cache {t8:.3f} ms ({fl / t8 / 1e9:.0f} TFLOP/s, {gb16 / 2 / t8:.2f} TB/s of KV)", flush=True)
