"""Forward and forward + backward rates on shapes taken from common models (dense op), to spot outliers.
  python tools/model_shapes.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn as fa
def t_ms(f, n=10):
    # >= 60 ms of the same calls first: an idle socket runs its next ~35 ms of launches on a clock ramp (profiles/r06_step_ramp.txt)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); b.synchronize()
    for _ in range(max(3, min(4000, int(60.0 / max(a.elapsed_time(b), 1e-3)) + 1))): f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2]
def pairs(Sq, Sk, causal, window):
    import numpy as np
    i = np.arange(Sq)[:, None] + (Sk - Sq); j = np.arange(Sk)[None, :]
    m = np.ones((Sq, Sk), bool)
    if causal: m &= j <= i
    if window[0] >= 0: m &= j >= i - window[0]
    if window[1] >= 0: m &= j <= i + window[1]
    return float(m.sum())
SHAPES = [
    ("BERT-base  B64 S512 H12 D64", 64, 512, 512, 12, 12, 64, False, (-1, -1), 0.0, torch.float16),
    ("ViT-L      B128 S257 H16 D64", 128, 257, 257, 16, 16, 64, False, (-1, -1), 0.0, torch.float16),
    ("GPT-2      B32 S1024 H12 D64 causal", 32, 1024, 1024, 12, 12, 64, True, (-1, -1), 0.0, torch.float16),
    ("Llama-7B   B4 S4096 H32 D128 causal", 4, 4096, 4096, 32, 32, 128, True, (-1, -1), 0.0, torch.bfloat16),
    ("Llama3-8B  B2 S8192 H32/8 D128 causal", 2, 8192, 8192, 32, 8, 128, True, (-1, -1), 0.0, torch.bfloat16),
    ("Llama3-70B B1 S8192 H64/8 D128 causal", 1, 8192, 8192, 64, 8, 128, True, (-1, -1), 0.0, torch.bfloat16),
    ("Gemma-2    B2 S4096 H16/8 D256 softcap50 causal", 2, 4096, 4096, 16, 8, 256, True, (-1, -1), 50.0, torch.bfloat16),
    ("Gemma-2    B2 S8192 H16/8 D256 softcap50 window4096", 2, 8192, 8192, 16, 8, 256, True, (4095, 0), 50.0, torch.bfloat16),
    ("Mistral    B1 S16384 H32/8 D128 window4096", 1, 16384, 16384, 32, 8, 128, True, (4095, 0), 0.0, torch.bfloat16),
    ("Whisper-x  B16 Sq448 Sk1500 H20 D64", 16, 448, 1500, 20, 20, 64, False, (-1, -1), 0.0, torch.float16),
    ("Phi-3-mini B4 S4096 H32 D96 causal", 4, 4096, 4096, 32, 32, 96, True, (-1, -1), 0.0, torch.bfloat16),
    ("Qwen2-0.5B B8 S4096 H14/2 D64 causal", 8, 4096, 4096, 14, 2, 64, True, (-1, -1), 0.0, torch.bfloat16),
    ("SD-UNet    B8 S4096 H8 D40", 8, 4096, 4096, 8, 8, 40, False, (-1, -1), 0.0, torch.float16),
    ("SD-UNet x  B8 Sq4096 Sk77 H8 D40", 8, 4096, 77, 8, 8, 40, False, (-1, -1), 0.0, torch.float16),
    ("DiT        B16 S1024 H16 D72", 16, 1024, 1024, 16, 16, 72, False, (-1, -1), 0.0, torch.float16),
]
for (name, B, Sq, Sk, Hq, Hk, D, causal, window, cap, dt) in SHAPES:
    q = torch.randn(B, Sq, Hq, D, device="cuda", dtype=dt, requires_grad=True)
    k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt, requires_grad=True)
    v = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt, requires_grad=True)
    do = torch.randn_like(q)
    kw = dict(causal=causal, window_size=window, softcap=cap)
    with torch.no_grad():
        tf = t_ms(lambda: fa.flash_attn_func(q, k, v, **kw))
    def fb():
        torch.autograd.grad(fa.flash_attn_func(q, k, v, **kw), (q, k, v), do)
    tfb = t_ms(fb, n=6)
    fl = 4.0 * B * Hq * D * pairs(Sq, Sk, causal, window)
    print(f"{name:52s} fwd {tf:7.3f} ms {fl / tf / 1e9:6.0f} TF | fwd+bwd {tfb:7.3f} ms {3.5 * fl / tfb / 1e9:6.0f} TF", flush=True)
