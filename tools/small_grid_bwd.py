"""Backward on shapes whose dK/dV grid (batch x kv-heads x 128-key blocks) is smaller than the chip: GQA at micro-batch 1,
short-key cross-attention.  Per shape: forward, dK/dV(+preprocess), dQ, full backward (medians of evented calls) and the
number of dK/dV workgroups the unsplit launch would have.   python tools/small_grid_bwd.py [csv of shape indices | all] [nosplit]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn as fa
from _bwdsel import bwd_call


def t_ms(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2]


def pairs(Sq, Sk, causal):
    if not causal: return float(Sq) * Sk
    return float(sum(max(0, min(Sk, i + Sk - Sq + 1)) for i in range(Sq)))


SHAPES = [
    ("Llama3-8B  B1 S4096 H32/8 D128 causal", 1, 4096, 4096, 32, 8, 128, True, torch.bfloat16),
    ("Llama3-70B B1 S4096 H64/8 D128 causal", 1, 4096, 4096, 64, 8, 128, True, torch.bfloat16),
    ("Llama3-8B  B1 S2048 H32/8 D128 causal", 1, 2048, 2048, 32, 8, 128, True, torch.bfloat16),
    ("Llama3-8B  B1 S8192 H32/8 D128 causal", 1, 8192, 8192, 32, 8, 128, True, torch.bfloat16),
    ("Llama3-8B  B2 S4096 H32/8 D128 causal", 2, 4096, 4096, 32, 8, 128, True, torch.bfloat16),
    ("Llama3-8B  B3 S4096 H32/8 D128 causal", 3, 4096, 4096, 32, 8, 128, True, torch.bfloat16),
    ("MHA        B1 S4096 H8 D128 causal", 1, 4096, 4096, 8, 8, 128, True, torch.bfloat16),
    ("MQA        B4 S4096 H32/1 D128 causal", 4, 4096, 4096, 32, 1, 128, True, torch.bfloat16),
    ("Qwen2-0.5B B2 S4096 H14/2 D64 causal", 2, 4096, 4096, 14, 2, 64, True, torch.bfloat16),
    ("Qwen2-0.5B B8 S4096 H14/2 D64 causal", 8, 4096, 4096, 14, 2, 64, True, torch.bfloat16),
    ("SD-UNet x  B8 Sq4096 Sk77 H8 D40", 8, 4096, 77, 8, 8, 40, False, torch.float16),
    ("SD-UNet x  B2 Sq4096 Sk77 H8 D40", 2, 4096, 77, 8, 8, 40, False, torch.float16),
    ("SD-UNet x  B8 Sq1024 Sk77 H8 D80", 8, 1024, 77, 8, 8, 80, False, torch.float16),
    ("T5 cross   B4 Sq512 Sk128 H12 D64", 4, 512, 128, 12, 12, 64, False, torch.float16),
    ("cross      B2 Sq8192 Sk256 H16 D128", 2, 8192, 256, 16, 16, 128, False, torch.bfloat16),
    # batch x kv-heads not a multiple of the 8 XCDs (the last round of units is spread over all of them)
    ("70B / TP8  B1 S8192 H8/1 D128 causal", 1, 8192, 8192, 8, 1, 128, True, torch.bfloat16),
    ("70B / TP8  B2 S4096 H8/1 D128 causal", 2, 4096, 4096, 8, 1, 128, True, torch.bfloat16),
    ("Qwen2.5-7B B1 S4096 H28/4 D128 causal", 1, 4096, 4096, 28, 4, 128, True, torch.bfloat16),
    ("GQA        B3 S4096 H16/4 D128 causal", 3, 4096, 4096, 16, 4, 128, True, torch.bfloat16),
    ("MHA        B1 S2048 H12 D64", 1, 2048, 2048, 12, 12, 64, False, torch.float16),
    ("Gemma-2    B1 S4096 H16/8 D256 causal", 1, 4096, 4096, 16, 8, 256, True, torch.bfloat16),
    ("Gemma-2    B2 S4096 H16/8 D256 causal", 2, 4096, 4096, 16, 8, 256, True, torch.bfloat16),
]
sel = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 and sys.argv[1] != "all" else range(len(SHAPES))
NOSPLIT = len(sys.argv) > 2 and sys.argv[2] == "nosplit"     # A/B: one workgroup per key block (deterministic=True: FA_FLAG_NO_DKV_SPLIT)
if NOSPLIT:
    print("(FA_FLAG_NO_DKV_SPLIT)")
for i in sel:
    name, B, Sq, Sk, Hq, Hk, D, causal, dt = SHAPES[i]
    torch.manual_seed(i)
    q = torch.randn(B, Sq, Hq, D, device="cuda", dtype=dt, requires_grad=True)
    k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt, requires_grad=True)
    v = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt, requires_grad=True)
    do = torch.randn_like(q)
    f = lambda a, b, c: fa.flash_attn_func(a, b, c, causal=causal, deterministic=NOSPLIT)
    with torch.no_grad():
        tf = t_ms(lambda: f(q, k, v))
    r = {nm: t_ms(bwd_call(f, q, k, v, do, nm), n=8) for nm in ("dkdv", "dq", "all")}
    fl = 4.0 * B * Hq * D * pairs(Sq, Sk, causal)
    nkb = (Sk + 127) // 128
    wgs = B * Hk * ((nkb + 1) // 2 if causal and nkb >= 2 else nkb)
    print(f"{name:40s} dK/dV grid {wgs:5d} | fwd {tf:7.3f} ms {fl / tf / 1e9:6.0f} TF | dkdv(+pre) {r['dkdv']:7.3f} ms {2 * fl / r['dkdv'] / 1e9:6.0f} TF"
          f" | dq {r['dq']:7.3f} | bwd {r['all']:7.3f} ms {2.5 * fl / r['all'] / 1e9:6.0f} TF | fwd+bwd {3.5 * fl / (tf + r['all']) / 1e9:6.0f} TF",
          flush=True)
