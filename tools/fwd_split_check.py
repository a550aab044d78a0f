"""Forward key split of one-wave causal launches (fa_fwd_asm.hip, opt-in FA_FLAG_FWD_KEY_SPLIT): output / LSE against the unsplit
kernel and back-to-back times -> profiles/r06_fwd_split.txt.   python tools/fwd_split_check.py"""
import os, sys, warnings, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import flash_attn
from flash_attn_mi355 import flash_attn_interface as fi
warnings.simplefilter("ignore")
def b2b(fn, n=500):
    # >= 60 ms of the same calls first (an idle socket runs its next ~35 ms of launches on a clock ramp, profiles/r06_step_ramp.txt: the
    # first version of this tool timed 50 launches behind 10 - ~4 ms in all - and measured the split variant FIRST, i.e. colder)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); b.synchronize()
    for _ in range(min(4000, int(60.0 / max(a.elapsed_time(b), 1e-3)) + 1)): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n
for (B, S, H, Hk, Sk) in ((1, 2048, 32, 32, None), (1, 2048, 32, 8, None), (1, 8192, 8, 1, None), (2, 4096, 8, 8, None), (1, 1024, 64, 8, None),
                          (1, 2048, 16, 16, None), (1, 1900, 32, 32, None), (1, 2048, 32, 32, 3000), (1, 4096, 16, 16, None), (4, 2048, 8, 8, None)):
    Sk = Sk or S
    g = torch.Generator(device="cuda").manual_seed(7)
    q = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16, generator=g)
    k = torch.randn(B, Sk, Hk, 128, device="cuda", dtype=torch.bfloat16, generator=g)
    v = torch.randn(B, Sk, Hk, 128, device="cuda", dtype=torch.bfloat16, generator=g)
    f = lambda: flash_attn.flash_attn_func(q, k, v, causal=True, return_attn_probs=True)
    with torch.no_grad():
        fi.FWD_SPLIT = True
        o1, l1, _ = f(); t1 = b2b(f)
        fi.FWD_SPLIT = False
        o0, l0, _ = f(); t0 = b2b(f)
        d = (o1.float() - o0.float()).abs().max().item(); dl = (l1 - l0).abs().max().item()
    fl = 4.0 * B * H * 128 * (S * (S + 1) / 2 + S * (Sk - S))
    print(f"B{B} S{S} Sk{Sk} H{H}/{Hk}: max|dO| {d:.2e} max|dLSE| {dl:.2e} finite {bool(torch.isfinite(o1).all())} | split {t1*1e3:7.1f} us {fl/t1/1e9:6.0f} TF | unsplit {t0*1e3:7.1f} us {fl/t0/1e9:6.0f} TF", flush=True)
