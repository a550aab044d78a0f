"""BASELINE config 3 forward (and optionally backward) timed N times in one process, for A/B of library variants
(FA_MI355_LIB=tools/variants/libfa_<name>.so): median / min of evented launches + max |diff| against the first run's output.
  python tools/cfg3_fwd.py [bwd]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn
B, H, D, W = 64, 32, 64, 512
g = torch.Generator().manual_seed(421)
lens = torch.randint(64, 2049, (B,), generator=g); lens[0] = 2048
cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0); T = int(cu[-1]); cu = cu.cuda()
gq = torch.Generator().manual_seed(422)
q, k, v, do = (torch.randn(T, H, D, generator=gq).to(torch.float16).cuda() for _ in range(4))
def pairs(L): return L * (L + 1) // 2 if L <= W + 1 else (W + 1) * (W + 2) // 2 + (L - W - 1) * (W + 1)
flops = 4.0 * D * H * sum(pairs(int(L)) for L in lens)
def ev(fn, n=30, warm=5):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2], ts[0]
fwd = lambda: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True, window_size=(W, 0))
def sustained(fn, n=200):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n
with torch.no_grad():
    med, mn = ev(fwd)
    sus = sustained(fwd)
print(f"lib={os.environ.get('FA_MI355_LIB', 'product')}: cfg3 fwd {med:.4f} ms (min {mn:.4f}) = {flops / med / 1e9:.0f} TFLOP/s | back-to-back {sus:.4f} ms = {flops / sus / 1e9:.0f} TFLOP/s", flush=True)
fwdc = lambda: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True)
flc = 4.0 * D * H * sum(int(L) * (int(L) + 1) // 2 for L in lens)
with torch.no_grad():
    med, mn = ev(fwdc)
print(f"   varlen causal (no window) fwd {med:.4f} ms (min {mn:.4f}) = {flc / med / 1e9:.0f} TFLOP/s", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "bwd":
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    def fb():
        o = fwd(); o.backward(do); q.grad = k.grad = v.grad = None
    med, mn = ev(fb, n=20)
    print(f"   cfg3 fwd+bwd {med:.4f} ms (min {mn:.4f}) = {3.5 * flops / med / 1e9:.0f} TFLOP/s", flush=True)
