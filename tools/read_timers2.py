"""Development aid: s_memtime phase sums of the two-per-CU dK/dV kernel over BASELINE config 3 (all waves, lane 0).
  build:  python tools/define_variant.py tm2 fa_bwd.hip -DFA_TIMERS2 [-DFA_DKV2_OCC64=2]
  run:    FA_MI355_LIB=tools/variants/libfa_tm2.so python tools/read_timers2.py"""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn
from flash_attn_mi355 import _lib
B, H, D, W = 64, 32, 64, 512
g = torch.Generator().manual_seed(421)
lens = torch.randint(64, 2049, (B,), generator=g); lens[0] = 2048
cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0); T = int(cu[-1]); cu = cu.cuda()
gq = torch.Generator().manual_seed(422)
q, k, v, do = (torch.randn(T, H, D, generator=gq).to(torch.float16).cuda() for _ in range(4))
q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
o = flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True, window_size=(W, 0))
NREC = 131072
buf = (ctypes.c_ulonglong * (16 * NREC))()
f = _lib.lib.fa_debug_read_timers2
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(4):
    torch.autograd.grad(o, (q, k, v), do, retain_graph=True)
torch.cuda.synchronize(); print("rc", f(buf, NREC))
import numpy as np
rec = np.frombuffer(buf, dtype=np.uint64).reshape(NREC, 16).astype(np.float64)
rec = rec[rec[:, 12] == 1]
t = rec.sum(0)
waves, stages, active = t[12], t[8], t[9]
names = ["barrier wait", "DMA issue + bookkeeping", "S/dP reads + MFMAs (to results)", "P/dS VALU", "dV/dK reads + MFMAs", "publish stats",
         "loop bookkeeping", "own-DMA wait (vmcnt 0)", "-", "-", "prologue", "epilogue"]
tot = sum(t[i] for i in (0, 1, 2, 3, 4, 5, 6, 7, 10, 11))
print(f"waves {waves:.0f}  stages/wave {stages / waves:.1f} (active {active / waves:.1f})  cycles/wave {tot / waves:.0f}")
for i in (10, 7, 0, 1, 2, 3, 4, 5, 6, 11):
    per = t[i] / (active if i in (2, 3, 4) else stages) if i not in (10, 11) else t[i] / waves
    unit = "per active stage" if i in (2, 3, 4) else ("per wave" if i in (10, 11) else "per stage")
    print(f"  {names[i]:34s} {100 * t[i] / tot:5.1f} %   {per:8.0f} cycles {unit}")
