"""fp8-e4m3 KV cache through fa_decode_kernel (GQA groups of 4 and more, multi-token blocks): TB/s of the cache stream.
A/B: product (fp8-operand MFMA, FA_DEC_F8M=1) vs `define_variant.py f8m0 fa_decode.hip -DFA_DEC_F8M=0` (dequantise while staging)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_decode import run
f8 = torch.float8_e4m3fn
print("lib =", os.environ.get("FA_MI355_LIB", "product"))
run(B=128, H=32, Hk=8, kv_dtype=f8)                  # G = 4
run(B=128, H=64, Hk=8, kv_dtype=f8)                  # G = 8  (config 4 at H_k = 8)
run(B=64, H=64, Hk=8, L=16384, kv_dtype=f8)
run(B=8, H=64, Hk=8, kv_dtype=f8)
run(B=128, H=32, Hk=8)                               # 16-bit cache, same shape
run(B=128, H=64, Hk=8)
