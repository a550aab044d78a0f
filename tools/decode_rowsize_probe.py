"""Is fp8 decode limited by the 128-byte row pieces?  Same bytes per token-head, different D / dtype."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, bench_decode
bench_decode.run(B=128, H=32, Hk=32, D=128, L=8192, kv_dtype=torch.float8_e4m3fn)      # 128-B rows
bench_decode.run(B=128, H=32, Hk=32, D=64, L=8192, kv_dtype=torch.float16)             # 128-B rows, fp16
bench_decode.run(B=128, H=32, Hk=32, D=128, L=8192, kv_dtype=torch.float16, rotary=False)
bench_decode.run(B=128, H=32, Hk=32, D=128, L=8192, kv_dtype=torch.float16, page=8192 + 256)   # one page per sequence
