"""BASELINE config 4 (decode B128 H32 D128, 8192-token paged cache + RoPE) timed N times: fp8 and fp16 KV.
  FA_MI355_LIB=<variant>.so python tools/cfg4_time.py [N]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn, bench
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for kv in (torch.float8_e4m3fn, torch.float16):
    r = [bench.config4(flash_attn, dev, kv) for _ in range(n)]
    print(str(kv).split(".")[-1], " ".join(f"{x['achieved_gbs']:.0f}" for x in r), "GB/s", flush=True)
    torch.cuda.empty_cache()
