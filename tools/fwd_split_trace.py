import os, sys, warnings, torch
sys.path.insert(0, "/root/repo/flash-attention-v100_amd")
import flash_attn
from flash_attn_mi355 import flash_attn_interface as fi
warnings.simplefilter("ignore")
B,S,H=1,2048,32
q,k,v=(torch.randn(B,S,H,128,device="cuda",dtype=torch.bfloat16) for _ in range(3))
fi.FWD_SPLIT = len(sys.argv) <= 1
with torch.no_grad():
    for _ in range(60): flash_attn.flash_attn_func(q,k,v,causal=True)
torch.cuda.synchronize()
