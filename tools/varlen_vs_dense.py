import os, sys
sys.path.insert(0, "/root/repo/flash-attention-v100_amd")
import torch, flash_attn
def t(fn, n=10):
    for _ in range(3): fn()
    evs=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for s,e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s,e in evs)[n//2]
H=16
for lens in ([4096]*8, [2048]*16, [1024]*32, [512]*64, [4096,3584,3072,2560,2048,1536,1024,512]*2, [3000]*10, [700]*40):
    B=len(lens); T=sum(lens)
    cu=torch.tensor([0]+list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    q=torch.randn(T,H,128,device="cuda",dtype=torch.bfloat16); k=torch.randn_like(q); v=torch.randn_like(q)
    fl=4.0*128*H*sum(L*(L+1)//2 for L in lens)
    with torch.no_grad():
        tv=t(lambda: flash_attn.flash_attn_varlen_func(q,k,v,cu,cu,max(lens),max(lens),causal=True))
    line=f"lens {lens[:3]}..x{B} ({T} tok): varlen fwd {tv:.3f} ms {fl/tv/1e9:.0f} TF"
    if len(set(lens))==1:
        L=lens[0]; qd=q.view(B,L,H,128); 
        with torch.no_grad():
            td=t(lambda: flash_attn.flash_attn_func(qd,qd,qd,causal=True))
        line+=f" | dense fwd {td:.3f} ms {fl/td/1e9:.0f} TF"
    print(line, flush=True)
