import os, sys, torch
sys.path.insert(0, "/root/repo/flash-attention-v100_amd"); sys.path.insert(0, "/root/repo")
import flash_attn
def run(B,S,H,Hk,D=128,causal=True,dt=torch.bfloat16,Sk=None,window=(-1,-1)):
    Sk = Sk or S
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn(B,S,H,D,device="cuda",dtype=dt,generator=g).requires_grad_(True)
    k = torch.randn(B,Sk,Hk,D,device="cuda",dtype=dt,generator=g).requires_grad_(True)
    v = torch.randn(B,Sk,Hk,D,device="cuda",dtype=dt,generator=g).requires_grad_(True)
    do = torch.randn(B,S,H,D,device="cuda",dtype=dt,generator=g)
    o = flash_attn.flash_attn_func(q,k,v,causal=causal,window_size=window)
    f = lambda: torch.autograd.grad(o,(q,k,v),do,retain_graph=True)
    gr = f()
    for _ in range(3): f()
    torch.cuda.synchronize()
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    return gr, s.elapsed_time(e)/20
mode = os.environ.get("FA_BWD_DS","default")
shapes = [(8,4096,16,16,128,True),(2,2048,8,8,128,True),(1,1024,4,4,128,False),(2,1500,8,2,128,True),(1,2048,4,4,128,True,torch.float16,3000),(1,3000,4,4,128,True,torch.float16,2048),(2,2048,4,4,128,False,torch.bfloat16,None,(300,0))]
out = {}
for sh in shapes:
    gr, ms = run(*sh)
    out[str(sh)] = [x.float().cpu() for x in gr]
    print(mode, sh, f"{ms:.4f} ms", flush=True)
torch.save(out, f"/tmp/ds_{mode}.pt")
if os.path.exists("/tmp/ds_0.pt") and os.path.exists("/tmp/ds_1.pt"):
    a, b = torch.load("/tmp/ds_0.pt"), torch.load("/tmp/ds_1.pt")
    for kk in a:
        for nm, x, y in zip(("dq","dk","dv"), a[kk], b[kk]):
            d = (x-y).abs().max().item(); r = x.abs().max().item()
            print(kk, nm, f"max diff {d:.3e} of {r:.3e}", "NaN!" if not torch.isfinite(y).all() else "")
