"""Quick numerical check + timing of the generated dQ kernel against a torch fp32 reference (development aid).
FA_BWD_DQ_ASM=0 selects the compiler kernel for comparison."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
torch.manual_seed(421)


def ref(q, k, v, do, causal, window):
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    qf, kf, vf, dof = (t.float().permute(0, 2, 1, 3) for t in (q, k, v, do))
    kf = kf.repeat_interleave(H // Hk, 1); vf = vf.repeat_interleave(H // Hk, 1)
    qf.requires_grad_(True)
    s = qf @ kf.transpose(-1, -2) * D ** -0.5
    i = torch.arange(Sq, device=q.device)[:, None]; j = torch.arange(Sk, device=q.device)[None, :] - (Sk - Sq)
    m = torch.zeros(Sq, Sk, dtype=torch.bool, device=q.device)
    if causal: m |= j > i
    if window[0] >= 0: m |= j < i - window[0]
    if window[1] >= 0: m |= j > i + window[1]
    s = s.masked_fill(m, float("-inf"))
    p = torch.softmax(s, -1).nan_to_num(0.0)
    o = p @ vf
    (dq,) = torch.autograd.grad(o, qf, dof)
    return dq.permute(0, 2, 1, 3)


cases = [(1, 256, 256, 2, 2, True, (-1, -1), torch.bfloat16), (2, 1024, 1024, 4, 2, True, (-1, -1), torch.bfloat16),
         (1, 1024, 1024, 2, 2, False, (-1, -1), torch.float16), (1, 333, 777, 2, 1, True, (-1, -1), torch.bfloat16),
         (1, 777, 333, 2, 2, True, (-1, -1), torch.float16), (1, 1024, 1500, 2, 2, False, (100, 50), torch.bfloat16),
         (1, 2000, 2000, 2, 2, False, (300, 0), torch.float16), (2, 4096, 4096, 2, 2, True, (-1, -1), torch.bfloat16)]
for (B, Sq, Sk, H, Hk, causal, window, dt) in cases:
    q = torch.randn(B, Sq, H, 128, device="cuda", dtype=dt, requires_grad=True)
    k = torch.randn(B, Sk, Hk, 128, device="cuda", dtype=dt); v = torch.randn(B, Sk, Hk, 128, device="cuda", dtype=dt)
    do = torch.randn(B, Sq, H, 128, device="cuda", dtype=dt)
    o = flash_attn.flash_attn_func(q, k, v, causal=causal, window_size=window)
    (dq,) = torch.autograd.grad(o, (q,), do)
    torch.cuda.synchronize()
    r = ref(q.detach(), k, v, do, causal, window)
    err = (dq.float() - r).abs().max().item() / r.abs().max().item()
    print(f"B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} causal={causal} window={window} {dt}: dq max-rel {err:.3e} finite={torch.isfinite(dq.float()).all().item()}", flush=True)
if "--time" in sys.argv:
    B, S, H = 8, 4096, 16
    q = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k, v, do = (torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    for causal in (True, False):
        o = flash_attn.flash_attn_func(q, k, v, causal=causal)
        for _ in range(3):
            torch.autograd.grad(o, (q,), do, retain_graph=True)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for s, e in evs:
            s.record(); torch.autograd.grad(o, (q,), do, retain_graph=True); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in evs)
        fl = 2.0 * B * H * S * S * 128 * (0.5 if causal else 1.0)
        print(f"dQ kernel causal={causal}: median {ts[10]:.4f} ms  min {ts[0]:.4f}  ({3 * fl / ts[10] / 1e9:.0f} TFLOP/s executed, {fl / ts[10] / 1e9:.0f} algorithmic)", flush=True)
