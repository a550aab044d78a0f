"""Which backward kernels a call launches follows from the gradients the op has to produce (autograd's
needs_input_grad, fixed at FORWARD time by which inputs require grad -> fa_bwd with dq == NULL or dk == dv == NULL):
"dq" = the dQ kernel (with the fused row-dot prologue), "dkdv" = the preprocess kernel + the dK/dV kernel, "all" = dQ
kernel + dK/dV kernel.  `fwd(q, k, v)` must build the graph."""
import torch


def bwd_call(fwd, q, k, v, do, which):
    qq, kk, vv = {"dq": (q, k.detach(), v.detach()), "dkdv": (q.detach(), k, v), "all": (q, k, v)}[which]
    o = fwd(qq, kk, vv)
    ins = {"dq": (q,), "dkdv": (k, v), "all": (q, k, v)}[which]
    return lambda: torch.autograd.grad(o, ins, do, retain_graph=True)
