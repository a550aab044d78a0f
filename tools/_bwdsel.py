"""Which backward kernels a call launches follows from the gradients asked for (autograd's needs_input_grad ->
fa_bwd with dq == NULL or dk == dv == NULL): "dq" = the dQ kernel (with the fused row-dot prologue), "dkdv" = the
preprocess kernel + the dK/dV kernel, "all" = dQ kernel + dK/dV kernel."""
import torch


def bwd_call(o, q, k, v, do, which):
    ins = {"dq": (q,), "dkdv": (k, v), "all": (q, k, v)}[which]
    return lambda: torch.autograd.grad(o, ins, do, retain_graph=True)
