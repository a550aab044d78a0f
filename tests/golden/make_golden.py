#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the reference implementation.

Runs ONLY in the build container (needs /root/reference).  It
  1. extracts the reference's own test oracle functions `ref_mha_forward` /
     `ref_mha_backward` from /root/reference/test.py (test.py:18-62) by AST - the module
     cannot be imported because it exit(0)s without the CUDA extension (test.py:12-16) -
     and runs them on CPU with the reference's protocol: torch.manual_seed(421), randn
     fp16 inputs, scale = 1/sqrt(D) (test.py:151-159);
  2. imports the reference's Python operator layer with a recording stub in place of the
     CUDA extension and stores the positional argument tuples it builds at the extension
     boundary (flash_attn_v100/flash_attn_interface.py:60,99,210,256,366).
Only data (inputs / expected outputs / call shapes) is written - no reference source.
"""
import ast
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_ref_oracle():
    src = open(os.path.join(REF, "test.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef)
            and n.name in ("ref_mha_forward", "ref_mha_backward")]
    mod = ast.Module(body=keep, type_ignores=[])
    ns = {"torch": torch}
    exec(compile(mod, "ref_test_oracle", "exec"), ns)
    return ns["ref_mha_forward"], ns["ref_mha_backward"]


def gen_dense(ref_fwd, ref_bwd):
    # the five 1-tile shapes of test.py:116-120 (D=M=N = 16 .. 256), x causal in {F,T},
    # plus two ragged-size cases exercising the same oracle functions.
    shapes = [(1, 1, 16, 16, 16), (1, 1, 32, 32, 32), (1, 1, 64, 64, 64),
              (1, 1, 128, 128, 128), (1, 1, 256, 256, 256), (1, 2, 96, 96, 64), (2, 1, 80, 80, 128)]
    for (B, H, M, N, D) in shapes:
        for causal in (False, True):
            torch.manual_seed(421)
            q = torch.randn(B, H, M, D, dtype=torch.float16)
            k = torch.randn(B, H, N, D, dtype=torch.float16)
            v = torch.randn(B, H, N, D, dtype=torch.float16)
            do = torch.randn(B, H, M, D, dtype=torch.float16)
            scale = 1.0 / (D ** 0.5)
            o = ref_fwd(q.float(), k.float(), v.float(), scale=scale, causal=causal)
            dq, dk, dv = ref_bwd(q.float(), k.float(), v.float(), do.float(), scale=scale,
                                 causal=causal)
            name = f"dense_B{B}H{H}M{M}N{N}D{D}_c{int(causal)}.npz"
            np.savez(os.path.join(OUT, name), q=q.numpy(), k=k.numpy(), v=v.numpy(),
                     do=do.numpy(), o=o.numpy(), dq=dq.numpy(), dk=dk.numpy(), dv=dv.numpy(),
                     scale=np.float64(scale), causal=np.bool_(causal))
            print("wrote", name)
    # BASELINE.json configs[0]: fp32 non-causal B2 H4 S128 D64 (inputs fp16-representable)
    torch.manual_seed(421)
    B, H, S, D = 2, 4, 128, 64
    q = torch.randn(B, H, S, D, dtype=torch.float16)
    k = torch.randn(B, H, S, D, dtype=torch.float16)
    v = torch.randn(B, H, S, D, dtype=torch.float16)
    o = ref_fwd(q.float(), k.float(), v.float(), scale=D ** -0.5, causal=False)
    np.savez(os.path.join(OUT, "config1_B2H4S128D64.npz"), q=q.numpy(), k=k.numpy(),
             v=v.numpy(), o=o.numpy(), scale=np.float64(D ** -0.5), causal=np.bool_(False))
    print("wrote config1")


def _desc(x):
    if isinstance(x, torch.Tensor):
        return {"tensor": list(x.shape), "dtype": str(x.dtype).replace("torch.", ""),
                "contiguous": bool(x.is_contiguous())}
    if isinstance(x, (tuple, list)):
        return [_desc(y) for y in x]
    if isinstance(x, float):
        return float(x)
    return x


def gen_boundary_calls():
    calls = {}
    stub = types.ModuleType("flash_attn_v100_cuda")

    def rec(name, n_out):
        def f(*args):
            calls.setdefault(name, []).append([_desc(a) for a in args])
            q = args[0]
            if name == "fwd":
                B, H, M, D = q.shape
                return (torch.zeros_like(q), torch.zeros(B, H, M), torch.zeros(1),
                        torch.zeros(2, dtype=torch.int64))
            if name == "varlen_fwd":
                T, H, D = q.shape
                return (torch.zeros_like(q), torch.zeros(H, T), torch.zeros(1),
                        torch.zeros(2, dtype=torch.int64))
            if name == "fwd_kvcache":
                B, T, H, D = q.shape
                return torch.zeros_like(q), torch.zeros(B, H, T)
            if name in ("bwd", "varlen_bwd"):
                return [torch.zeros_like(args[1]), torch.zeros_like(args[2]),
                        torch.zeros_like(args[3]), torch.zeros(1)]
        return f

    for n in ("fwd", "bwd", "varlen_fwd", "varlen_bwd", "fwd_kvcache"):
        setattr(stub, n, rec(n, 0))
    sys.modules["flash_attn_v100_cuda"] = stub
    sys.path.insert(0, REF)
    import flash_attn  # noqa: E402

    meta = {"flash_attn.__version__": flash_attn.__version__,
            "flash_attn.__all__": list(flash_attn.__all__),
            "flash_attn.__doc__": flash_attn.__doc__}
    torch.manual_seed(421)
    # config 1 shape through flash_attn_func (B,S,H,D layout at the Python API)
    q = torch.randn(2, 128, 4, 64, requires_grad=True)
    k = torch.randn(2, 128, 4, 64, requires_grad=True)
    v = torch.randn(2, 128, 4, 64, requires_grad=True)
    o = flash_attn.flash_attn_func(q, k, v)
    o.sum().backward()
    # head dim not a multiple of 8 -> padded (flash_attn_interface.py:44-49)
    q2 = torch.randn(1, 16, 2, 20)
    flash_attn.flash_attn_func(q2, q2, q2, causal=True, window_size=(7, 0), softcap=0.0,
                               alibi_slopes=torch.ones(2))
    cu = torch.tensor([0, 5, 12], dtype=torch.int64)
    qv = torch.randn(12, 4, 32, requires_grad=True)
    kv = torch.randn(12, 2, 32, requires_grad=True)
    ov = flash_attn.flash_attn_varlen_func(qv, kv, kv, cu, cu, 7, 7, causal=True)
    ov.sum().backward()
    qc = torch.randn(3, 1, 8, 64)
    kc = torch.randn(3, 256, 2, 64)
    flash_attn.flash_attn_with_kvcache(qc, kc, kc.clone(), k=torch.randn(3, 1, 2, 64),
                                       v=torch.randn(3, 1, 2, 64), cache_seqlens=17,
                                       rotary_cos=torch.randn(512, 16),
                                       rotary_sin=torch.randn(512, 16), causal=True)
    with open(os.path.join(OUT, "boundary_calls.json"), "w") as f:
        json.dump({"meta": meta, "calls": calls}, f, indent=1, default=str)
    print("wrote boundary_calls.json")


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference checkout not present - run in the build container"
    fwd, bwd = load_ref_oracle()
    gen_dense(fwd, bwd)
    gen_boundary_calls()
