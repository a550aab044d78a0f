"""GPU parity (bit-exact): the row gather / scatter kernels behind flash_attn.bert_padding vs the numpy oracle,
through the Python helpers and through the C ABI directly."""
import ctypes

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _bp():
    from flash_attn import bert_padding
    return bert_padding


def _bits(t):
    return t.detach().contiguous().view(torch.int16 if t.element_size() == 2 else torch.int32).cpu().numpy()


@pytest.mark.parametrize("B,S,H,D,dt", [(3, 37, 4, 64, torch.float16), (5, 128, 8, 128, torch.bfloat16),
                                        (2, 9, 1, 8, torch.float16), (4, 33, 2, 40, torch.float32)])
def test_unpad_pad_roundtrip_bit_exact(B, S, H, D, dt):
    torch.manual_seed(421)
    x = torch.randn(B, S, H, D, device="cuda", dtype=dt)
    lens = torch.randint(0, S + 1, (B,))
    lens[0] = S
    mask = (torch.arange(S)[None, :] < lens[:, None]).cuda()
    packed, idx, cu, mx, seqlens = _bp().unpad_input(x, mask)
    p_ref, i_ref, cu_ref, mx_ref, sl_ref = oracle.unpad_input(_bits(x).reshape(B, S, H, -1), mask.cpu().numpy())
    assert np.array_equal(idx.cpu().numpy(), i_ref) and np.array_equal(cu.cpu().numpy(), cu_ref)
    assert mx == mx_ref and np.array_equal(seqlens.cpu().numpy(), sl_ref)
    assert np.array_equal(_bits(packed).reshape(p_ref.shape), p_ref)
    back = _bp().pad_input(packed, idx, B, S)
    assert back.shape == x.shape
    b_ref = oracle.pad_input(p_ref, i_ref, B, S)
    assert np.array_equal(_bits(back).reshape(b_ref.shape), b_ref)
    # the padded positions are exactly zero, the valid ones exactly the input
    assert torch.equal(back, torch.where(mask[:, :, None, None], x, torch.zeros_like(x)))


def test_index_ops_autograd_and_unsorted_indices():
    torch.manual_seed(7)
    bp = _bp()
    x = torch.randn(50, 4, 32, device="cuda", dtype=torch.float16, requires_grad=True)
    idx = torch.randperm(50, device="cuda")[:31]                 # unsorted, unique: the memset + scatter path
    y = bp.index_first_axis(x, idx)
    assert torch.equal(y, x[idx])
    g = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, x, g)
    ref = torch.zeros_like(x)
    ref[idx] = g
    assert torch.equal(gx, ref)
    v = torch.randn(31, 4, 32, device="cuda", dtype=torch.float16, requires_grad=True)
    out = bp.index_put_first_axis(v, idx, 50)
    ref2 = oracle.scatter_rows(_bits(v).reshape(31, 4, 32), idx.cpu().numpy(), 50)
    assert np.array_equal(_bits(out).reshape(ref2.shape), ref2)
    (gv,) = torch.autograd.grad(out, v, torch.ones_like(out))
    assert torch.equal(gv, torch.ones_like(v))
    # rows that are not 16-byte multiples fall back to torch indexing with the same results
    z = torch.randn(20, 3, device="cuda", dtype=torch.float16)
    assert torch.equal(bp.index_first_axis(z, idx[:5] % 20), z[idx[:5] % 20])


def test_rows_c_abi_direct_and_errors():
    from flash_attn_mi355 import _lib
    torch.manual_seed(3)
    src = torch.randn(1000, 256, device="cuda", dtype=torch.float16)          # 512-byte rows
    view = src[:, :128]                                                       # row-strided source: 256 of 512 bytes
    idx = torch.randint(-1000, 1000, (4096,), device="cuda", dtype=torch.int64)
    out = torch.empty(4096, 128, device="cuda", dtype=torch.float16)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call_rows("fa_gather_rows", view.data_ptr(), idx.data_ptr(), out.data_ptr(), 4096, 256, 512, 1000, st)
    ref = oracle.gather_rows(_bits(src)[:, :128], idx.cpu().numpy())
    assert np.array_equal(_bits(out), ref)
    sidx = torch.sort(torch.randperm(5000, device="cuda")[:1000]).values
    dst = torch.full((5000, 256), 7.0, device="cuda", dtype=torch.float16)
    for sorted_flag in (1, 0):
        dst.fill_(7.0)
        _lib.call_rows("fa_scatter_rows", src.data_ptr(), sidx.data_ptr(), dst.data_ptr(), 1000, 5000, 512, sorted_flag, st)
        assert np.array_equal(_bits(dst), oracle.scatter_rows(_bits(src), sidx.cpu().numpy(), 5000))
    with pytest.raises(RuntimeError, match="multiples of 16"):
        _lib.call_rows("fa_gather_rows", src.data_ptr(), idx.data_ptr(), out.data_ptr(), 4, 24, 24, 1000, st)
    with pytest.raises(RuntimeError, match="NULL"):
        _lib.call_rows("fa_scatter_rows", 0, sidx.data_ptr(), dst.data_ptr(), 10, 5000, 512, 1, st)
    # empty requests are no-ops
    _lib.call_rows("fa_gather_rows", 0, 0, 0, 0, 512, 512, 0, st)
    _lib.call_rows("fa_scatter_rows", 0, 0, dst.data_ptr(), 0, 0, 512, 1, st)
