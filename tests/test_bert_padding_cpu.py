"""CPU: flash_attn.bert_padding helpers (host glue around the varlen path)."""
import torch

from flash_attn.bert_padding import (index_first_axis, index_first_axis_residual, index_put_first_axis,
                                     pad_input, unpad_input, unpad_input_for_concatenated_sequences)


def test_unpad_pad_roundtrip_and_grads():
    torch.manual_seed(0)
    B, S, H = 3, 7, 5
    x = torch.randn(B, S, H, dtype=torch.float64, requires_grad=True)
    lens = torch.tensor([7, 2, 4])
    mask = (torch.arange(S)[None] < lens[:, None]).int()
    xu, idx, cu, mx, sl = unpad_input(x, mask)
    assert xu.shape == (13, H) and cu.tolist() == [0, 7, 9, 13] and mx == 7 and sl.tolist() == [7, 2, 4]
    assert cu.dtype == torch.int32 and sl.dtype == torch.int32
    back = pad_input(xu, idx, B, S)
    assert torch.equal(back, x * mask[..., None])
    (back * torch.arange(B * S * H, dtype=torch.float64).reshape(B, S, H)).sum().backward()
    expect = torch.arange(B * S * H, dtype=torch.float64).reshape(B, S, H) * mask[..., None]
    assert torch.equal(x.grad, expect)


def test_unused_mask_and_concatenated():
    x = torch.arange(2 * 6 * 1, dtype=torch.float32).reshape(2, 6, 1)
    am = torch.tensor([[1, 1, 0, 0, 0, 0], [1, 1, 1, 0, 0, 0]])
    um = torch.tensor([[0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0]])
    xu, idx, cu, mx, sl = unpad_input(x, am, um)
    assert idx.tolist() == [0, 1, 2, 6, 7, 8] and cu.tolist() == [0, 3, 6] and mx == 3
    lens = torch.tensor([[2, 3, 0, 0, 0, 0], [6, 0, 0, 0, 0, 0]])
    xu, idx, cu, mx = unpad_input_for_concatenated_sequences(x, lens)
    assert idx.tolist() == [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11] and cu.tolist() == [0, 2, 5, 11] and mx == 6


def test_index_ops_autograd():
    x = torch.randn(6, 3, dtype=torch.float64, requires_grad=True)
    idx = torch.tensor([4, 0, 2])
    assert torch.autograd.gradcheck(lambda t: index_first_axis(t, idx), (x,))
    v = torch.randn(3, 3, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: index_put_first_axis(t, idx, 6), (v,))
    out, res = index_first_axis_residual(x, idx)
    (out.sum() + 2 * res.sum()).backward()
    g = torch.full((6, 3), 2.0, dtype=torch.float64); g[idx] += 1.0
    assert torch.equal(x.grad, g)
