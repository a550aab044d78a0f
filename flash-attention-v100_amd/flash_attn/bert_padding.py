"""Padding helpers for the varlen path: (batch, seqlen, ...) <-> packed (total_tokens, ...).

Same public names, argument order and return tuples as the reference's
flash_attn/bert_padding.py:9-147 (callers of `flash_attn_varlen_func` use them), written from
the documented behaviour with plain row gather / scatter ops (index_select / index_copy)."""
import torch


def _flat_rows(x: torch.Tensor) -> torch.Tensor:
    """(batch, seqlen, ...) -> (batch * seqlen, ...) without copying when possible."""
    return x.reshape(x.shape[0] * x.shape[1], *x.shape[2:])


class IndexFirstAxis(torch.autograd.Function):
    """out[i] = input[indices[i]] along the first axis; backward scatters rows back."""

    @staticmethod
    def forward(ctx, input, indices):
        ctx.save_for_backward(indices)
        ctx.first_axis_dim = input.shape[0]
        return input.index_select(0, indices)

    @staticmethod
    def backward(ctx, grad_output):
        (indices,) = ctx.saved_tensors
        grad_input = grad_output.new_zeros((ctx.first_axis_dim,) + tuple(grad_output.shape[1:]))
        grad_input.index_copy_(0, indices, grad_output)
        return grad_input, None


index_first_axis = IndexFirstAxis.apply


class IndexPutFirstAxis(torch.autograd.Function):
    """out = zeros(first_axis_dim, ...); out[indices] = values."""

    @staticmethod
    def forward(ctx, values, indices, first_axis_dim):
        ctx.save_for_backward(indices)
        out = values.new_zeros((first_axis_dim,) + tuple(values.shape[1:]))
        out.index_copy_(0, indices, values)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (indices,) = ctx.saved_tensors
        return grad_output.index_select(0, indices), None, None


index_put_first_axis = IndexPutFirstAxis.apply


class IndexFirstAxisResidual(torch.autograd.Function):
    """Returns (input[indices], input.detach()); both gradients flow back into `input`."""

    @staticmethod
    def forward(ctx, input, indices):
        ctx.save_for_backward(indices)
        ctx.first_axis_dim = input.shape[0]
        return input.index_select(0, indices), input.detach()

    @staticmethod
    def backward(ctx, grad_output, grad_residual):
        (indices,) = ctx.saved_tensors
        grad_input = grad_residual.clone()
        grad_input.index_add_(0, indices, grad_output)
        return grad_input, None


index_first_axis_residual = IndexFirstAxisResidual.apply


def _cu_seqlens(seqlens: torch.Tensor) -> torch.Tensor:
    out = torch.zeros(seqlens.numel() + 1, dtype=torch.int32, device=seqlens.device)
    out[1:] = torch.cumsum(seqlens, dim=0, dtype=torch.int32)
    return out


def unpad_input(hidden_states, attention_mask, unused_mask=None):
    """-> (packed hidden_states, indices, cu_seqlens int32, max_seqlen_in_batch, seqlens int32)."""
    mask = attention_mask if unused_mask is None else attention_mask + unused_mask
    seqlens = mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(mask.flatten(), as_tuple=False).flatten()
    return (index_first_axis(_flat_rows(hidden_states), indices), indices, _cu_seqlens(seqlens),
            int(seqlens.max().item()), seqlens)


def unpad_input_for_concatenated_sequences(hidden_states, attention_mask_in_length):
    """Rows hold the lengths of the samples concatenated in them (non-zero entries)."""
    row_len = attention_mask_in_length.sum(dim=-1)
    seqlen = attention_mask_in_length.shape[-1]
    pos = torch.arange(seqlen, device=row_len.device, dtype=row_len.dtype)
    token_mask = pos.unsqueeze(0) < row_len.unsqueeze(1)
    flat_len = attention_mask_in_length.flatten()
    seqlens = flat_len[torch.nonzero(flat_len, as_tuple=False).flatten()]
    indices = torch.nonzero(token_mask.flatten(), as_tuple=False).flatten()
    return (index_first_axis(_flat_rows(hidden_states), indices), indices, _cu_seqlens(seqlens),
            int(seqlens.max().item()))


def pad_input(hidden_states, indices, batch, seqlen):
    """(total_nnz, ...) -> (batch, seqlen, ...) with zeros at the padded positions."""
    out = index_put_first_axis(hidden_states, indices, batch * seqlen)
    return out.reshape(batch, seqlen, *hidden_states.shape[1:])
