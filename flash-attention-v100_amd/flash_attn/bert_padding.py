"""Padding helpers for the varlen path: (batch, seqlen, ...) <-> packed (total_tokens, ...).

Same public names, argument order and return tuples as the reference's
flash_attn/bert_padding.py:9-147 (callers of `flash_attn_varlen_func` use them), written from
the documented behaviour.  On the GPU the row moves go through the library's HBM-bound row kernels
(`fa_gather_rows` / `fa_scatter_rows`, include/fa_mi355.h; csrc/fa_rows.hip); CPU tensors and rows that are not
16-byte multiples use index_select / index_copy."""
import torch


def _flat_rows(x: torch.Tensor) -> torch.Tensor:
    """(batch, seqlen, ...) -> (batch * seqlen, ...) without copying when possible."""
    return x.reshape(x.shape[0] * x.shape[1], *x.shape[2:])


def _row_bytes(x: torch.Tensor) -> int:
    return int(torch.Size(x.shape[1:]).numel()) * x.element_size()


def _row_kernel_ok(x: torch.Tensor, indices: torch.Tensor) -> bool:
    if not (x.is_cuda and indices.is_cuda and indices.dtype == torch.int64 and indices.dim() == 1 and x.dim() >= 1):
        return False
    rb = _row_bytes(x)
    return rb > 0 and rb % 16 == 0


def _gather_rows(x: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
    """x[indices] along the first axis."""
    if not _row_kernel_ok(x, indices):
        return x.index_select(0, indices)
    from flash_attn_mi355 import _lib
    inner = x[0].is_contiguous() if x.shape[0] else True
    if not inner or (x.shape[0] > 1 and (x.stride(0) * x.element_size()) % 16):
        x = x.contiguous()
    indices = indices.contiguous()
    out = torch.empty((indices.numel(),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    row_bytes = _row_bytes(x)
    stride_b = x.stride(0) * x.element_size() if x.shape[0] > 1 else row_bytes
    if x.data_ptr() % 16 or out.data_ptr() % 16 or stride_b < row_bytes:
        return x.index_select(0, indices)
    with torch.cuda.device(x.device):
        _lib.call_rows("fa_gather_rows", x.data_ptr(), indices.data_ptr(), out.data_ptr(), indices.numel(), row_bytes,
                       stride_b, x.shape[0], torch.cuda.current_stream().cuda_stream)
    return out


def _scatter_rows(values: torch.Tensor, indices: torch.Tensor, n_rows: int, sorted_unique: bool) -> torch.Tensor:
    """zeros(n_rows, ...) with out[indices] = values."""
    if not _row_kernel_ok(values, indices):
        out = values.new_zeros((n_rows,) + tuple(values.shape[1:]))
        out.index_copy_(0, indices, values)
        return out
    from flash_attn_mi355 import _lib
    values = values.contiguous()
    indices = indices.contiguous()
    out = torch.empty((n_rows,) + tuple(values.shape[1:]), dtype=values.dtype, device=values.device)
    row_bytes = _row_bytes(values)
    if values.data_ptr() % 16 or out.data_ptr() % 16:
        out.zero_()
        out.index_copy_(0, indices, values)
        return out
    with torch.cuda.device(values.device):
        _lib.call_rows("fa_scatter_rows", values.data_ptr(), indices.data_ptr(), out.data_ptr(), indices.numel(), n_rows,
                       row_bytes, 1 if sorted_unique else 0, torch.cuda.current_stream().cuda_stream)
    return out


def _is_sorted_unique(indices: torch.Tensor, sorted_unique=None) -> bool:
    """Ascending indices without repeats let pad_input take the one-pass scatter.  The caller can say so explicitly
    (`sorted_unique=True/False` on pad_input / index_put_first_axis / index_first_axis); with None the marker that
    unpad_input leaves on the index tensor it returns is consulted - a copy of that tensor (`.to()`, `.clone()`) has no
    marker and takes the general two-pass path, which is always correct."""
    if sorted_unique is not None:
        return bool(sorted_unique)
    return bool(getattr(indices, "_fa_sorted_unique", False))


class IndexFirstAxis(torch.autograd.Function):
    """out[i] = input[indices[i]] along the first axis; backward scatters rows back."""

    @staticmethod
    def forward(ctx, input, indices, sorted_unique=None):
        ctx.save_for_backward(indices)
        ctx.first_axis_dim = input.shape[0]
        ctx.sorted_unique = _is_sorted_unique(indices, sorted_unique)
        return _gather_rows(input, indices)

    @staticmethod
    def backward(ctx, grad_output):
        (indices,) = ctx.saved_tensors
        return _scatter_rows(grad_output, indices, ctx.first_axis_dim, ctx.sorted_unique), None, None


def index_first_axis(input, indices, sorted_unique=None):
    return IndexFirstAxis.apply(input, indices, sorted_unique)


class IndexPutFirstAxis(torch.autograd.Function):
    """out = zeros(first_axis_dim, ...); out[indices] = values."""

    @staticmethod
    def forward(ctx, values, indices, first_axis_dim, sorted_unique=None):
        ctx.save_for_backward(indices)
        return _scatter_rows(values, indices, first_axis_dim, _is_sorted_unique(indices, sorted_unique))

    @staticmethod
    def backward(ctx, grad_output):
        (indices,) = ctx.saved_tensors
        return _gather_rows(grad_output, indices), None, None, None


def index_put_first_axis(values, indices, first_axis_dim, sorted_unique=None):
    return IndexPutFirstAxis.apply(values, indices, first_axis_dim, sorted_unique)


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
    indices._fa_sorted_unique = True
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
    indices._fa_sorted_unique = True
    return (index_first_axis(_flat_rows(hidden_states), indices), indices, _cu_seqlens(seqlens),
            int(seqlens.max().item()))


def pad_input(hidden_states, indices, batch, seqlen, *, sorted_unique=None):
    """(total_nnz, ...) -> (batch, seqlen, ...) with zeros at the padded positions.  sorted_unique=True promises
    ascending indices without repeats (what unpad_input returns) and selects the one-pass kernel."""
    out = index_put_first_axis(hidden_states, indices, batch * seqlen, sorted_unique)
    return out.reshape(batch, seqlen, *hidden_states.shape[1:])
