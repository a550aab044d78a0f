"""CPU restatement of the reference's padding helpers (test infrastructure only - never imported by the product).

Follows flash_attn/bert_padding.py of the reference:
  index_first_axis      :9-34    out = input[indices]            (torch.gather over an expanded index)
  index_put_first_axis  :36-60   out = zeros(first_axis_dim, ...); out[indices] = values
  unpad_input           :79-104  mask -> (rows of the valid tokens, indices, cu_seqlens, max_seqlen, seqlens)
  pad_input             :135-146 inverse of unpad_input with zeros at the padded positions
Plain numpy fancy indexing; byte-exact by construction, so the GPU row kernels are compared bit for bit."""
import numpy as np


def gather_rows(src: np.ndarray, indices: np.ndarray) -> np.ndarray:
    return src[np.asarray(indices, dtype=np.int64)]


def scatter_rows(values: np.ndarray, indices: np.ndarray, n_rows: int) -> np.ndarray:
    out = np.zeros((n_rows,) + values.shape[1:], dtype=values.dtype)
    out[np.asarray(indices, dtype=np.int64)] = values
    return out


def unpad_input(hidden: np.ndarray, mask: np.ndarray):
    seqlens = mask.sum(axis=-1).astype(np.int32)
    indices = np.flatnonzero(mask.reshape(-1)).astype(np.int64)
    cu = np.concatenate([[0], np.cumsum(seqlens)]).astype(np.int32)
    flat = hidden.reshape((hidden.shape[0] * hidden.shape[1],) + hidden.shape[2:])
    return gather_rows(flat, indices), indices, cu, int(seqlens.max()), seqlens


def pad_input(packed: np.ndarray, indices: np.ndarray, batch: int, seqlen: int) -> np.ndarray:
    return scatter_rows(packed, indices, batch * seqlen).reshape((batch, seqlen) + packed.shape[1:])
