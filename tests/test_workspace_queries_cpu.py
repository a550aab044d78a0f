"""CPU: the workspace queries of the C ABI are pure host logic - check the sizes the kernels rely on (no GPU needed).

  * backward: the dense / packed D = 128 paths without bias or dropout ask for the row-statistics planes of the
    hand-scheduled dK/dV kernel (2 x rows x 4 bytes); everything else asks for nothing;
  * kv-cache: split-KV partials = partial rows x (D + 1) x 4 bytes, where the token-major decode kernel multiplies the
    grid splits by its key sub-ranges when there are fewer head groups than waves."""
import ctypes

import pytest


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from flash_attn_mi355 import _lib
    return _lib


def _dense(lib, B, S, H, Hk, D, dtype=None):
    p = lib.FaParams()
    p.dtype = p.kv_dtype = lib.FA_BF16 if dtype is None else dtype
    p.batch, p.nheads_q, p.nheads_k, p.seqlen_q, p.seqlen_k, p.head_dim = B, H, Hk, S, S, D
    for n, h in (("q", H), ("o", H), ("do", H), ("dq", H), ("k", Hk), ("v", Hk), ("dk", Hk), ("dv", Hk)):
        setattr(p, n + "_batch_stride", S * h * D); setattr(p, n + "_row_stride", h * D); setattr(p, n + "_head_stride", D)
    p.is_causal = 1
    p.window_left = p.window_right = -1
    p.softmax_scale = D ** -0.5
    return p


def test_backward_statistics_workspace(lib):
    q = lib.lib.fa_bwd_workspace_bytes
    p = _dense(lib, 8, 4096, 16, 16, 128)
    assert q(ctypes.byref(p)) == 2 * 8 * 16 * 4096 * 4
    p = _dense(lib, 2, 777, 8, 2, 128)                       # GQA: statistics per q-head
    assert q(ctypes.byref(p)) == 2 * 2 * 8 * 777 * 4
    p = _dense(lib, 2, 512, 8, 8, 64)                        # other head dims: compiler kernels, no workspace
    assert q(ctypes.byref(p)) == 0
    p = _dense(lib, 2, 512, 8, 8, 128)
    p.p_dropout = 0.1
    assert q(ctypes.byref(p)) == 0
    p = _dense(lib, 2, 512, 8, 8, 128)
    p.softcap = 30.0
    assert q(ctypes.byref(p)) == 0
    # packed sequences: [H][total_q] planes
    p = _dense(lib, 3, 2048, 8, 4, 128)
    cu = (ctypes.c_int32 * 4)(0, 1000, 1300, 3348)
    p.cu_seqlens_q = p.cu_seqlens_k = ctypes.addressof(cu)   # (host memory: the query only tests the pointers for NULL)
    p.total_q = p.total_k = 3348
    assert q(ctypes.byref(p)) == 2 * 8 * 3348 * 4


def _decode(lib, B, H, Hk, L, kv8, num_splits=0):
    p = lib.FaParams()
    p.dtype = lib.FA_FP16
    p.kv_dtype = lib.FA_FP8_E4M3 if kv8 else lib.FA_FP16
    p.batch, p.nheads_q, p.nheads_k, p.seqlen_q, p.seqlen_k, p.head_dim = B, H, Hk, 1, L, 128
    p.q_batch_stride, p.q_row_stride, p.q_head_stride = H * 128, H * 128, 128
    p.o_batch_stride, p.o_row_stride, p.o_head_stride = H * 128, H * 128, 128
    p.k_batch_stride = p.v_batch_stride = L * Hk * 128
    p.k_row_stride = p.v_row_stride = Hk * 128
    p.k_head_stride = p.v_head_stride = 128
    p.window_left = p.window_right = -1
    p.is_causal = 1
    p.softmax_scale = 128 ** -0.5
    p.k_descale = p.v_descale = 1.0
    p.num_splits = num_splits
    return p


def test_decode_partials_workspace(lib):
    q = lib.lib.fa_fwd_kvcache_workspace_bytes
    per_part = lambda B, H: B * H * (128 + 1) * 4
    # config 4 (B128, 32 heads, fp8): token-major kernel, one workgroup per CU -> 2 splits on a 256-CU part
    n = q(ctypes.byref(_decode(lib, 128, 32, 32, 8192, True)))
    assert n % per_part(128, 32) == 0 and 1 <= n // per_part(128, 32) <= 8
    # explicit splits are honoured; fewer head groups than waves multiply the partial rows (8 kv-heads, fp8: 1 group -> x4)
    assert q(ctypes.byref(_decode(lib, 16, 32, 32, 4096, True, num_splits=3))) == 3 * per_part(16, 32)
    assert q(ctypes.byref(_decode(lib, 16, 16, 8, 4096, True, num_splits=3))) == 3 * 4 * per_part(16, 16)    # G = 2, fp8
    # fp8 caches with groups of 4 run the MFMA decode kernel (round 3: two waves per SIMD beat the VALU-bound form): splits only
    assert q(ctypes.byref(_decode(lib, 16, 32, 8, 4096, True, num_splits=3))) == 3 * per_part(16, 32)
    assert q(ctypes.byref(_decode(lib, 16, 16, 8, 4096, False, num_splits=2))) == 2 * 2 * per_part(16, 16)   # 16 bit: 4 heads per wave step
    assert q(ctypes.byref(_decode(lib, 16, 32, 32, 4096, False, num_splits=1))) == 0                         # one partial = written in place
    # four kv-heads with fp8: the head-major kernel, splits only
    assert q(ctypes.byref(_decode(lib, 16, 4, 4, 4096, True, num_splits=5))) == 5 * per_part(16, 4)
