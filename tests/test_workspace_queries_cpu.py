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
    p.flags = lib.FA_FLAG_NO_DKV_SPLIT
    assert q(ctypes.byref(p)) == 2 * 2 * 8 * 777 * 4
    p = _dense(lib, 2, 512, 8, 8, 64)                        # other head dims: compiler kernels, no workspace
    p.flags = lib.FA_FLAG_NO_DKV_SPLIT
    assert q(ctypes.byref(p)) == 0
    p = _dense(lib, 2, 512, 8, 8, 128)
    p.p_dropout = 0.1
    p.flags = lib.FA_FLAG_NO_DKV_SPLIT
    assert q(ctypes.byref(p)) == 0
    p = _dense(lib, 2, 512, 8, 8, 128)
    p.softcap = 30.0
    p.flags = lib.FA_FLAG_NO_DKV_SPLIT
    assert q(ctypes.byref(p)) == 0
    # packed sequences: [H][total_q] planes
    p = _dense(lib, 3, 2048, 8, 4, 128)
    cu = (ctypes.c_int32 * 4)(0, 1000, 1300, 3348)
    p.cu_seqlens_q = p.cu_seqlens_k = ctypes.addressof(cu)   # (host memory: the query only tests the pointers for NULL)
    p.total_q = p.total_k = 3348
    p.flags = lib.FA_FLAG_NO_DKV_SPLIT
    assert q(ctypes.byref(p)) == 2 * 8 * 3348 * 4
    # ... (26 + 3) key blocks x 4 kv-heads = 116 workgroups, 35 stages of the average causal pass: 2 splits, slabs of total_k rows
    p.flags = 0
    assert q(ctypes.byref(p)) == ((2 * 8 * 3348 * 4 + 255) & ~255) + 2 * 2 * 3348 * 4 * 128 * 4


def test_backward_split_workspace(lib):
    """dK/dV launches smaller than the chip (batch x kv-heads x 128-key blocks, causal: mirrored pairs, < 256 CUs x the
    kernel's workgroups per CU) add fp32 partial dK / dV slabs behind the statistics planes: 2 x splits x B x Sk x Hk x D
    floats (fa_bwd.hip: dkv_split_factor; without a GPU the CU count defaults to 256)."""
    q = lib.lib.fa_bwd_workspace_bytes
    al = lambda x: (x + 255) & ~255
    # Llama-3 layer at micro-batch 1: 8 kv-heads x 16 mirrored pairs = 128 workgroups of the hand-scheduled kernel -> 2 splits
    p = _dense(lib, 1, 4096, 32, 8, 128)
    assert q(ctypes.byref(p)) == al(2 * 32 * 4096 * 4) + 2 * 2 * 4096 * 8 * 128 * 4
    p.flags = lib.FA_FLAG_NO_DKV_SPLIT
    assert q(ctypes.byref(p)) == 2 * 32 * 4096 * 4
    # ... at micro-batch 2 the launch fills the chip
    p = _dense(lib, 2, 4096, 32, 8, 128)
    assert q(ctypes.byref(p)) == 2 * 2 * 32 * 4096 * 4
    # 16 workgroups, 25 stages x 4 q-heads: the most splits (8)
    p = _dense(lib, 2, 777, 8, 2, 128)
    assert q(ctypes.byref(p)) == al(2 * 2 * 8 * 777 * 4) + 2 * 8 * 2 * 777 * 2 * 128 * 4
    # head dim 64: three workgroups per CU, 64-row stages; 2 x 8 x 2 = 32 workgroups, 8 stages each: no split keeps 8 stages
    p = _dense(lib, 2, 512, 8, 8, 64)
    assert q(ctypes.byref(p)) == 0
    p = _dense(lib, 2, 2048, 8, 8, 64)                       # 2 x 8 x 8 pairs = 128 workgroups of 768 slots, 32 stages: 4 splits
    assert q(ctypes.byref(p)) == 2 * 4 * 2 * 2048 * 8 * 64 * 4
    # head dim 256 (two waves per key block, one workgroup per CU): 64 workgroups, 64 stages -> 4 splits
    p = _dense(lib, 1, 2048, 8, 8, 256)
    assert q(ctypes.byref(p)) == 2 * 4 * 1 * 2048 * 8 * 256 * 4
    # a kernel without a split form: dropout with a bias (fa_bwd_dkdv_kernel)
    p = _dense(lib, 1, 2048, 8, 8, 128)
    p.p_dropout = 0.1
    p.softcap = 30.0
    assert q(ctypes.byref(p)) == 0


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


def _varlen_paged(lib, B, total_q, max_q, H, Hk, D=128, page=256, max_k=8192, kv8=False):
    p = lib.FaParams()
    p.dtype = lib.FA_BF16
    p.kv_dtype = lib.FA_FP8_E4M3 if kv8 else lib.FA_BF16
    p.batch, p.nheads_q, p.nheads_k, p.seqlen_q, p.seqlen_k, p.head_dim = B, H, Hk, max_q, max_k, D
    p.total_q = total_q
    p.q_row_stride, p.q_head_stride = H * D, D
    p.o_row_stride, p.o_head_stride = H * D, D
    p.k_batch_stride = p.v_batch_stride = page * Hk * D
    p.k_row_stride = p.v_row_stride = Hk * D
    p.k_head_stride = p.v_head_stride = D
    p.lse_head_stride = total_q
    p.page_block_size = page
    p.block_table_batch_stride = max_k // page
    dummy = (ctypes.c_int32 * 8)()
    p.cu_seqlens_q = p.cu_seqlens_k = p.block_table = ctypes.addressof(dummy)     # (host logic only tests for NULL)
    p._keep = dummy
    p.is_causal = 1
    p.window_left = p.window_right = -1
    p.softmax_scale = D ** -0.5
    p.k_descale = p.v_descale = 1.0
    return p


def test_varlen_forward_workspace_marks_the_decode_routes(lib):
    """fa_fwd_workspace_bytes is how a caller learns that fa_varlen_fwd can hand a call (or its short sequences) to the decode
    kernels (fa_api.hip: varlen_decode_route, varlen_mixed_route) - pure host logic on sizes the host can see."""
    q = lib.lib.fa_fwd_workspace_bytes
    per_part = lambda rows: rows * (128 + 1) * 4
    # dense calls never ask
    assert q(ctypes.byref(_dense(lib, 8, 4096, 16, 16, 128))) == 0
    # uniform decode through the varlen op: batch 2, one token each, H 32/8 over 8 k -> split until the chip is full
    n = q(ctypes.byref(_varlen_paged(lib, 2, 2, 1, 32, 8)))
    assert n > 0 and n % per_part(2 * 32 * 1) == 0
    # uniform 4 speculative tokens
    n = q(ctypes.byref(_varlen_paged(lib, 2, 8, 4, 32, 8)))
    assert n > 0 and n % per_part(2 * 32 * 4) == 0
    # a big uniform decode batch: one partial, written in place
    assert q(ctypes.byref(_varlen_paged(lib, 512, 512, 1, 32, 8))) == 0
    # ragged prefill (8 prompts averaging 1000 rows): the general kernel alone
    assert q(ctypes.byref(_varlen_paged(lib, 8, 8000, 2048, 32, 8))) == 0
    # mixed: 12 decode sequences + one 300-token chunk -> partial rows for T = min(8, 32 / G) = 8 positions per sequence
    n = q(ctypes.byref(_varlen_paged(lib, 13, 312, 300, 32, 8)))
    assert n > 0 and n % per_part(13 * 32 * 8) == 0
    # G = 8 -> T = 4
    n = q(ctypes.byref(_varlen_paged(lib, 13, 312, 300, 64, 8)))
    assert n > 0 and n % per_part(13 * 64 * 4) == 0
    # too few sequences for the mixed route, and unpaged K / V, ask for nothing
    assert q(ctypes.byref(_varlen_paged(lib, 3, 302, 300, 32, 8))) == 0
    p = _varlen_paged(lib, 13, 312, 300, 32, 8)
    p.block_table = None
    assert q(ctypes.byref(p)) == 0


def test_kvcache_plan_key_is_geometry_only():
    """flash_attn_with_kvcache's plan key (flash_attn_interface._kv_plan_key): equal for tensors of equal geometry and different
    data, different when a dtype / shape / stride / optional / scalar option differs, None when the call needs the slow path
    (an int cache_seqlens, a non-contiguous last dimension)."""
    import torch
    from flash_attn_mi355 import flash_attn_interface as fi

    def mk(B=2, Hk=2, page=16, seed=0, dt=torch.float16):
        g = torch.Generator().manual_seed(seed)
        q = torch.randn(B, 1, 4, 64, generator=g).to(dt)
        kc = torch.randn(B * 3, page, Hk, 64, generator=g).to(dt); vc = torch.randn(B * 3, page, Hk, 64, generator=g).to(dt)
        lens = torch.full((B,), 5, dtype=torch.int32)
        bt = torch.arange(B * 3, dtype=torch.int32).reshape(B, 3)
        return q, kc, vc, lens, bt

    def key(q, kc, vc, lens, bt, **kw):
        a = dict(k=None, v=None, rotary_cos=None, rotary_sin=None, cache_seqlens=lens, cache_batch_idx=None, cache_leftpad=None,
                 block_table=bt, softmax_scale=None, causal=True, window_size=(-1, -1), softcap=0.0, rotary_interleaved=True,
                 alibi_slopes=None, num_splits=0, k_descale=None, v_descale=None)
        a.update(kw)
        return fi._kv_plan_key(q, kc, vc, a["k"], a["v"], a["rotary_cos"], a["rotary_sin"], a["cache_seqlens"], a["cache_batch_idx"],
                               a["cache_leftpad"], a["block_table"], a["softmax_scale"], a["causal"], a["window_size"], a["softcap"],
                               a["rotary_interleaved"], a["alibi_slopes"], a["num_splits"], a["k_descale"], a["v_descale"])

    k0 = key(*mk(seed=0))
    assert k0 is not None and k0 == key(*mk(seed=1)) and hash(k0) == hash(key(*mk(seed=1)))
    assert k0 != key(*mk(B=3))
    assert k0 != key(*mk(dt=torch.bfloat16))
    assert k0 != key(*mk(), causal=False)
    assert k0 != key(*mk(), window_size=(7, 0))
    assert k0 != key(*mk(), num_splits=4)
    q, kc, vc, lens, bt = mk()
    assert k0 != key(q, kc, vc, lens, None)
    assert k0 != key(q, kc[:, :, :1].expand(-1, -1, 2, -1), vc, lens, bt)        # same shape, other strides
    assert key(q, kc, vc, 5, bt) is None                                          # int cache_seqlens: slow path
    assert key(q[..., ::2], kc, vc, lens, bt) is None                             # last dim not contiguous: slow path
    # unit last stride but NOT contiguous: the slow path copies these (maybe_contiguous), so the plan's strides would belong
    # to the copy - such calls must not be keyed (round-4 advisor finding: q = hidden[:, -1:], block_table[:, :n])
    hidden = torch.randn(2, 7, 4, 64).to(torch.float16)
    assert hidden[:, -1:].stride(-1) == 1 and not hidden[:, -1:].is_contiguous()
    assert key(hidden[:, -1:], kc, vc, lens, bt) is None
    wide = torch.arange(2 * 5, dtype=torch.int32).reshape(2, 5)
    assert key(q, kc, vc, lens, wide[:, :3]) is None
    assert key(q, kc, vc, torch.zeros(4, dtype=torch.int32)[::2], bt) is None
    kn = torch.randn(2, 3, 2, 64).to(torch.float16)
    assert key(q, kc, vc, lens, bt, k=kn[:, :1], v=kn[:, :1]) is None
    assert key(q, kc, vc, lens, bt, k=kn[:, :1].contiguous(), v=kn[:, :1].contiguous()) is not None
    # descales: numbers are part of the key as floats, tensors take the slow path (no device read while building a key)
    assert key(q, kc, vc, lens, bt, k_descale=2, v_descale=1) == key(q, kc, vc, lens, bt, k_descale=2.0, v_descale=1.0)
    assert key(q, kc, vc, lens, bt, k_descale=torch.tensor(2.0)) is None


def test_kvcache_plan_table_is_lru():
    """_KV_PLANS drops the least recently USED geometry when full (it used to be wiped as a whole)."""
    from flash_attn_mi355 import flash_attn_interface as fi
    saved = fi._KV_PLANS.copy()
    try:
        fi._KV_PLANS.clear()
        for i in range(fi._KV_PLANS_MAX):
            fi._KV_PLANS[("g", i)] = (b"", 0, ())
        fi._KV_PLANS.move_to_end(("g", 0))                # what a hit does
        while len(fi._KV_PLANS) >= fi._KV_PLANS_MAX:      # what an insert does
            fi._KV_PLANS.popitem(last=False)
        fi._KV_PLANS[("g", "new")] = (b"", 0, ())
        assert ("g", 0) in fi._KV_PLANS and ("g", 1) not in fi._KV_PLANS and len(fi._KV_PLANS) == fi._KV_PLANS_MAX
    finally:
        fi._KV_PLANS.clear(); fi._KV_PLANS.update(saved)


def test_backward_ds_handoff_workspace(lib):
    """FA_FLAG_DS_HANDOFF (opt-in): where the dense D = 128 backward runs its two generated kernels, fills the chip (no split
    launch) and all three gradients are requested, the workspace grows by the dS tiles - [B][Hq][4 ceil(Sk / 128)][ceil(Sq / 32)]
    tiles of 2 KiB behind the statistics planes (fa_bwd_dq_ds.hip); anywhere else the flag changes nothing."""
    q = lib.lib.fa_bwd_workspace_bytes
    al = lambda x: (x + 255) & ~255
    p = _dense(lib, 8, 4096, 16, 16, 128)
    base = q(ctypes.byref(p))
    p.flags = lib.FA_FLAG_DS_HANDOFF
    assert q(ctypes.byref(p)) == base                        # (no gradient pointers yet: dq / dk / dv all have to be asked for)
    p.dq = p.dk = p.dv = 4096                                # (any non-NULL value: the query only tests the pointers)
    assert q(ctypes.byref(p)) == al(2 * 8 * 16 * 4096 * 4) + 8 * 16 * 128 * 128 * 2048
    p.dq = 0
    assert q(ctypes.byref(p)) == base
    p = _dense(lib, 2, 1000, 8, 8, 128)                      # ragged: 32 row tiles, 4 x 8 key blocks
    p.dq = p.dk = p.dv = 4096
    p.flags = lib.FA_FLAG_DS_HANDOFF | lib.FA_FLAG_NO_DKV_SPLIT
    assert q(ctypes.byref(p)) == al(2 * 2 * 8 * 1000 * 4) + 2 * 8 * 32 * 32 * 2048
    for change in ("head64", "dropout", "softcap", "split"):
        p = _dense(lib, 8, 4096, 16, 16, 64 if change == "head64" else 128) if change != "split" else _dense(lib, 1, 4096, 32, 8, 128)
        p.dq = p.dk = p.dv = 4096
        if change == "dropout":
            p.p_dropout = 0.1
        if change == "softcap":
            p.softcap = 30.0
        want = q(ctypes.byref(p))
        p.flags = lib.FA_FLAG_DS_HANDOFF
        assert q(ctypes.byref(p)) == want, change            # other kernels / a split launch: the flag is ignored


def test_forward_key_split_workspace(lib):
    """FA_FLAG_FWD_KEY_SPLIT (opt-in): a causal dense D = 128 launch of at most one wave of 256-row blocks (B x Hq x ceil(Sq / 256)
    <= 256 CUs without a GPU) asks for fp32 partial outputs + LSEs of its split blocks; everything else asks for nothing."""
    q = lib.lib.fa_fwd_workspace_bytes
    p = _dense(lib, 1, 2048, 32, 32, 128)
    assert q(ctypes.byref(p)) == 0                             # opt-in
    p.flags = lib.FA_FLAG_FWD_KEY_SPLIT
    # tiles per block 4 .. 32, 18 per CU if the work could be cut at will -> parts of <= 13 tiles: blocks 3 .. 7 split (2, 2, 2, 3, 3)
    rows_split = 2048 - 3 * 256
    assert q(ctypes.byref(p)) == 3 * 1 * 32 * rows_split * (128 + 1) * 4
    p.is_causal = 0
    assert q(ctypes.byref(p)) == 0                             # no causal imbalance
    p = _dense(lib, 8, 4096, 16, 16, 128)                      # BASELINE config 2: 2048 blocks, the paired queue balances them
    p.flags = lib.FA_FLAG_FWD_KEY_SPLIT
    assert q(ctypes.byref(p)) == 0
    p = _dense(lib, 1, 2048, 32, 32, 64)                       # other head dims: compiler kernels
    p.flags = lib.FA_FLAG_FWD_KEY_SPLIT
    assert q(ctypes.byref(p)) == 0
