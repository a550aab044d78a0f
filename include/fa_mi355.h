/*
 * fa_mi355.h - C ABI of libfa_mi355.so: the MI355X (gfx950 / CDNA4) fused attention path.
 *
 * This is the drop-in boundary for the reference's private extension module
 * `flash_attn_v100_cuda` (ai-bond/flash-attention-v100).  Each entry point replaces one
 * op of that module; the reference prototypes are cited per function below
 * (include/mha.h and kernel/fused_mha_api.cpp of the reference).
 *
 * Conventions
 *   - plain C: raw device pointers, sizes, strides IN ELEMENTS, POD structs, an explicit
 *     HIP stream (void* == hipStream_t).  No torch / ATen types.
 *   - never allocates: the caller owns every buffer (outputs, workspaces).  Workspace
 *     sizes come from the *_workspace_bytes() queries.
 *   - never throws: returns FA_OK (0) or a negative fa_status; the message of the last
 *     failure on the calling thread is returned by fa_last_error().
 *   - inputs are borrowed; outputs are written in place; k_cache / v_cache are mutated
 *     in place by fa_fwd_kvcache (reference: kernel/fused_mha_forward_kvcache.cu:134-141).
 *   - launches are asynchronous on `stream`; no host synchronisation inside.
 *   - Tensor layouts are described by strides, so the (B,S,H,D) tensors of the Python API
 *     are passed as they are (the reference permutes + copies to (B,H,S,D) first,
 *     flash_attn_v100/flash_attn_interface.py:36-53).
 */
#ifndef FA_MI355_H
#define FA_MI355_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FA_ABI_VERSION 3   /* 2: fa_bwd / fa_varlen_bwd skip outputs passed as NULL; 3: fa_params::flags (was reserved0) */

/* fa_params::flags.  The reference drops a window of >= seqlen_k keys before anything else (fused_mha_forward.cu:343-352); so
 * do the five ops.  With seqlen_q > seqlen_k that also drops right windows that still hide keys from the first rows - the
 * mask of a context-parallel shard (all queries over a slice of the keys).  FA_FLAG_KEEP_WINDOW keeps such a window; it is an
 * extension for this library's own sharding wrapper and never set by the drop-in Python API. */
#define FA_FLAG_KEEP_WINDOW 1
/* fa_bwd / fa_varlen_bwd.  A dK/dV launch with fewer workgroups than the GPU has room for (batch x kv-heads x 128-key blocks: GQA at
 * micro-batch 1, short-key cross-attention) divides the query rows of each key block over several workgroups and adds their
 * 16-bit partial dK / dV in fp32 (deterministic; needs the workspace fa_bwd_workspace_bytes() reports).  dK / dV then differ in
 * the last bit from the one-workgroup-per-key-block result.  FA_FLAG_NO_DKV_SPLIT keeps one workgroup per key block. */
#define FA_FLAG_NO_DKV_SPLIT 2
/* fa_bwd, opt-in, measured at break-even over short runs and 5 % slower sustained (profiles/r06_ds_handoff.txt): where the dense D = 128 backward would run its two generated
 * kernels and all three gradients are requested, the dK/dV kernel hands its 16-bit dS tiles to a one-GEMM dQ kernel through the
 * workspace (2 bytes per (query, key) pair and head: fa_bwd_workspace_bytes() reports it) instead of dQ recomputing S and dP.
 * Same results up to the order of fp32 additions.  Ignored where it does not apply. */
#define FA_FLAG_DS_HANDOFF 4
/* fa_fwd, opt-in, measured a net loss at the shapes it was built for (profiles/r06_fwd_split.txt).  A causal-like dense D = 128 launch
 * whose 256-row query blocks all get a compute unit at once (batch x heads x ceil(Sq / 256) <= the CU count: micro-batch-1 training,
 * batch-1 prefill, a tensor-parallel shard) runs as long as its heaviest block; with this flag and the workspace
 * fa_fwd_workspace_bytes() reports, the heavy blocks' key ranges are cut into 2 - 4 parts whose fp32 partial outputs a merge kernel
 * combines (deterministic; out / LSE equal the unsplit result up to the order of fp32 additions).  Ignored where it does not apply. */
#define FA_FLAG_FWD_KEY_SPLIT 8

typedef enum fa_dtype {
    FA_FP16 = 0,      /* IEEE half */
    FA_BF16 = 1,      /* bfloat16 */
    FA_FP8_E4M3 = 2   /* OCP e4m3fn, KV cache only */
} fa_dtype;

typedef enum fa_status {
    FA_OK = 0,
    FA_ERR_INVALID_ARGUMENT = -1,   /* a TORCH_CHECK of the reference would have fired */
    FA_ERR_UNSUPPORTED = -2,        /* valid request this build has no kernel for */
    FA_ERR_LAUNCH = -3,             /* HIP runtime reported an error */
    FA_ERR_NO_DEVICE = -4
} fa_status;

/*
 * One parameter block serves all five ops; each op reads the fields that apply to it
 * and ignores the rest (zero-initialise the struct).  Index conventions:
 *   element (b, i, h, d) of q  =  q[b*q_batch_stride + i*q_row_stride + h*q_head_stride + d]
 *   (last dimension contiguous).  Varlen: b*batch_stride is replaced by cu_seqlens[b]*row_stride.
 *   Paged K/V: logical key j of batch b lives in page block_table[b*block_table_batch_stride
 *   + j / page_block_size], row j % page_block_size; k_batch_stride is then the PAGE stride.
 */
typedef struct fa_params {
    /* ---- forward tensors ---- */
    const void* q;            /* [B, Sq, Hq, D]  (varlen: [Tq, Hq, D]) */
    const void* k;            /* [B, Sk, Hk, D]  (varlen: [Tk, Hk, D]; paged: [nblk, page, Hk, D]) */
    const void* v;
    void*       o;            /* same shape as q */
    float*      lse;          /* dense/kvcache [B, Hq, Sq]; varlen [Hq, Tq]  (natural log) */
    int64_t q_batch_stride, q_row_stride, q_head_stride;
    int64_t k_batch_stride, k_row_stride, k_head_stride;
    int64_t v_batch_stride, v_row_stride, v_head_stride;
    int64_t o_batch_stride, o_row_stride, o_head_stride;
    int64_t lse_batch_stride, lse_head_stride;   /* row stride is 1 */

    /* ---- backward tensors (fa_bwd / fa_varlen_bwd) ---- */
    const void* dout;         /* same layout family as o, own strides */
    void*       dq;
    void*       dk;
    void*       dv;
    float*      softmax_d;    /* rowsum(dO * O), same layout as lse; REQUIRED (written) */
    int64_t do_batch_stride, do_row_stride, do_head_stride;
    int64_t dq_batch_stride, dq_row_stride, dq_head_stride;
    int64_t dk_batch_stride, dk_row_stride, dk_head_stride;
    int64_t dv_batch_stride, dv_row_stride, dv_head_stride;

    /* ---- sizes ---- */
    int32_t batch;
    int32_t nheads_q;
    int32_t nheads_k;
    int32_t seqlen_q;         /* dense: Sq; varlen: max_seqlen_q; kvcache: Tq */
    int32_t seqlen_k;         /* dense: Sk; varlen: max_seqlen_k; kvcache: cache capacity
                                 (S_max, or pages_per_seq * page_block_size when paged) */
    int32_t head_dim;         /* kernel width: 64, 128 or 256 */
    int32_t dtype;            /* fa_dtype of q/o/dout/dq/dk/dv */
    int32_t kv_dtype;         /* fa_dtype of k/v (== dtype, or FA_FP8_E4M3 for fa_fwd_kvcache) */

    /* ---- attention options ---- */
    float   softmax_scale;
    float   softcap;          /* 0 = off.  s = softcap * tanh(s / softcap), applied AFTER ALiBi
                                 (reference order, include/mat_mul.h:113-116) */
    int32_t is_causal;        /* bottom-right aligned: key j visible iff j - (Sk - Sq) <= i */
    int32_t window_left;      /* -1 = unbounded */
    int32_t window_right;     /* -1 = unbounded */
    const float* alibi_slopes;      /* fp32 [Hq] or [B, Hq]; NULL = off */
    int64_t alibi_batch_stride;     /* 0 for [Hq] */

    /* ---- dropout (Philox-4x32-10, reference stream: include/softmax.h:97-104) ---- */
    float    p_dropout;       /* probability of dropping */
    uint64_t philox_seed;
    uint64_t philox_offset;
    void*    dmask;           /* optional [B,Hq,Sq,Sk] (varlen: [Tq,Hq,max_sk]) of `dtype`:
                                 +1 kept / -1 dropped; NULL = not requested */

    /* ---- varlen ---- */
    const int32_t* cu_seqlens_q;    /* [B+1] */
    const int32_t* cu_seqlens_k;    /* [B+1] */
    const int32_t* seqused_k;       /* [B] or NULL */
    int32_t        total_q;         /* Tq (varlen) */
    int32_t        total_k;         /* Tk (varlen, non-paged) */

    /* ---- paged KV ---- */
    const int32_t* block_table;     /* [B, max_blocks] or NULL */
    int64_t        block_table_batch_stride;
    int32_t        page_block_size;   /* tokens per page: any multiple of 16 */
    int32_t        head_dim_v;      /* valid columns of every row (multiple of 8, <= head_dim); 0 = head_dim.
                                       Columns [head_dim_v, head_dim) are read as zero and never written: odd
                                       head dims (40, 80, 96, 192 ...) run on the next kernel width without
                                       padded copies of the tensors. */

    /* ---- KV cache (fa_fwd_kvcache) ---- */
    const int32_t* cache_seqlens;   /* [B] or NULL (=0) */
    const int32_t* cache_batch_idx; /* [B] or NULL */
    const int32_t* cache_leftpad;   /* [B] or NULL */
    const void*    k_new;           /* [B, T_new, Hk, D] of `dtype`, or NULL */
    const void*    v_new;
    int64_t knew_batch_stride, knew_row_stride, knew_head_stride;
    int64_t vnew_batch_stride, vnew_row_stride, vnew_head_stride;
    int32_t        seqlen_new;      /* T_new */
    int32_t        rotary_dim;      /* 0 = no rotary */
    const void*    rotary_cos;      /* [seqlen_ro, rotary_dim/2] of `dtype` */
    const void*    rotary_sin;
    int32_t        rotary_interleaved;
    int32_t        seqlen_ro;
    float          k_descale;       /* fp8 cache: value = code * descale (1.0 otherwise) */
    float          v_descale;

    /* ---- split-KV (decode) ---- */
    int32_t num_splits;             /* 0 = heuristic, 1 = no split */
    int32_t flags;                  /* FA_FLAG_* bits, 0 for the reference's semantics (unknown bits are rejected) */
    void*   workspace;              /* >= fa_*_workspace_bytes(params) bytes, or NULL if 0 */
    size_t  workspace_bytes;
} fa_params;

/* ABI self-description (checked by the Python ctypes mirror at load time). */
int         fa_abi_version(void);
size_t      fa_params_size(void);
const char* fa_last_error(void);
const char* fa_build_info(void);           /* arch, compiler, kernel variants */


/* Workspace queries (bytes; 0 = none needed).  fa_fwd_workspace_bytes covers fa_fwd (non-zero only with FA_FLAG_FWD_KEY_SPLIT on one-wave causal launches) and fa_varlen_fwd: non-zero
 * when the call is a decode step issued through the varlen op (every sequence brings the same <= 32 query tokens, paged
 * K / V) or a mixed batch whose sequences are mostly short (decode sequences next to a prefill chunk) - with the workspace
 * the split-KV decode kernels serve the short sequences, without it the general kernel serves everything (same results). */
size_t fa_fwd_workspace_bytes(const fa_params* p);
/* fa_bwd_workspace_bytes: the row-statistics planes of the hand-scheduled D = 128 dK/dV kernel (2 x rows x heads x 4 bytes) and, for
 * dK/dV launches smaller than the GPU (see FA_FLAG_NO_DKV_SPLIT), fp32 partial dK / dV slabs behind them.  A smaller or NULL workspace
 * is legal: the kernels that need no workspace run (same results up to the order of fp32 additions).
 * One query serves fa_bwd AND fa_varlen_bwd: pass the struct EXACTLY as it goes into the op - cu_seqlens_q / cu_seqlens_k select the
 * packed sizing and split decision, and fa_bwd itself ignores them (a dense caller that recycles a struct must clear the varlen
 * fields before the query, or the sizes answer the varlen op; the `workspace_bytes` guard of the ops keeps a mismatch safe - the
 * split is then dropped or the slabs go unused - but not fast).  The split decision reads the CU count of the CURRENT device:
 * query on the device the op will be launched on. */
size_t fa_bwd_workspace_bytes(const fa_params* p);
size_t fa_fwd_kvcache_workspace_bytes(const fa_params* p);

/*
 * fa_fwd - dense forward.  Replaces `flash_attn_v100_cuda.fwd`
 *   reference: include/mha.h:27-41, kernel/fused_mha_forward.cu:301-432
 *   writes o, lse (and dmask when requested and p_dropout > 0).
 */
int fa_fwd(const fa_params* p, void* stream);

/*
 * fa_bwd - dense backward.  Replaces `flash_attn_v100_cuda.bwd`
 *   reference: include/mha.h:67-87, kernel/fused_mha_backward.cu:577-721
 *   reads dout, q, k, v, o, lse; writes dq, dk, dv, softmax_d.  Deterministic
 *   (no atomics).  GQA: dk/dv are summed over the q-heads of each kv-head in-kernel.
 *   Gradients the caller does not need are skipped: dq == NULL -> the dQ kernel does not run;
 *   dk == dv == NULL -> the dK/dV kernel does not run (autograd's needs_input_grad; softmax_d is
 *   always written).  The same holds for fa_varlen_bwd.
 */
int fa_bwd(const fa_params* p, void* stream);

/*
 * fa_varlen_fwd - packed variable-length forward (optionally paged K/V).
 *   Replaces `flash_attn_v100_cuda.varlen_fwd`
 *   reference: include/mha.h:116-139, kernel/fused_mha_forward_varlen.cu:371-566
 */
int fa_varlen_fwd(const fa_params* p, void* stream);

/*
 * fa_varlen_bwd - packed variable-length backward.  Replaces `flash_attn_v100_cuda.varlen_bwd`
 *   reference: include/mha.h:170-195, kernel/fused_mha_backward_varlen.cu:636-807
 */
int fa_varlen_bwd(const fa_params* p, void* stream);

/*
 * fa_fwd_kvcache - append new K/V (+RoPE) into the cache, then attention of q over the
 *   cache (decode / chunked prefill).  Replaces `flash_attn_v100_cuda.fwd_kvcache`
 *   reference: include/mha.h:224-245, kernel/fused_mha_forward_kvcache.cu:416-652
 */
int fa_fwd_kvcache(const fa_params* p, void* stream);

/*
 * Row gather / scatter for the padding helpers on both sides of the varlen path (HBM-bound byte movement).
 * Rows are `row_bytes` bytes (a multiple of 16, 16-byte aligned base pointers), indices are int64 on the device
 * (negative values count from the end, as in torch); no bounds checks beyond that (same contract as the reference's
 * torch.gather / index assignment, flash_attn/bert_padding.py:9-60).
 *
 * fa_gather_rows : dst[i, :] = src[indices[i], :] for i < n_idx.  `src_row_stride_bytes` >= row_bytes lets the
 *   source be a row-strided view.  Replaces `index_first_axis` forward / `index_put_first_axis` backward
 *   (bert_padding.py:9-34, :52-60) as used by `unpad_input` (:79-104).
 * fa_scatter_rows: dst[:] = 0; dst[indices[i], :] = src[i, :].  With `sorted_unique` != 0 (indices ascending
 *   without repeats - what unpad_input produces) it is one pass over dst; otherwise memset + scatter (repeated
 *   indices: one of the rows wins, as in the reference).  Replaces `index_put_first_axis` forward /
 *   `index_first_axis` backward (bert_padding.py:36-50, :22-34) as used by `pad_input` (:135-146).
 */
int fa_gather_rows(const void* src, const int64_t* indices, void* dst, int64_t n_idx, int64_t row_bytes,
                   int64_t src_row_stride_bytes, int64_t n_src_rows, void* stream);
int fa_scatter_rows(const void* src, const int64_t* indices, void* dst, int64_t n_idx, int64_t n_dst_rows,
                    int64_t row_bytes, int sorted_unique, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FA_MI355_H */
