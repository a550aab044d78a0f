// fa_api.hip - extern "C" entry points of libfa_mi355.so (see include/fa_mi355.h).
//
// Host-side validation and flag normalisation mirror the reference's wrappers:
//   dense    kernel/fused_mha_forward.cu:317-371,409-413
//   varlen   kernel/fused_mha_forward_varlen.cu:371-482
//   kvcache  kernel/fused_mha_forward_kvcache.cu:416-472,488,582-598
//   backward kernel/fused_mha_backward.cu:577-692, kernel/fused_mha_backward_varlen.cu:636-765
// Errors never cross the boundary as exceptions: negative status + fa_last_error().
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include "fa_common.h"

namespace fa {
int launch_fwd(const KArgs& a, hipStream_t stream);
size_t fwd_split_workspace_bytes(const KArgs& a);        // fa_fwd_asm.hip: forward key split of one-wave causal launches
int launch_gather_rows(const void* src, const int64_t* idx, void* dst, int64_t n_idx, int64_t row_bytes,
                       int64_t src_stride, int64_t n_src_rows, hipStream_t stream);
int launch_scatter_rows(const void* src, const int64_t* idx, void* dst, int64_t n_idx, int64_t n_dst_rows,
                        int64_t row_bytes, int sorted_unique, hipStream_t stream);
int launch_bwd(const KArgs& a, hipStream_t stream);
size_t bwd_workspace_bytes(const fa_params& p);
int launch_kvcache_append(const KArgs& a, hipStream_t stream);
int launch_decode(const KArgs& a, hipStream_t stream);
size_t decode_workspace_bytes(const fa_params& p);
bool decode_applicable(const fa_params& p);
bool decode_takes(const fa_params& p);
}  // namespace fa

static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define FA_CHECK(cond, ...)                                              \
    do {                                                                 \
        if (!(cond)) return fail(FA_ERR_INVALID_ARGUMENT, __VA_ARGS__);  \
    } while (0)

static int check_hip(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return FA_OK;
}

// Experiment switches, read ONCE (first call): FA_VARLEN_GRID=1 keeps the batch x max_seqlen grid for varlen.
static bool varlen_grid_env() {
    static const bool on = getenv("FA_VARLEN_GRID") != nullptr;
    return on;
}

static bool supported_head_dim(int d) { return d == 64 || d == 128 || d == 256; }

// Checks shared by every op (reference: fused_mha_forward.cu:324-340).
static int check_common(const fa_params& p, bool need_out) {
    const bool no_keys = (p.seqlen_k == 0 && !p.cu_seqlens_k);
    FA_CHECK(p.q && (no_keys || (p.k && p.v)), "q, k, v must not be NULL");
    FA_CHECK(!need_out || (p.o && p.lse), "o and lse must not be NULL");
    FA_CHECK(p.dtype == FA_FP16 || p.dtype == FA_BF16, "q must be fp16 or bf16");
    FA_CHECK((p.flags & ~(FA_FLAG_KEEP_WINDOW | FA_FLAG_NO_DKV_SPLIT | FA_FLAG_DS_HANDOFF | FA_FLAG_FWD_KEY_SPLIT)) == 0, "fa_params::flags has unknown bits set (zero-initialise the struct)");
    FA_CHECK(p.batch > 0, "batch size must be positive");
    FA_CHECK(p.head_dim <= 256, "head dimension must be <= 256");
    FA_CHECK(p.head_dim % 8 == 0, "head dimension must be multiple of 8");
    FA_CHECK(p.nheads_k > 0 && p.nheads_q % p.nheads_k == 0, "H_Q must be divisible by H_K for GQA/MQA");
    FA_CHECK(p.p_dropout >= 0.f && p.p_dropout < 1.f, "p_dropout must be in [0, 1)");
    if (p.softcap > 0.f) FA_CHECK(p.p_dropout == 0.f, "Softcapping does not support dropout for now");
    const int kv_al = p.kv_dtype == FA_FP8_E4M3 ? 16 : 8;
    FA_CHECK((p.q_row_stride % 8) == 0 && (p.q_head_stride % 8) == 0 && (p.k_row_stride % kv_al) == 0 &&
             (p.k_head_stride % kv_al) == 0 && (p.v_row_stride % kv_al) == 0 && (p.v_head_stride % kv_al) == 0,
             "q/k/v strides must be multiples of 16 bytes");
    FA_CHECK((reinterpret_cast<uintptr_t>(p.q) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.k) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(p.v) & 15) == 0, "q/k/v must be 16-byte aligned");
    if (!supported_head_dim(p.head_dim))
        return fail(FA_ERR_UNSUPPORTED, "head dimension %d has no gfx950 kernel in this build (64, 128, 256)", p.head_dim);
    // The kernels address one (batch, head) slice through a buffer descriptor: 32-bit byte offsets.  Packed (varlen) tensors:
    // every kernel rebases its pointers at the sequence's first row in 64-bit arithmetic (q_row0 / k_row0 x row stride - the
    // reference offsets with size_t, include/template.h:199-217), so the slice is ONE SEQUENCE (max_seqlen rows), not the
    // total_q / total_k rows of the packed tensor; paged caches: one page.
    {
        const int64_t rows_q = p.seqlen_q;
        const int64_t rows_k = p.block_table ? p.page_block_size : p.seqlen_k;
        FA_CHECK(p.head_dim_v >= 0 && p.head_dim_v <= p.head_dim && p.head_dim_v % 8 == 0,
                 "head_dim_v must be a multiple of 8 in [0, head_dim]");
        const bool narrow = p.head_dim_v > 0 && p.head_dim_v < p.head_dim;
        const int64_t lim = (int64_t)1 << (narrow ? 31 : 32);
        // q: 2 GiB - its rows are fetched through one descriptor whose offset 0x80000000 must lie OUTSIDE the slice (rows past the
        // sequence and columns past the valid width come back as zeros from the range check)
        FA_CHECK(rows_q * p.q_row_stride * 2 < ((int64_t)1 << 31),
                 "one (batch / sequence, head) slice of q spans more than 2 GiB: not addressable by the gfx950 kernels");
        FA_CHECK(rows_k * p.k_row_stride * 2 < lim && rows_k * p.v_row_stride * 2 < lim,
                 "one (batch / sequence, head) slice of k/v spans more than 4 GiB: not addressable by the gfx950 kernels");
    }
    return FA_OK;
}

// Flag normalisation, the reference's (fused_mha_forward.cu:343-352): a window of >= seqlen_k keys is dropped.  For
// seqlen_q <= seqlen_k that cannot change a result.  For seqlen_q > seqlen_k a right window of seqlen_k <= wr < seqlen_q - 1
// keys still hides key j' > i + wr from the first rows and the reference un-masks them; the drop-in API does the same.
// FA_FLAG_KEEP_WINDOW (this library's context-parallel wrapper: all queries over a slice of the keys, the shard's causal
// offset as a right window) drops a right window only where it hides nothing.
static void normalize(fa_params& p, bool kvcache) {
    if (p.seqlen_q == 1 && !p.alibi_slopes) p.is_causal = 0;
    if (kvcache && p.is_causal) p.window_right = 0;
    if (p.window_left >= p.seqlen_k) p.window_left = -1;
    const bool keep = (p.flags & FA_FLAG_KEEP_WINDOW) != 0;
    if (p.window_right >= p.seqlen_k && (!keep || p.window_right >= p.seqlen_q - 1)) p.window_right = -1;
}

static fa::KArgs make_args(const fa_params& p, int block_m) {
    fa::KArgs a;
    memset(&a, 0, sizeof(a));
    a.p = p;
    a.n_qblocks_total = (p.seqlen_q + block_m - 1) / block_m;
    // causal-like masks make late q-blocks heavier: pair block i with its mirror (equal work)
    a.pair_qblocks = ((p.is_causal || p.window_right >= 0) && p.window_left < 0 && a.n_qblocks_total >= 2) ? 1 : 0;
    a.n_qblocks = a.pair_qblocks ? (a.n_qblocks_total + 1) / 2 : a.n_qblocks_total;
    a.has_bias = (p.alibi_slopes != nullptr) || (p.softcap > 0.f);
    a.scale_log2e = p.softmax_scale * fa::kLog2e;
    a.rp_dropout = 1.0f;
    if (p.p_dropout > 0.f) {
        const float keep = 1.0f - p.p_dropout;
        const float t = keep * 4294967295.0f;            // fp32 on purpose (== 2^32 * keep)
        a.drop_thr = t >= 4294967295.0f ? 0xffffffffu : (uint32_t)t;
        a.rp_dropout = 1.0f / keep;
    }
    return a;
}

extern "C" {

int fa_abi_version(void) { return FA_ABI_VERSION; }
size_t fa_params_size(void) { return sizeof(fa_params); }
const char* fa_last_error(void) { return g_last_error.c_str(); }
const char* fa_build_info(void) {
    return "libfa_mi355: gfx950 (CDNA4) hand-written HIP; mfma_f32_32x32x16_{bf16,f16}; head_dim {64,128,256}; "
           "ops fwd/bwd/varlen_fwd/varlen_bwd/fwd_kvcache";
}


// Decode issued through the varlen op (vLLM-style callers: every sequence brings the same few query tokens, K / V are
// paged with a block_table, lengths come from seqused_k or cu_seqlens_k): the layout is the kv-cache op's - q [B, T_q, H, D]
// with batch stride T_q rows, LSE [H, B T_q] - so the decode kernels (GQA packing, split-KV) can serve it instead of
// fa_fwd_kernel's one workgroup per sequence and head (B 1, H 32/8, 8 k context: 240 -> 32 us; tools/varlen_decode_probe.py).
// Needs the split-KV workspace: fa_fwd_workspace_bytes() reports it, and without it the general path runs as before.
static bool varlen_decode_route(const fa_params& p, fa_params& d) {
    if (!p.block_table || !p.cu_seqlens_q || !p.cu_seqlens_k || p.p_dropout > 0.f || p.dmask) return false;
    if (p.batch <= 0 || p.seqlen_q <= 0 || p.total_q != (int64_t)p.batch * p.seqlen_q) return false;   // uniform T_q (host-checkable)
    if ((p.kv_dtype != p.dtype && p.kv_dtype != FA_FP8_E4M3) || p.page_block_size <= 0 || p.page_block_size % 16 != 0) return false;
    d = p;
    d.cache_seqlens = p.seqused_k;                       // NULL: cu_seqlens_k differences (dec_cache_len in fa_decode.hip)
    d.seqused_k = nullptr;
    // cu_seqlens_q stays: the kernels run in varlen-q mode (DecArgs::cu_q, class bound T = seqlen_q) and take every
    // sequence's rows and row count from the device.  total_q == batch x max_seqlen_q does NOT prove cu_seqlens_q[-1] ==
    // total_q - q may carry padding rows behind the last sequence (graph-captured serving steps), and then sequence b is
    // not at row b T (round-3 advisor finding)
    d.q_batch_stride = 0; d.o_batch_stride = 0; d.lse_batch_stride = 0;
    d.k_new = d.v_new = nullptr; d.seqlen_new = 0;
    d.rotary_cos = d.rotary_sin = nullptr; d.rotary_dim = 0;
    d.cache_batch_idx = nullptr; d.cache_leftpad = nullptr;
    d.num_splits = 0;
    if (d.seqlen_q == 1 && !d.alibi_slopes) d.is_causal = 0;
    if (d.is_causal) d.window_right = 0;
    return fa::decode_takes(d);
}

// A MIXED batch through the varlen op (vLLM-style unified step: many sequences with one - or a few - query tokens next to a
// prefill chunk, paged K / V): the host cannot see the lengths, but it can see that most sequences must be short
// ((total_q - max_seqlen_q) / (batch - 1) <= 64).  Then the decode kernels run over ALL sequences in varlen-q mode and keep
// the ones with 1 .. T query rows (DecArgs::cu_q; the others' workgroups leave at once), and fa_fwd_kernel runs with
// KArgs::skip_short_q = T for the rest: 32 decode sequences + a 512-token chunk over 8 k contexts 792 -> ~350 us
// (tools/mixed_batch_probe.py).  T = 32 / G query rows (at most 8): one 32-row block per kv-head.
static bool varlen_mixed_route(const fa_params& p, fa_params& d) {
    if (!p.block_table || !p.cu_seqlens_q || !p.cu_seqlens_k || p.p_dropout > 0.f || p.dmask) return false;
    if ((p.kv_dtype != p.dtype && p.kv_dtype != FA_FP8_E4M3) || p.page_block_size <= 0 || p.page_block_size % 16 != 0) return false;
    if (p.batch < 4 || p.nheads_k <= 0 || p.total_q >= (int64_t)p.batch * p.seqlen_q) return false;    // uniform batches: above
    const int G = p.nheads_q / p.nheads_k;
    int T = 32 / (G > 0 ? G : 1);
    T = T < 1 ? 1 : (T > 8 ? 8 : T);
    if (p.seqlen_q <= T) return false;                                           // everything is short: the uniform route or the general kernel
    // (the other sequences average more than 64 rows: few of them can be decode steps.  A wrong yes costs one launch of
    //  workgroups that leave at once, ~10 us; a wrong no costs the decode sequences a 128-row tile and a full stream each)
    if ((p.total_q - p.seqlen_q) > (int64_t)(p.batch - 1) * 64) return false;
    d = p;
    d.cache_seqlens = p.seqused_k;                       // NULL: cu_seqlens_k differences
    d.seqused_k = nullptr;
    d.seqlen_q = T;                                      // the class bound; rows per sequence come from cu_seqlens_q (kept)
    d.q_batch_stride = 0; d.o_batch_stride = 0; d.lse_batch_stride = 0;
    d.k_new = d.v_new = nullptr; d.seqlen_new = 0;
    d.rotary_cos = d.rotary_sin = nullptr; d.rotary_dim = 0;
    d.cache_batch_idx = nullptr; d.cache_leftpad = nullptr;
    d.num_splits = 0;
    if (d.is_causal) d.window_right = 0;
    return fa::decode_takes(d);
}

size_t fa_fwd_workspace_bytes(const fa_params* p) {
    fa_params d;
    if (!p) return 0;
    if (varlen_decode_route(*p, d)) return fa::decode_workspace_bytes(d);
    if (varlen_mixed_route(*p, d)) return fa::decode_workspace_bytes(d);
    if (!p->cu_seqlens_q && !p->cu_seqlens_k && !p->block_table && p->seqlen_q > 0 && p->seqlen_k > 0 && p->kv_dtype == p->dtype) {
        // fa_fwd: partial outputs of a key-split one-wave causal launch (the struct as fa_fwd will see it)
        fa_params q = *p;
        q.seqused_k = nullptr;
        normalize(q, false);
        return fa::fwd_split_workspace_bytes(make_args(q, 128));
    }
    return 0;
}
size_t fa_bwd_workspace_bytes(const fa_params* pp) {
    if (!pp) return 0;
    fa_params p = *pp;
    // the flags as fa_bwd / fa_varlen_bwd will see them (the split of small dK/dV launches depends on the mask's shape)
    if (p.seqlen_q > 0 && p.seqlen_k > 0) normalize(p, false);
    return fa::bwd_workspace_bytes(p);
}
size_t fa_fwd_kvcache_workspace_bytes(const fa_params* pp) {
    if (!pp) return 0;
    fa_params p = *pp;                                   // what fa_fwd_kvcache launches with (the decode code reads these fields)
    p.cu_seqlens_q = p.cu_seqlens_k = p.seqused_k = nullptr;
    return fa::decode_workspace_bytes(p);
}

int fa_fwd(const fa_params* pp, void* stream) {
    if (!pp) return fail(FA_ERR_INVALID_ARGUMENT, "params is NULL");
    fa_params p = *pp;
    p.cu_seqlens_q = p.cu_seqlens_k = p.seqused_k = nullptr;
    p.block_table = nullptr;
    int rc = check_common(p, true);
    if (rc) return rc;
    FA_CHECK(p.kv_dtype == p.dtype, "k/v must have the same dtype as q");
    FA_CHECK(p.seqlen_q >= 0 && p.seqlen_k >= 0, "sequence lengths must be non-negative");
    if (p.seqlen_q == 0) return FA_OK;
    normalize(p, false);
    fa::KArgs a = make_args(p, 128);
    rc = fa::launch_fwd(a, static_cast<hipStream_t>(stream));
    if (rc) return fail(FA_ERR_UNSUPPORTED, "no forward kernel for this configuration");
    return check_hip("fa_fwd launch");
}

int fa_varlen_fwd(const fa_params* pp, void* stream) {
    if (!pp) return fail(FA_ERR_INVALID_ARGUMENT, "params is NULL");
    fa_params p = *pp;
    int rc = check_common(p, true);
    if (rc) return rc;
    // fp8-e4m3 K / V (this build's extension, as in fa_fwd_kvcache): paged caches, forward only, head dim 64 / 128
    if (p.kv_dtype == FA_FP8_E4M3) {
        FA_CHECK(p.block_table, "fp8 K/V through the varlen op: paged K/V (block_table) only");
        FA_CHECK((p.head_dim == 64 || p.head_dim == 128) && p.head_dim_v == 0, "fp8 K/V: head dimension 64 or 128");
        FA_CHECK(p.p_dropout == 0.f && !p.dmask, "fp8 K/V: no dropout");
    } else {
        FA_CHECK(p.kv_dtype == p.dtype, "k/v must have the same dtype as q (or fp8-e4m3 for paged K/V)");
    }
    FA_CHECK(p.cu_seqlens_q && p.cu_seqlens_k, "cu_seqlens_q and cu_seqlens_k are required");
    if (p.block_table) {
        FA_CHECK(p.page_block_size > 0, "page_block_size must be positive");
        FA_CHECK(p.page_block_size % 16 == 0, "Paged KV cache block size must be divisible by 16");
    }
    if (p.p_dropout > 0.f && p.block_table) return fail(FA_ERR_UNSUPPORTED, "dropout with paged K/V is not supported");
    if (p.total_q == 0 || p.seqlen_q == 0) return FA_OK;
    {
        fa_params d;
        if (varlen_decode_route(p, d)) {
            const size_t need = fa::decode_workspace_bytes(d);
            if (need == 0 || (d.workspace && d.workspace_bytes >= need)) {
                fa::KArgs ad = make_args(d, 128);
                ad.seqlens_k = d.cache_seqlens;
                ad.kv_mode = 1;
                rc = fa::launch_decode(ad, static_cast<hipStream_t>(stream));
                if (rc) return fail(FA_ERR_UNSUPPORTED, "no decode kernel for this varlen configuration");
                return check_hip("fa_varlen_fwd (decode kernels) launch");
            }
        }
    }
    int skip_short = 0;
    {
        fa_params d;
        if (varlen_mixed_route(p, d)) {
            const size_t need = fa::decode_workspace_bytes(d);
            if (need == 0 || (d.workspace && d.workspace_bytes >= need)) {
                fa::KArgs ad = make_args(d, 128);
                ad.seqlens_k = d.cache_seqlens;
                ad.kv_mode = 1;
                rc = fa::launch_decode(ad, static_cast<hipStream_t>(stream));
                if (rc) return fail(FA_ERR_UNSUPPORTED, "no decode kernel for the short sequences of this varlen batch");
                skip_short = d.seqlen_q;             // the general kernel below leaves those sequences out
            }
        }
    }
    normalize(p, false);
    fa::KArgs a = make_args(p, 128);
    a.seqlens_k = p.seqused_k;
    a.skip_short_q = skip_short;
    if (p.total_q > 0 && !varlen_grid_env()) {     // flat work list (fa_common.h: decode_work_flat)
        a.flat_blocks = p.total_q / 128 + p.batch;
        a.pair_qblocks = 0;
        a.n_qblocks = a.n_qblocks_total;
    }
    rc = fa::launch_fwd(a, static_cast<hipStream_t>(stream));
    if (rc) return fail(FA_ERR_UNSUPPORTED, "no varlen forward kernel for this configuration");
    return check_hip("fa_varlen_fwd launch");
}

int fa_fwd_kvcache(const fa_params* pp, void* stream) {
    if (!pp) return fail(FA_ERR_INVALID_ARGUMENT, "params is NULL");
    fa_params p = *pp;
    p.cu_seqlens_q = p.cu_seqlens_k = p.seqused_k = nullptr;
    int rc = check_common(p, true);
    if (rc) return rc;
    FA_CHECK(p.kv_dtype == p.dtype || p.kv_dtype == FA_FP8_E4M3, "kcache/vcache must match q dtype or be fp8-e4m3");
    FA_CHECK(p.p_dropout == 0.f, "kvcache attention has no dropout");
    const bool paged = p.block_table != nullptr;
    if (paged) {
        FA_CHECK(!p.cache_batch_idx, "Paged KVcache does not support cache_batch_idx");
        // (the reference wants multiples of 256, fused_mha_forward_kvcache.cu:488; here any multiple of 16 works - pages of
        //  64 tokens and more take the aligned fast paths, smaller ones the per-row lookups)
        FA_CHECK(p.page_block_size > 0 && p.page_block_size % 16 == 0,
                 "Paged KV cache block size must be divisible by 16");
    }
    if (p.k_new || p.v_new) {
        FA_CHECK(p.k_new && p.v_new, "If key is supplied, value must also be passed in");
        FA_CHECK(p.cache_seqlens, "If key is supplied, seqlens_k must also be passed in");
        FA_CHECK(p.seqlen_new > 0, "seqlen_new must be positive when k/v are supplied");
        FA_CHECK(p.seqlen_new <= p.seqlen_k && p.seqlen_q <= p.seqlen_k,
                 "new keys / queries do not fit the cache (fused_mha_forward_kvcache.cu: T_Q <= max_seqlen_k)");
    } else {
        p.seqlen_new = 0;
    }
    if (p.rotary_dim > 0) {
        FA_CHECK(p.k_new, "If rotary cos/sin are provided, new key / value to be appended to KV cache must also be provided");
        FA_CHECK(p.rotary_cos && p.rotary_sin, "rotary_cos and rotary_sin must both be given");
        FA_CHECK(p.rotary_dim <= (p.head_dim_v > 0 ? p.head_dim_v : p.head_dim), "rotary_dim must be <= headdim");
        FA_CHECK(p.rotary_dim % 16 == 0, "rotary_dim must be divisible by 16");
        // every position the kernels can touch is < the cache capacity (append and rotation are range-guarded)
        FA_CHECK(p.seqlen_ro >= p.seqlen_k, "rotary_cos / rotary_sin must cover the cache capacity (seqlen_ro >= seqlen_k)");
    }
    if (p.num_splits < 0) return fail(FA_ERR_INVALID_ARGUMENT, "num_splits must be >= 0");
    // reference: fused_mha_forward_kvcache.cu:465-472
    normalize(p, true);
    if (p.softcap > 0.f) {
        FA_CHECK(p.window_left < 0 && p.window_right < 0, "Softcap + window not supported");
        FA_CHECK(!p.alibi_slopes, "Softcap + ALiBi not supported");
    }
    if (p.seqlen_q == 0) return FA_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    fa::KArgs a = make_args(p, 128);
    a.seqlens_k = p.cache_seqlens;
    a.seqlen_k_add = p.seqlen_new;
    a.kv_batch_idx = p.cache_batch_idx;
    a.leftpad_k = p.cache_leftpad;
    a.kv_mode = 1;
    rc = fa::launch_decode(a, s);
    if (rc == -2) return fail(FA_ERR_UNSUPPORTED, "no kvcache kernel for this configuration (fp8 caches: head_dim 64 or 128)");
    if (rc == -1) return fail(FA_ERR_INVALID_ARGUMENT, "workspace too small: query fa_fwd_kvcache_workspace_bytes()");
    if (rc) return rc;
    return check_hip("fa_fwd_kvcache launch");
}

int fa_bwd(const fa_params* pp, void* stream) {
    if (!pp) return fail(FA_ERR_INVALID_ARGUMENT, "params is NULL");
    fa_params p = *pp;
    p.cu_seqlens_q = p.cu_seqlens_k = p.seqused_k = nullptr;
    p.block_table = nullptr;
    int rc = check_common(p, true);
    if (rc) return rc;
    FA_CHECK(p.dout && p.softmax_d, "dout and softmax_d must not be NULL");
    FA_CHECK((p.dk == nullptr) == (p.dv == nullptr), "dk and dv must be given (or left NULL) together");
    FA_CHECK(p.kv_dtype == p.dtype, "k/v must have the same dtype as q");
    if (p.seqlen_q == 0 && p.seqlen_k == 0) return FA_OK;
    normalize(p, false);
    fa::KArgs a = make_args(p, 128);
    rc = fa::launch_bwd(a, static_cast<hipStream_t>(stream));
    if (rc == -2) return fail(FA_ERR_UNSUPPORTED, "no backward kernel for this configuration");
    if (rc) return rc;
    return check_hip("fa_bwd launch");
}

int fa_varlen_bwd(const fa_params* pp, void* stream) {
    if (!pp) return fail(FA_ERR_INVALID_ARGUMENT, "params is NULL");
    fa_params p = *pp;
    p.block_table = nullptr;
    int rc = check_common(p, true);
    if (rc) return rc;
    FA_CHECK(p.dout && p.softmax_d, "dout and softmax_d must not be NULL");
    FA_CHECK((p.dk == nullptr) == (p.dv == nullptr), "dk and dv must be given (or left NULL) together");
    FA_CHECK(p.cu_seqlens_q && p.cu_seqlens_k, "cu_seqlens_q and cu_seqlens_k are required");
    FA_CHECK(p.kv_dtype == p.dtype, "k/v must have the same dtype as q");
    if (p.total_q == 0) {
        // no query rows: nothing flows into K / V - the gradients are zeros, written here (not left to the caller)
        const int64_t rows = p.total_k;
        const size_t width = (size_t)(p.head_dim_v > 0 ? p.head_dim_v : p.head_dim) * 2;
        hipStream_t s = static_cast<hipStream_t>(stream);
        for (int h = 0; h < p.nheads_k && rows > 0 && p.dk; ++h) {
            if (hipMemset2DAsync(reinterpret_cast<uint16_t*>(p.dk) + (int64_t)h * p.dk_head_stride, (size_t)p.dk_row_stride * 2, 0, width, (size_t)rows, s) != hipSuccess ||
                hipMemset2DAsync(reinterpret_cast<uint16_t*>(p.dv) + (int64_t)h * p.dv_head_stride, (size_t)p.dv_row_stride * 2, 0, width, (size_t)rows, s) != hipSuccess)
                return fail(FA_ERR_LAUNCH, "hipMemset2DAsync(dk / dv) failed");
        }
        return FA_OK;
    }
    normalize(p, false);
    fa::KArgs a = make_args(p, 128);
    if (!varlen_grid_env()) {                      // flat work lists (fa_common.h: decode_work_flat)
        a.flat_blocks = p.total_q / 128 + p.batch;
        a.pair_qblocks = 0;
        a.n_qblocks = a.n_qblocks_total;
    }
    rc = fa::launch_bwd(a, static_cast<hipStream_t>(stream));
    if (rc == -2) return fail(FA_ERR_UNSUPPORTED, "no varlen backward kernel for this configuration");
    if (rc) return rc;
    return check_hip("fa_varlen_bwd launch");
}

int fa_gather_rows(const void* src, const int64_t* indices, void* dst, int64_t n_idx, int64_t row_bytes,
                   int64_t src_row_stride_bytes, int64_t n_src_rows, void* stream) {
    FA_CHECK(n_idx >= 0 && row_bytes >= 0 && n_src_rows >= 0, "sizes must be non-negative");
    if (n_idx == 0 || row_bytes == 0) return FA_OK;
    FA_CHECK(src && indices && dst, "src, indices and dst must not be NULL");
    FA_CHECK(row_bytes % 16 == 0 && src_row_stride_bytes % 16 == 0 && src_row_stride_bytes >= row_bytes,
             "row_bytes and src_row_stride_bytes must be multiples of 16 (stride >= row)");
    FA_CHECK(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0), "src and dst must be 16-byte aligned");
    fa::launch_gather_rows(src, indices, dst, n_idx, row_bytes, src_row_stride_bytes, n_src_rows, static_cast<hipStream_t>(stream));
    return check_hip("fa_gather_rows launch");
}

int fa_scatter_rows(const void* src, const int64_t* indices, void* dst, int64_t n_idx, int64_t n_dst_rows,
                    int64_t row_bytes, int sorted_unique, void* stream) {
    FA_CHECK(n_idx >= 0 && row_bytes >= 0 && n_dst_rows >= 0, "sizes must be non-negative");
    if (n_dst_rows == 0 || row_bytes == 0) return FA_OK;
    FA_CHECK(dst && (n_idx == 0 || (src && indices)), "src, indices and dst must not be NULL");
    FA_CHECK(row_bytes % 16 == 0, "row_bytes must be a multiple of 16");
    FA_CHECK(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0), "src and dst must be 16-byte aligned");
    if (fa::launch_scatter_rows(src, indices, dst, n_idx, n_dst_rows, row_bytes, sorted_unique, static_cast<hipStream_t>(stream)))
        return fail(FA_ERR_INVALID_ARGUMENT, "hipMemsetAsync failed");
    return check_hip("fa_scatter_rows launch");
}

}  // extern "C"
