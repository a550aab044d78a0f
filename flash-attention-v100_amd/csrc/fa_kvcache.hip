// fa_kvcache.hip - KV-cache path: append new K/V rows into the (optionally paged) cache with
// RoPE on K, rotate Q, then run attention over the cache.
//
// Replaces kernel/fused_mha_forward_kvcache.cu:24-295 + include/rotary.h:14-264 of the
// reference.  Unlike the reference (where every (q-head, q-block) CTA repeats the append,
// fused_mha_forward_kvcache.cu:134-141) the append is done exactly once by a small
// bandwidth-bound prologue kernel on the same stream.
//
// RoPE (include/rotary.h:91-141): math in fp32, result rounded to the 16-bit io type.
//   interleaved: (x[2t], x[2t+1]) -> (x0 c - x1 s, x0 s + x1 c)
//   NeoX:        (x[t], x[t+rd/2]) -> (x0 c - x1 s, x0 s + x1 c)
#include "fa_common.h"
#include "fa_rope.h"

namespace fa {

int launch_fwd(const KArgs& a, hipStream_t stream);
bool decode_applicable(const fa_params& p);
bool decode_takes(const fa_params& p);
size_t decode_split_workspace_bytes(const fa_params& p);
int launch_decode_splitkv(const KArgs& a, void* ws, hipStream_t stream);

// 8 x 16-bit -> 8 x fp8-e4m3 (OCP), value / descale, saturating at +-448
template <typename T>
__device__ __forceinline__ u32x2 to_fp8x8(const u32x4& x, float inv_descale) {
    using E = Elem<T>;
    u32x2 r = {0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a0 = fminf(fmaxf(E::lo(x[i]) * inv_descale, -448.f), 448.f);
        const float a1 = fminf(fmaxf(E::hi(x[i]) * inv_descale, -448.f), 448.f);
        if (i == 0) r[0] = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, r[0], false);
        if (i == 1) r[0] = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, r[0], true);
        if (i == 2) r[1] = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, r[1], false);
        if (i == 3) r[1] = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, r[1], true);
    }
    return r;
}

// One thread per 16-byte chunk of one new (b, r, hk) row; K and V.
template <typename T, bool KV8>
__global__ void __launch_bounds__(256) kv_append_kernel(const KArgs a) {
    const fa_params& p = a.p;
    const int cpr = valid_cols(p) / 8;
    const int64_t total = (int64_t)p.batch * p.seqlen_new * p.nheads_k * cpr;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cc = idx % cpr;
    int64_t t = idx / cpr;
    const int hk = t % p.nheads_k; t /= p.nheads_k;
    const int r = t % p.seqlen_new;
    const int b = t / p.seqlen_new;
    const int L = p.cache_seqlens ? p.cache_seqlens[b] : 0;
    const int lp = p.cache_leftpad ? p.cache_leftpad[b] : 0;
    const int pos = L + lp + r;
    if (pos < 0 || pos >= p.seqlen_k) return;           // beyond the cache capacity (dense: S_max, paged: table columns x page)
    const int d_base = cc * 8;
    const uint16_t* kn = reinterpret_cast<const uint16_t*>(p.k_new) + (int64_t)b * p.knew_batch_stride +
                         (int64_t)r * p.knew_row_stride + (int64_t)hk * p.knew_head_stride;
    const uint16_t* vn = reinterpret_cast<const uint16_t*>(p.v_new) + (int64_t)b * p.vnew_batch_stride +
                         (int64_t)r * p.vnew_row_stride + (int64_t)hk * p.vnew_head_stride;
    u32x4 kx = *reinterpret_cast<const u32x4*>(kn + d_base);
    const u32x4 vx = *reinterpret_cast<const u32x4*>(vn + d_base);
    if (p.rotary_dim > 0 && d_base < p.rotary_dim && pos < p.seqlen_ro) {
        const int half = p.rotary_dim >> 1;
        u32x4 kp = kx;
        if (!p.rotary_interleaved) {
            const int pd = d_base < half ? d_base + half : d_base - half;
            kp = *reinterpret_cast<const u32x4*>(kn + pd);
        }
        const uint16_t* cosp = reinterpret_cast<const uint16_t*>(p.rotary_cos) + (int64_t)pos * half;
        const uint16_t* sinp = reinterpret_cast<const uint16_t*>(p.rotary_sin) + (int64_t)pos * half;
        rope_chunk<T>(kx, kp, cosp, sinp, d_base, p.rotary_dim, p.rotary_interleaved != 0);
    }
    int64_t koff, voff;
    if (p.block_table) {
        const int pg = pos / p.page_block_size, pr = pos - pg * p.page_block_size;
        const int64_t phys = p.block_table[(int64_t)b * p.block_table_batch_stride + pg];
        koff = phys * p.k_batch_stride + (int64_t)pr * p.k_row_stride;
        voff = phys * p.v_batch_stride + (int64_t)pr * p.v_row_stride;
    } else {
        const int cb = p.cache_batch_idx ? p.cache_batch_idx[b] : b;
        koff = (int64_t)cb * p.k_batch_stride + (int64_t)pos * p.k_row_stride;
        voff = (int64_t)cb * p.v_batch_stride + (int64_t)pos * p.v_row_stride;
    }
    koff += (int64_t)hk * p.k_head_stride + d_base;
    voff += (int64_t)hk * p.v_head_stride + d_base;
    if (KV8) {
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint8_t*>(const_cast<void*>(p.k)) + koff) = to_fp8x8<T>(kx, 1.0f / p.k_descale);
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint8_t*>(const_cast<void*>(p.v)) + voff) = to_fp8x8<T>(vx, 1.0f / p.v_descale);
    } else {
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(const_cast<void*>(p.k)) + koff) = kx;
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(const_cast<void*>(p.v)) + voff) = vx;
    }
}

size_t decode_workspace_bytes(const fa_params& p) {
    if (decode_takes(p)) return decode_split_workspace_bytes(p);
    return 0;                       // the general path rotates Q inside fa_fwd_kernel (KArgs::rope_q): no scratch
}

int launch_decode(const KArgs& a_in, hipStream_t stream) {
    KArgs a = a_in;
    fa_params& p = a.p;
    const bool bf = p.dtype == FA_BF16;
    const bool kv8 = p.kv_dtype == FA_FP8_E4M3;
    if (kv8 && (!(p.head_dim == 64 || p.head_dim == 128) || p.head_dim_v != 0)) return -2;
    // few query positions per sequence: the decode kernels (one K / V stream per kv-head and 32 packed rows, split-KV); more -
    // chunked prefill, long speculative blocks -: fa_fwd_kernel on the cache, which dequantises an fp8 tile once per 128
    // query rows.  decode_takes() (fa_decode.hip) holds the rule and the measurements behind it.
    const bool fast = decode_takes(p);
    if (p.k_new) {
        const int64_t total = (int64_t)p.batch * p.seqlen_new * p.nheads_k * (valid_cols(p) / 8);
        const int grid = (int)((total + 255) / 256);
        if (kv8) {
            if (bf) hipLaunchKernelGGL((kv_append_kernel<bf16_tag, true>), dim3(grid), dim3(256), 0, stream, a);
            else    hipLaunchKernelGGL((kv_append_kernel<fp16_tag, true>), dim3(grid), dim3(256), 0, stream, a);
        } else {
            if (bf) hipLaunchKernelGGL((kv_append_kernel<bf16_tag, false>), dim3(grid), dim3(256), 0, stream, a);
            else    hipLaunchKernelGGL((kv_append_kernel<fp16_tag, false>), dim3(grid), dim3(256), 0, stream, a);
        }
    }
    if (fast) {
        const size_t need = decode_split_workspace_bytes(p);
        if (need > 0 && (!p.workspace || p.workspace_bytes < need)) return -1;
        return launch_decode_splitkv(a, p.workspace, stream);
    }
    a.rope_q = p.rotary_dim > 0 ? 1 : 0;
    return launch_fwd(a, stream);
}

}  // namespace fa
