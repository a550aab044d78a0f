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

namespace fa {

int launch_fwd(const KArgs& a, hipStream_t stream);

template <typename T>
__device__ __forceinline__ void rope_chunk(u32x4& x, const u32x4& xp, const uint16_t* cosp, const uint16_t* sinp,
                                           int d_base, int rd, bool interleaved) {
    // x: 8 elements at d_base..d_base+7; xp: partner chunk (NeoX only).
    using E = Elem<T>;
    if (d_base >= rd) return;
    const int half = rd >> 1;
    if (interleaved) {
        const u32x2 cw = *reinterpret_cast<const u32x2*>(cosp + d_base / 2);   // 4 cos values
        const u32x2 sw = *reinterpret_cast<const u32x2*>(sinp + d_base / 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x0 = E::lo(x[i]), x1 = E::hi(x[i]);
            const uint32_t cword = cw[i >> 1], sword = sw[i >> 1];
            const float c = (i & 1) ? E::hi(cword) : E::lo(cword);
            const float s = (i & 1) ? E::hi(sword) : E::lo(sword);
            x[i] = E::pack2(fmaf(x0, c, -x1 * s), fmaf(x0, s, x1 * c));
        }
    } else {
        const bool first = d_base < half;
        const int t0 = first ? d_base : d_base - half;
        const u32x4 cw = *reinterpret_cast<const u32x4*>(cosp + t0);           // 8 cos values
        const u32x4 sw = *reinterpret_cast<const u32x4*>(sinp + t0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a0 = E::lo(x[i]), a1 = E::hi(x[i]);
            const float b0 = E::lo(xp[i]), b1 = E::hi(xp[i]);
            const float c0 = E::lo(cw[i]), c1 = E::hi(cw[i]);
            const float s0 = E::lo(sw[i]), s1 = E::hi(sw[i]);
            // first half: y = x0 c - x1 s (x0 = own, x1 = partner); second: y = x0 s + x1 c (x0 = partner, x1 = own)
            const float y0 = first ? fmaf(a0, c0, -b0 * s0) : fmaf(b0, s0, a0 * c0);
            const float y1 = first ? fmaf(a1, c1, -b1 * s1) : fmaf(b1, s1, a1 * c1);
            x[i] = E::pack2(y0, y1);
        }
    }
}

// One thread per 16-byte chunk of one new (b, r, hk) row; K and V.
template <typename T>
__global__ void __launch_bounds__(256) kv_append_kernel(const KArgs a) {
    const fa_params& p = a.p;
    const int cpr = p.head_dim / 8;
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
    const int d_base = cc * 8;
    const uint16_t* kn = reinterpret_cast<const uint16_t*>(p.k_new) + (int64_t)b * p.knew_batch_stride +
                         (int64_t)r * p.knew_row_stride + (int64_t)hk * p.knew_head_stride;
    const uint16_t* vn = reinterpret_cast<const uint16_t*>(p.v_new) + (int64_t)b * p.vnew_batch_stride +
                         (int64_t)r * p.vnew_row_stride + (int64_t)hk * p.vnew_head_stride;
    u32x4 kx = *reinterpret_cast<const u32x4*>(kn + d_base);
    const u32x4 vx = *reinterpret_cast<const u32x4*>(vn + d_base);
    if (p.rotary_dim > 0 && d_base < p.rotary_dim) {
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
    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(const_cast<void*>(p.k)) + koff) = kx;
    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(const_cast<void*>(p.v)) + voff) = vx;
}

// Rotate q [B, Tq, Hq, D] into a contiguous workspace of the same shape.
template <typename T>
__global__ void __launch_bounds__(256) q_rope_kernel(const KArgs a, uint16_t* out, int local) {
    const fa_params& p = a.p;
    const int cpr = p.head_dim / 8;
    const int64_t total = (int64_t)p.batch * p.seqlen_q * p.nheads_q * cpr;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cc = idx % cpr;
    int64_t t = idx / cpr;
    const int h = t % p.nheads_q; t /= p.nheads_q;
    const int i = t % p.seqlen_q;
    const int b = t / p.seqlen_q;
    const int L = p.cache_seqlens ? p.cache_seqlens[b] : 0;
    const int lp = p.cache_leftpad ? p.cache_leftpad[b] : 0;
    const int pos = L + lp + (local ? i : 0);          // include/rotary.h:177,201-202
    const int d_base = cc * 8;
    const uint16_t* qr = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.q_batch_stride +
                         (int64_t)i * p.q_row_stride + (int64_t)h * p.q_head_stride;
    u32x4 x = *reinterpret_cast<const u32x4*>(qr + d_base);
    if (d_base < p.rotary_dim) {
        const int half = p.rotary_dim >> 1;
        u32x4 xp = x;
        if (!p.rotary_interleaved) {
            const int pd = d_base < half ? d_base + half : d_base - half;
            xp = *reinterpret_cast<const u32x4*>(qr + pd);
        }
        const uint16_t* cosp = reinterpret_cast<const uint16_t*>(p.rotary_cos) + (int64_t)pos * half;
        const uint16_t* sinp = reinterpret_cast<const uint16_t*>(p.rotary_sin) + (int64_t)pos * half;
        rope_chunk<T>(x, xp, cosp, sinp, d_base, p.rotary_dim, p.rotary_interleaved != 0);
    }
    *reinterpret_cast<u32x4*>(out + idx * 8) = x;
}

size_t decode_workspace_bytes(const fa_params& p) {
    size_t bytes = 0;
    if (p.rotary_dim > 0) bytes += (size_t)p.batch * p.seqlen_q * p.nheads_q * p.head_dim * 2;
    return bytes;
}

int launch_decode(const KArgs& a_in, hipStream_t stream) {
    KArgs a = a_in;
    fa_params& p = a.p;
    if (p.kv_dtype == FA_FP8_E4M3) return -2;
    const bool bf = p.dtype == FA_BF16;
    if (p.k_new) {
        const int64_t total = (int64_t)p.batch * p.seqlen_new * p.nheads_k * (p.head_dim / 8);
        const int grid = (int)((total + 255) / 256);
        if (bf) hipLaunchKernelGGL(kv_append_kernel<bf16_tag>, dim3(grid), dim3(256), 0, stream, a);
        else    hipLaunchKernelGGL(kv_append_kernel<fp16_tag>, dim3(grid), dim3(256), 0, stream, a);
    }
    if (p.rotary_dim > 0) {
        const size_t need = decode_workspace_bytes(p);
        if (!p.workspace || p.workspace_bytes < need) return -1;
        uint16_t* qrot = reinterpret_cast<uint16_t*>(p.workspace);
        const int local = (p.is_causal || p.window_left >= 0 || p.window_right >= 0) ? 1 : 0;
        const int64_t total = (int64_t)p.batch * p.seqlen_q * p.nheads_q * (p.head_dim / 8);
        const int grid = (int)((total + 255) / 256);
        if (bf) hipLaunchKernelGGL(q_rope_kernel<bf16_tag>, dim3(grid), dim3(256), 0, stream, a, qrot, local);
        else    hipLaunchKernelGGL(q_rope_kernel<fp16_tag>, dim3(grid), dim3(256), 0, stream, a, qrot, local);
        p.q = qrot;
        p.q_head_stride = p.head_dim;
        p.q_row_stride = (int64_t)p.nheads_q * p.head_dim;
        p.q_batch_stride = (int64_t)p.seqlen_q * p.q_row_stride;
    }
    return launch_fwd(a, stream);
}

}  // namespace fa
