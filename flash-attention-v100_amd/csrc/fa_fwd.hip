// fa_fwd.hip - fused attention forward for gfx950 (dense / varlen / paged / kv-cache prefill).
//
// Replaces the reference's three forward kernels
//   kernel/fused_mha_forward.cu:25-224, kernel/fused_mha_forward_varlen.cu:25-275,
//   kernel/fused_mha_forward_kvcache.cu:24-295 (attention part)
// with ONE CDNA4 design:
//   * workgroup = 4 waves, 128 query rows (32 per wave), KV tile = 64 keys;
//   * S^T = K Q^T on v_mfma_f32_32x32x16 ("swapped" so a lane owns ONE query row: the
//     online-softmax max/sum are lane-local + one cross-half exchange, no LDS for S/P);
//   * the S^T accumulator registers are packed to 16-bit and fed straight back as the B
//     operand of O^T = V^T P^T (the key index is only a contraction index, so the MFMA
//     C-layout key order is used as-is: no permute, no LDS round trip);
//   * V^T comes from LDS through ds_read_b64_tr_b16 (hardware 4x4 transpose), K through
//     XOR-swizzled ds_read_b128; both tiles are double buffered, global->register loads
//     of tile t+1 are issued before the MFMAs of tile t and written to LDS after them
//     (one barrier per tile);
//   * running max / sum and the O accumulator never leave registers.
#include "fa_common.h"

namespace fa {

constexpr int FWD_BM = 128;
constexpr int FWD_BN = 64;
constexpr int FWD_THREADS = 256;

template <int D> struct FwdSmem {
    static constexpr int TILE = FWD_BN * D * 2;        // bytes of one K (or V) tile
    static constexpr int STAGE = 2 * TILE;             // K + V
    static constexpr int TOTAL = 2 * STAGE;            // double buffered
};

template <typename T, int D, bool BIAS, bool PAGED>
__global__ void __launch_bounds__(FWD_THREADS, 2) fa_fwd_kernel(const KArgs a) {
    using E = Elem<T>;
    constexpr int KSTEPS = D / 16;
    constexpr int DBLKS = D / 32;
    constexpr int CPR = D / 8;                          // 16-B chunks per row
    constexpr int CHUNKS = FWD_BN * CPR / FWD_THREADS;  // chunks per thread per tile
    constexpr int TILE = FwdSmem<D>::TILE;
    constexpr int STAGE = FwdSmem<D>::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const fa_params& p = a.p;
    const WorkItem w = decode_work(blockIdx.x, p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
    if (!w.valid) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int g = lane >> 5;

    // ---- per-sequence geometry --------------------------------------------------------
    int seqlen_q = p.seqlen_q, seqlen_k = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;                    // row offsets (varlen packing / leftpad)
    int kv_b = w.b;
    if (p.cu_seqlens_q) {
        q_row0 = p.cu_seqlens_q[w.b];
        seqlen_q = p.cu_seqlens_q[w.b + 1] - (int)q_row0;
    }
    if (p.cu_seqlens_k) {
        const int k0 = p.cu_seqlens_k[w.b];
        seqlen_k = p.cu_seqlens_k[w.b + 1] - k0;
        if (!PAGED) k_row0 = k0;
    }
    if (a.seqlens_k) {
        const int su = a.seqlens_k[w.b] + a.seqlen_k_add;
        if (p.cu_seqlens_k) seqlen_k = su > 0 ? (su < seqlen_k ? su : seqlen_k) : 0;   // seqused_k
        else seqlen_k = su;                                                             // kv cache
    }
    if (a.kv_batch_idx) kv_b = a.kv_batch_idx[w.b];
    if (a.leftpad_k) k_row0 += a.leftpad_k[w.b];

    const int m_block = w.qb * FWD_BM;
    if (m_block >= seqlen_q) return;
    const int off = seqlen_k - seqlen_q;               // bottom-right alignment
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;   // causal == window_right 0 (include/mat_mul.h:92,103)
    // key-tile range for this 128-row block
    int n_min = 0, n_max = (seqlen_k + FWD_BN - 1) / FWD_BN;
    {
        const int m_last = (m_block + FWD_BM < seqlen_q ? m_block + FWD_BM : seqlen_q) - 1;
        if (wr >= 0) {
            const int kmax = m_last + off + wr;
            const int t = kmax < 0 ? 0 : kmax / FWD_BN + 1;
            n_max = t < n_max ? t : n_max;
        }
        if (wl >= 0) {
            const int kmin = m_block + off - wl;
            if (kmin > 0) n_min = kmin / FWD_BN;
        }
    }

    const int wave_row0 = m_block + wave * 32;
    const int my_row = wave_row0 + l31;                // this lane's query row
    // visible keys of my row: lo <= j <= hi
    int lo = 0, hi = seqlen_k - 1;
    if (wr >= 0) { const int h2 = my_row + off + wr; hi = h2 < hi ? h2 : hi; }
    if (wl >= 0) { const int l2 = my_row + off - wl; lo = l2 > lo ? l2 : lo; }
    // wave-uniform bounds for tile skipping / mask elision
    const int wrow_last = wave_row0 + 31;
    int w_hi_min = seqlen_k - 1, w_hi_max = seqlen_k - 1, w_lo_max = 0;
    if (wr >= 0) {
        const int h0 = wave_row0 + off + wr, h1 = wrow_last + off + wr;
        w_hi_min = h0 < w_hi_min ? h0 : w_hi_min;
        w_hi_max = h1 < w_hi_max ? h1 : w_hi_max;
    }
    if (wl >= 0) { const int l1 = wrow_last + off - wl; w_lo_max = l1 > 0 ? l1 : 0; }
    const int w_lo_min = (wl >= 0 && wave_row0 + off - wl > 0) ? wave_row0 + off - wl : 0;

    // ---- pointers -------------------------------------------------------------------------
    const uint16_t* qp = reinterpret_cast<const uint16_t*>(p.q) + (p.cu_seqlens_q ? 0 : (int64_t)w.b * p.q_batch_stride)
                         + q_row0 * p.q_row_stride + (int64_t)w.h * p.q_head_stride;
    const uint16_t* kp = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)w.hk * p.k_head_stride;
    const uint16_t* vp = reinterpret_cast<const uint16_t*>(p.v) + (int64_t)w.hk * p.v_head_stride;
    if (!PAGED) {
        const int64_t kb_off = (p.cu_seqlens_k ? 0 : (int64_t)kv_b * p.k_batch_stride);
        const int64_t vb_off = (p.cu_seqlens_k ? 0 : (int64_t)kv_b * p.v_batch_stride);
        kp += kb_off + k_row0 * p.k_row_stride;
        vp += vb_off + k_row0 * p.v_row_stride;
    }
    const int32_t* btab = PAGED ? p.block_table + (int64_t)w.b * p.block_table_batch_stride : nullptr;

    float slope = 0.f;                                 // ALiBi slope (natural-log score units)
    if (BIAS && p.alibi_slopes) slope = p.alibi_slopes[w.b * p.alibi_batch_stride + w.h];

    // ---- Q fragments: B operand of S^T = K Q^T, lane holds Q[my_row][16ks + 8g .. +7] --------
    u32x4 qf[KSTEPS];
    {
        const bool ok = my_row < seqlen_q;
        const uint16_t* qrow = qp + (int64_t)my_row * p.q_row_stride + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            u32x4 z = {0, 0, 0, 0};
            qf[ks] = ok ? *reinterpret_cast<const u32x4*>(qrow + 16 * ks) : z;
        }
    }

    // ---- staging: global -> registers -> LDS ------------------------------------------------
    u32x4 kreg[CHUNKS], vreg[CHUNKS];
    auto load_tile = [&](int nb) {
        const int n0 = nb * FWD_BN;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const int c = tid + i * FWD_THREADS;
            const int row = c / CPR, cc = c % CPR;
            const int j = n0 + row;
            u32x4 z = {0, 0, 0, 0};
            kreg[i] = z; vreg[i] = z;
            if (j < seqlen_k) {
                if (PAGED) {
                    const int pos = j + (int)k_row0;
                    const int pg = pos / p.page_block_size;
                    const int pr = pos - pg * p.page_block_size;
                    const int64_t phys = btab[pg];
                    kreg[i] = *reinterpret_cast<const u32x4*>(kp + phys * p.k_batch_stride + (int64_t)pr * p.k_row_stride + cc * 8);
                    vreg[i] = *reinterpret_cast<const u32x4*>(vp + phys * p.v_batch_stride + (int64_t)pr * p.v_row_stride + cc * 8);
                } else {
                    kreg[i] = *reinterpret_cast<const u32x4*>(kp + (int64_t)j * p.k_row_stride + cc * 8);
                    vreg[i] = *reinterpret_cast<const u32x4*>(vp + (int64_t)j * p.v_row_stride + cc * 8);
                }
            }
        }
    };
    auto store_tile = [&](int stage) {
        char* ks = smem + stage * STAGE;
        char* vs = ks + TILE;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const int c = tid + i * FWD_THREADS;
            const int row = c / CPR, cc = c % CPR;
            lds_write_b128(ks + swz_row_off<D>(row, cc * 16), kreg[i]);
            lds_write_b128(vs + vtile_off<D>(row, cc * 8), vreg[i]);
        }
    };

    // ---- accumulators -------------------------------------------------------------------------
    f32x16 oacc[DBLKS];
#pragma unroll
    for (int d = 0; d < DBLKS; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY;     // running max, log2 domain (scaled)
    float l_run = 0.f;           // this lane's partial row sum (its 32 keys per tile)

    // lane-constant LDS read offsets
    const int k_lane_row = l31;                                     // + 32 kb
    const int v_lane_off = (g * DBLKS << 8) + (((lane & 15) >> 2) << 6) + (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);

    if (n_min < n_max) {
        load_tile(n_min);
        store_tile(0);
    }
    __syncthreads();

    for (int nb = n_min; nb < n_max; ++nb) {
        const int stage = (nb - n_min) & 1;
        const bool has_next = nb + 1 < n_max;
        if (has_next) load_tile(nb + 1);

        const int n0 = nb * FWD_BN;
        // wave-uniform: does this wave see anything in this tile?
        const bool wave_active = (n0 <= w_hi_max) && (n0 + FWD_BN - 1 >= w_lo_min);
        if (wave_active) {
            const char* ks_base = smem + stage * STAGE;
            const char* vs_base = ks_base + TILE;
            // ---- S^T = K Q^T : sacc[kb][r] = S[my_row][n0 + 32 kb + row(r, g)] ----
            f32x16 sacc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[0][r] = 0.f; sacc[1][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const u32x4 k0 = lds_read_b128(ks_base + swz_row_off<D>(k_lane_row, 32 * ks + 16 * g));
                const u32x4 k1 = lds_read_b128(ks_base + swz_row_off<D>(k_lane_row + 32, 32 * ks + 16 * g));
                sacc[0] = E::mfma(k0, qf[ks], sacc[0]);
                sacc[1] = E::mfma(k1, qf[ks], sacc[1]);
            }
            // ---- bias / softcap (rare variants), then masking on edge tiles ----
            if (BIAS) {
                const float cap = p.softcap;
                const float rcap = cap > 0.f ? 1.0f / cap : 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = n0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                        float s = sacc[kb][r] * p.softmax_scale;
                        const int dist = my_row + off - j;
                        s = fmaf(-slope, fabsf((float)dist), s);
                        if (cap > 0.f) s = cap * fast_tanh(s * rcap);
                        sacc[kb][r] = s * kLog2e;
                    }
            }
            const bool need_mask = (n0 + FWD_BN - 1 > w_hi_min) || (n0 < w_lo_max);
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = n0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                        if (j < lo || j > hi) sacc[kb][r] = -INFINITY;
                    }
            }
            // ---- online softmax (log2 domain) ----
            float mx = sacc[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[1][r]);
            mx = fmaxf(mx, shfl_xor32(mx));
            const float c = BIAS ? 1.0f : a.scale_log2e;
            const float m_new = fmaxf(m_run, mx * c);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - m_use);
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = fast_exp2(fmaf(sacc[kb][r], c, -m_use));
                    sacc[kb][r] = e;
                    psum += e;
                }
            l_run = fmaf(l_run, alpha, psum);
#pragma unroll
            for (int d = 0; d < DBLKS; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;

            // ---- O^T += V^T P^T : k-step t covers C-layout regs 8 (t&1) .. +7 of sacc[t>>1] ----
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kb = t >> 1, ks2 = t & 1;
                u32x4 pf;
                pf[0] = E::pack2(sacc[kb][8 * ks2 + 0], sacc[kb][8 * ks2 + 1]);
                pf[1] = E::pack2(sacc[kb][8 * ks2 + 2], sacc[kb][8 * ks2 + 3]);
                pf[2] = E::pack2(sacc[kb][8 * ks2 + 4], sacc[kb][8 * ks2 + 5]);
                pf[3] = E::pack2(sacc[kb][8 * ks2 + 6], sacc[kb][8 * ks2 + 7]);
#pragma unroll
                for (int d = 0; d < DBLKS; ++d) {
                    // rows kb*32 + 16 ks2 + 8 hf + 4 g + (0..3) -> 4-row block index kb*8 + 4 ks2 + 2 hf + g
                    const int blk0 = (kb * 8 + 4 * ks2) * DBLKS + d;
                    const u32x2 v0 = lds_read_tr16(vs_base + v_lane_off + (blk0 << 8));
                    const u32x2 v1 = lds_read_tr16(vs_base + v_lane_off + ((blk0 + 2 * DBLKS) << 8));
                    u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
                    oacc[d] = E::mfma(vf, pf, oacc[d]);
                }
            }
        }
        if (has_next) store_tile(stage ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O / l, LSE ---------------------------------------------------------------
    const float l_tot = l_run + shfl_xor32(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (my_row < seqlen_q) {
        uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + (p.cu_seqlens_q ? 0 : (int64_t)w.b * p.o_batch_stride)
                       + (q_row0 + my_row) * p.o_row_stride + (int64_t)w.h * p.o_head_stride;
#pragma unroll
        for (int d = 0; d < DBLKS; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u32x2 o2;
                o2[0] = E::pack2(oacc[d][4 * rq + 0] * inv, oacc[d][4 * rq + 1] * inv);
                o2[1] = E::pack2(oacc[d][4 * rq + 2] * inv, oacc[d][4 * rq + 3] * inv);
                *reinterpret_cast<u32x2*>(op + d * 32 + 8 * rq + 4 * g) = o2;
            }
        if (g == 0) {
            const float lse = l_tot > 0.f ? (m_run + fast_log2(l_tot)) * kLn2 : -INFINITY;
            p.lse[(int64_t)w.b * p.lse_batch_stride + (int64_t)w.h * p.lse_head_stride + q_row0 + my_row] = lse;
        }
    }
}

// ---- host launcher ---------------------------------------------------------------------------
template <typename T, int D>
static int launch_fwd_td(const KArgs& a, bool paged, hipStream_t stream) {
    const int grid = work_grid(a.p.batch, a.p.nheads_q, a.p.nheads_k, a.n_qblocks);
    const size_t smem = FwdSmem<D>::TOTAL;
    if (grid == 0) return 0;
#define FA_LAUNCH(BIAS, PAGED)                                                                  \
    do {                                                                                        \
        auto kern = fa_fwd_kernel<T, D, BIAS, PAGED>;                                           \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);             \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(FWD_THREADS), smem, stream, a);               \
    } while (0)
    if (a.has_bias) { if (paged) FA_LAUNCH(true, true); else FA_LAUNCH(true, false); }
    else            { if (paged) FA_LAUNCH(false, true); else FA_LAUNCH(false, false); }
#undef FA_LAUNCH
    return 0;
}

int launch_fwd(const KArgs& a, hipStream_t stream) {
    const bool paged = a.p.block_table != nullptr;
    const bool bf = a.p.dtype == FA_BF16;
    switch (a.p.head_dim) {
        case 64:  return bf ? launch_fwd_td<bf16_tag, 64>(a, paged, stream) : launch_fwd_td<fp16_tag, 64>(a, paged, stream);
        case 128: return bf ? launch_fwd_td<bf16_tag, 128>(a, paged, stream) : launch_fwd_td<fp16_tag, 128>(a, paged, stream);
        default:  return -2;
    }
}

}  // namespace fa
