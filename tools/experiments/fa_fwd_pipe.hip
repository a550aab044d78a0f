// fa_fwd_pipe.hip - software-pipelined forward kernel: the fast path for plain dense / varlen
// attention (no ALiBi / softcap / dropout / paging; fa_fwd.hip keeps the general kernel).
//
// Why: in fa_fwd_kernel the per-wave stream is QK^T (MFMA) -> softmax (VALU) -> PV (MFMA), and an
// in-order wave only overlaps MFMA and VALU work that ALTERNATES in program order.  PMC on that
// kernel: MFMA busy ~50 %, VALU busy ~48 %, i.e. the two pipes take turns.  Here each wave carries
// two 32-key score tiles and every sub-step issues
//     X(j+1): S(j+1) = K(j+1) Q^T      8 MFMA      } interleaved by hand (sched_barrier fences):
//     Y(j)  : max / exp2 / pack of S(j)  ~60 VALU    } the softmax of tile j runs under the MFMAs
//     Z(j)  : O += V(j)^T P(j)^T         8 MFMA      }
// so the matrix pipe always has independent work while the VALU chain of tile j resolves.
//
// Geometry: workgroup = 8 waves (2 per SIMD) x 32 query rows = 256 rows; KV tiles of 64 keys are
// LDS-DMA'd (buffer_load ... lds) into a 3-deep K ring and a 2-deep V ring (K runs one tile ahead
// because X reads the NEXT sub-tile); one barrier per 64-key tile.  Same MFMA data plan as
// fa_fwd.hip (swapped S^T, C-layout reuse as the PV B operand, ds_read_b64_tr_b16 for V^T).
#include <type_traits>
#include "fa_common.h"

namespace fa {

constexpr int P3_BM = 256;
constexpr int P3_BN = 64;
constexpr int P3_THREADS = 512;
constexpr float P3_RESCALE_THR = 8.0f;                 // log2 units

template <int D> struct P3Smem {
    static constexpr int TILE = P3_BN * D * 2;         // one K (or V) tile
    static constexpr int KRING = 3, VRING = 2;
    static constexpr int V_OFF = KRING * TILE;
    static constexpr int TOTAL = (KRING + VRING) * TILE;
};

#define FA_FENCE() __builtin_amdgcn_sched_barrier(0)

template <typename T, int D>
__global__ void __launch_bounds__(P3_THREADS, 2) fa_fwd_p3_kernel(const KArgs a) {
    using E = Elem<T>;
    constexpr int KSTEPS = D / 16;                      // MFMAs of one X stage
    constexpr int DBLKS = D / 32;
    constexpr int CPR = D / 8;
    constexpr int ROWS_PI = 64 / CPR;                   // rows per DMA instruction
    constexpr int CHUNKS = P3_BN * CPR / P3_THREADS;    // DMA instructions per wave per tile (K or V)
    constexpr int TILE = P3Smem<D>::TILE;
    constexpr int V_OFF = P3Smem<D>::V_OFF;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const fa_params& p = a.p;
    const WorkItem w = decode_work(blockIdx.x, p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
    if (!w.valid) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;

    int seqlen_q = p.seqlen_q, seqlen_k = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    if (p.cu_seqlens_q) { q_row0 = p.cu_seqlens_q[w.b]; seqlen_q = p.cu_seqlens_q[w.b + 1] - (int)q_row0; }
    if (p.cu_seqlens_k) { const int k0 = p.cu_seqlens_k[w.b]; seqlen_k = p.cu_seqlens_k[w.b + 1] - k0; k_row0 = k0; }
    if (a.seqlens_k) { const int su = a.seqlens_k[w.b]; seqlen_k = su > 0 ? (su < seqlen_k ? su : seqlen_k) : 0; }
    const int off = seqlen_k - seqlen_q;
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;
    const float c = a.scale_log2e;

    const uint16_t* qp = reinterpret_cast<const uint16_t*>(p.q) + (p.cu_seqlens_q ? 0 : (int64_t)w.b * p.q_batch_stride)
                         + q_row0 * p.q_row_stride + (int64_t)w.h * p.q_head_stride;
    const uint16_t* kp = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)w.hk * p.k_head_stride +
                         (p.cu_seqlens_k ? 0 : (int64_t)w.b * p.k_batch_stride) + k_row0 * p.k_row_stride;
    const uint16_t* vp = reinterpret_cast<const uint16_t*>(p.v) + (int64_t)w.hk * p.v_head_stride +
                         (p.cu_seqlens_k ? 0 : (int64_t)w.b * p.v_batch_stride) + k_row0 * p.v_row_stride;
    const __amdgpu_buffer_rsrc_t k_rsrc = make_rsrc(kp, p.k_row_stride, seqlen_k, D);
    const __amdgpu_buffer_rsrc_t v_rsrc = make_rsrc(vp, p.v_row_stride, seqlen_k, D);
    const uint32_t k_tile_bytes = (uint32_t)(P3_BN * p.k_row_stride * 2);
    const uint32_t v_tile_bytes = (uint32_t)(P3_BN * p.v_row_stride * 2);

    // DMA geometry: instruction inst = wave*CHUNKS + i covers rows inst*ROWS_PI ..; lane -> (row, slot)
    uint32_t k_voff[CHUNKS], v_voff[CHUNKS];
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int inst = wave * CHUNKS + i;
        const int row = inst * ROWS_PI + lane / CPR;
        const int slot = lane % CPR;
        k_voff[i] = (uint32_t)(row * p.k_row_stride * 2 + (swz_row_off<D>(row, slot * 16) - row * D * 2));
        v_voff[i] = (uint32_t)(row * p.v_row_stride * 2 + (swzt_row_off<D>(row, slot * 16) - row * D * 2));
    }
    auto dma_k = [&](int nb, int ring) {
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i)
            buf_load_lds_b128(k_rsrc, smem + ring * TILE + (wave * CHUNKS + i) * 1024, k_voff[i], (uint32_t)nb * k_tile_bytes);
    };
    auto dma_v = [&](int nb, int ring) {
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i)
            buf_load_lds_b128(v_rsrc, smem + V_OFF + ring * TILE + (wave * CHUNKS + i) * 1024, v_voff[i], (uint32_t)nb * v_tile_bytes);
    };

    // lane-constant read offsets
    int k_rd[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) k_rd[ks] = swz_row_off<D>(l31, 32 * ks + 16 * g);
    const int v_rr = (lane & 15) >> 2;
    const int v_cb = (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);
    int v_rd[2][DBLKS];                                  // [+0 / +8 rows][d-block]; k-step = +16 rows (immediate)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int d = 0; d < DBLKS; ++d)
            v_rd[hf][d] = swzt_row_off<D>(8 * hf + 4 * g + v_rr, d * 64 + v_cb);

    const int n_pass = (a.pair_qblocks && (a.n_qblocks_total - 1 - w.qb) != w.qb) ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
    const int qb_cur = pass == 0 ? w.qb : a.n_qblocks_total - 1 - w.qb;
    const int m_block = qb_cur * P3_BM;
    if (m_block >= seqlen_q) continue;
    int n_min = 0, n_max = (seqlen_k + P3_BN - 1) / P3_BN;
    {
        const int m_last = (m_block + P3_BM < seqlen_q ? m_block + P3_BM : seqlen_q) - 1;
        if (wr >= 0) {
            const int kmax = m_last + off + wr;
            const int t = kmax < 0 ? 0 : kmax / P3_BN + 1;
            n_max = t < n_max ? t : n_max;
        }
        if (wl >= 0) { const int kmin = m_block + off - wl; if (kmin > 0) n_min = kmin / P3_BN; }
    }
    const int wave_row0 = m_block + wave * 32;
    const int my_row = wave_row0 + l31;
    int lo = 0, hi = seqlen_k - 1;
    if (wr >= 0) { const int h2 = my_row + off + wr; hi = h2 < hi ? h2 : hi; }
    if (wl >= 0) { const int l2 = my_row + off - wl; lo = l2 > lo ? l2 : lo; }
    // wave-uniform visibility bounds (in keys) -> 32-key sub-tile range this wave has to touch
    const int wrow_last = wave_row0 + 31;
    int w_hi_min = seqlen_k - 1, w_hi_max = seqlen_k - 1, w_lo_max = 0, w_lo_min = 0;
    if (wr >= 0) {
        const int h0 = wave_row0 + off + wr, h1 = wrow_last + off + wr;
        w_hi_min = h0 < w_hi_min ? h0 : w_hi_min;
        w_hi_max = h1 < w_hi_max ? h1 : w_hi_max;
    }
    if (wl >= 0) {
        const int l1 = wrow_last + off - wl; w_lo_max = l1 > 0 ? l1 : 0;
        const int l0 = wave_row0 + off - wl; w_lo_min = l0 > 0 ? l0 : 0;
    }
    const bool wave_rows = wave_row0 < seqlen_q;
    const int j_first = w_lo_min / 32;                                   // first sub-tile with a visible key
    const int j_last = (wave_rows && w_hi_max >= 0) ? w_hi_max / 32 : -1;  // last one

    // ---- Q fragments (B operand of S^T = K Q^T) ----
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

    f32x16 oacc[DBLKS];
#pragma unroll
    for (int d = 0; d < DBLKS; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ------------------------------------------------------------------------------------------
    // stages
    // ------------------------------------------------------------------------------------------
    // X: scores of one 32-key sub-tile; kbase = LDS address of its first key row
    auto stage_x = [&](const char* kbase, f32x16& s) {
        f32x16 s2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks += 2) {
            s = E::mfma(lds_read_b128(kbase + k_rd[ks]), qf[ks], s);
            s2 = E::mfma(lds_read_b128(kbase + k_rd[ks + 1]), qf[ks + 1], s2);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] += s2[r];
    };
    auto mask_scores = [&](f32x16& s, int n0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = n0 + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (j < lo || j > hi) s[r] = -INFINITY;
        }
    };
    // row max of 16 scores + running-max update (deferred rescale); returns the max to subtract
    auto softmax_max = [&](const f32x16& s) -> float {
        float mx = fmaxf(s[0], s[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s[r], s[r + 1]));
        mx = xhalf_max(mx) * c;
        if (!__all(mx - m_run <= P3_RESCALE_THR)) {       // rare after the first tiles
            const float m_new = fmaxf(m_run, mx);
            const float m_use0 = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - m_use0);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DBLKS; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }
        return (m_run == -INFINITY) ? 0.f : m_run;
    };
    // exp2 + pack of 8 scores (one PV k-step)
    auto softmax_exp8 = [&](const f32x16& s, int base, float m_use, u32x4& pf) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { e[j] = fast_exp2(fmaf(s[base + j], c, -m_use)); l_run += e[j]; }
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) pf[w2] = E::pack2(e[2 * w2], e[2 * w2 + 1]);
    };
    auto read_v = [&](const char* vbase, int ks2, u32x4 (&vf)[DBLKS]) {
#pragma unroll
        for (int d = 0; d < DBLKS; ++d) {
            const u32x2 v0 = lds_read_tr16(vbase + v_rd[0][d] + ks2 * 16 * D * 2);   // swizzle is invariant under +16 rows
            const u32x2 v1 = lds_read_tr16(vbase + v_rd[1][d] + ks2 * 16 * D * 2);
            vf[d] = u32x4{v0[0], v0[1], v1[0], v1[1]};
        }
    };

    // One sub-step.  cur: scores of sub-tile j (keys n0 .. n0+31, V rows at vbase); nxt: receives the
    // scores of sub-tile j+1 (K rows at kbase_next) when do_x.  The X MFMAs are fenced between the
    // VALU slices of Y so that the softmax of tile j executes under them.
    auto substep = [&](f32x16& cur, f32x16& nxt, const char* kbase_next, const char* vbase, int n0,
                       bool do_x, bool do_yz) {
        if (do_x && do_yz) {
            const bool need_mask = (n0 + 31 > w_hi_min) || (n0 < w_lo_max);
            if (need_mask) mask_scores(cur, n0);
            // ---- first half of X  ||  row max ----
            // X accumulates into TWO chains (even / odd k-steps): a dependent MFMA issued right behind
            // its predecessor stalls, two alternating chains do not; they are summed into `nxt` below.
            u32x4 kfr[KSTEPS / 2];
            f32x16 nx2;
#pragma unroll
            for (int ks = 0; ks < KSTEPS / 2; ++ks) kfr[ks] = lds_read_b128(kbase_next + k_rd[ks]);
#pragma unroll
            for (int r = 0; r < 16; ++r) { nxt[r] = 0.f; nx2[r] = 0.f; }
            FA_FENCE();
#pragma unroll
            for (int ks = 0; ks < KSTEPS / 2; ++ks) {
                if (ks & 1) nx2 = E::mfma(kfr[ks], qf[ks], nx2);
                else        nxt = E::mfma(kfr[ks], qf[ks], nxt);
            }
            const float m_use = softmax_max(cur);
            FA_FENCE();
            // ---- second half of X  ||  exp2 / pack of the first PV k-step, V fragments of k-step 0 ----
            u32x4 pf0, pf1;
            u32x4 vf[DBLKS];
#pragma unroll
            for (int ks = 0; ks < KSTEPS / 2; ++ks) kfr[ks] = lds_read_b128(kbase_next + k_rd[KSTEPS / 2 + ks]);
            read_v(vbase, 0, vf);
            FA_FENCE();
#pragma unroll
            for (int ks = 0; ks < KSTEPS / 2; ++ks) {
                if (ks & 1) nx2 = E::mfma(kfr[ks], qf[KSTEPS / 2 + ks], nx2);
                else        nxt = E::mfma(kfr[ks], qf[KSTEPS / 2 + ks], nxt);
                if (ks == 0) softmax_exp8(cur, 0, m_use, pf0);
            }
            FA_FENCE();
            // ---- Z k-step 0  ||  exp2 / pack of the second k-step ----
#pragma unroll
            for (int d = 0; d < DBLKS; ++d) {
                oacc[d] = E::mfma(vf[d], pf0, oacc[d]);
                if (d == 0) softmax_exp8(cur, 8, m_use, pf1);
            }
            FA_FENCE();
            read_v(vbase, 1, vf);
#pragma unroll
            for (int d = 0; d < DBLKS; ++d) oacc[d] = E::mfma(vf[d], pf1, oacc[d]);
#pragma unroll
            for (int r = 0; r < 16; ++r) nxt[r] += nx2[r];
        } else {
            if (do_x) stage_x(kbase_next, nxt);
            if (do_yz) {
                const bool need_mask = (n0 + 31 > w_hi_min) || (n0 < w_lo_max);
                if (need_mask) mask_scores(cur, n0);
                const float m_use = softmax_max(cur);
                u32x4 pf0, pf1;
                u32x4 vf[DBLKS];
                softmax_exp8(cur, 0, m_use, pf0);
                softmax_exp8(cur, 8, m_use, pf1);
                read_v(vbase, 0, vf);
#pragma unroll
                for (int d = 0; d < DBLKS; ++d) oacc[d] = E::mfma(vf[d], pf0, oacc[d]);
                read_v(vbase, 1, vf);
#pragma unroll
                for (int d = 0; d < DBLKS; ++d) oacc[d] = E::mfma(vf[d], pf1, oacc[d]);
            }
        }
    };

    // ------------------------------------------------------------------------------------------
    // pipeline over tiles
    // ------------------------------------------------------------------------------------------
    const int n_tiles = n_max - n_min;
    f32x16 sA, sB;                                       // ping-pong score tiles
    if (n_tiles > 0) {
        dma_k(n_min, 0);
        dma_v(n_min, 0);
        if (n_tiles > 1) dma_k(n_min + 1, 1);
    }
    __syncthreads();                                     // (hipcc drains the DMAs in front of the barrier)
    {
        const int j0 = 2 * n_min;
        if (n_tiles > 0 && j0 >= j_first && j0 <= j_last) stage_x(smem, sA);
    }
    int kr = 0, vr = 0;                                  // ring slots of K(t), V(t)
    for (int t = n_min; t < n_max; ++t) {
        const int kr1 = kr == 2 ? 0 : kr + 1;            // K(t+1)
        const int kr2 = kr1 == 2 ? 0 : kr1 + 1;          // receives K(t+2)
        if (t + 2 < n_max) dma_k(t + 2, kr2);
        if (t + 1 < n_max) dma_v(t + 1, vr ^ 1);
        const char* kt = smem + kr * TILE;
        const char* kt1 = smem + kr1 * TILE;
        const char* vt = smem + V_OFF + vr * TILE;
        const int j = 2 * t;
        const bool in0 = j >= j_first && j <= j_last;
        const bool in1 = j + 1 >= j_first && j + 1 <= j_last;
        const bool in2 = (t + 1 < n_max) && j + 2 >= j_first && j + 2 <= j_last;
        // sub-step 0: X(t,1) || Y,Z(t,0)
        substep(sA, sB, kt + 32 * D * 2, vt, t * P3_BN, in1, in0);
        // sub-step 1: X(t+1,0) || Y,Z(t,1)
        substep(sB, sA, kt1, vt + 32 * D * 2, t * P3_BN + 32, in2, in1);
        __syncthreads();
        kr = kr1; vr ^= 1;
    }

    // ---- epilogue ----
    const float l_tot = xhalf_sum(l_run);
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
    }   // pass
}

template <typename T, int D>
static int launch_p3_td(const KArgs& a_in, hipStream_t stream) {
    KArgs a = a_in;
    a.n_qblocks_total = (a.p.seqlen_q + P3_BM - 1) / P3_BM;
    a.pair_qblocks = (a_in.pair_qblocks && a.n_qblocks_total >= 2) ? 1 : 0;
    a.n_qblocks = a.pair_qblocks ? (a.n_qblocks_total + 1) / 2 : a.n_qblocks_total;
    const int grid = work_grid(a.p.batch, a.p.nheads_q, a.p.nheads_k, a.n_qblocks);
    const size_t smem = P3Smem<D>::TOTAL;
    if (grid == 0) return 0;
    auto kern = fa_fwd_p3_kernel<T, D>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(P3_THREADS), smem, stream, a);
    return 0;
}

// returns 1 when this configuration is not served by the pipelined kernel
int launch_fwd_pipe(const KArgs& a, hipStream_t stream) {
    const fa_params& p = a.p;
    if (a.has_bias || p.block_table || p.p_dropout > 0.f || a.kv_batch_idx || a.leftpad_k || a.seqlen_k_add) return 1;
    if (a.seqlens_k && !p.cu_seqlens_k) return 1;                    // kv-cache lengths: general kernel
    if (p.seqlen_q < 512) return 1;                                   // small problems: 128-row blocks fill the chip better
    const bool bf = p.dtype == FA_BF16;
    switch (p.head_dim) {
        case 64:  return bf ? launch_p3_td<bf16_tag, 64>(a, stream) : launch_p3_td<fp16_tag, 64>(a, stream);
        case 128: return bf ? launch_p3_td<bf16_tag, 128>(a, stream) : launch_p3_td<fp16_tag, 128>(a, stream);
        default:  return 1;
    }
}

}  // namespace fa
