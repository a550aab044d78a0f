// fa_fwd_pp.hip - forward for the plain case (no bias, no dropout, contiguous K/V): 8 waves, two per SIMD, in
// ENFORCED alternation ("ping-pong").
//
// Same algorithm, tile formats and results as fa_fwd_kernel (fa_fwd.hip; reference kernel/fused_mha_forward.cu:25-224,
// fused_mha_forward_varlen.cu:25-275).  In fa_fwd_kernel the two waves that share a SIMD belong to two independent
// workgroups: whether one wave's softmax (VALU) runs beside the other's MFMAs is left to chance, and the measured
// time per tile is close to the SUM of the two.  Here a workgroup is 512 threads = two groups of four waves, each
// group owning 128 of the workgroup's 256 query rows; both groups walk the same K/V tiles (one LDS copy, half the
// DMA traffic per row) and every wave alternates between
//     SM(j):  softmax of S(j) -> P(j)                      VALU only
//     MM(j):  O += V(j)^T P(j) ;  S(j+1) = K(j+1) Q^T      MFMA + LDS reads only
// with one s_barrier after each.  Group 1 starts one barrier later than group 0, so on every SIMD one wave is in
// SM while the other is in MM, for the whole loop.
//
// LDS: K ring of 2 tiles + V ring of 2 tiles (64 KiB).  K(j+2) / V(j+1) are fetched by LDS-DMA during the interval
// in which group 0 runs MM(j) and group 1 runs SM(j) (their slots were last read one interval earlier) and are first
// read two barriers later; the barriers are bare s_barrier (a __syncthreads would drain the DMA queue at every one),
// each wave waits for its own DMA share (vmcnt(0)) just before the last barrier in front of the first reader.
#include <cstdlib>
#include <type_traits>
#include "fa_common.h"

namespace fa {

constexpr int PP_BM = 256;
constexpr int PP_BN = 64;
constexpr int PP_NKB = PP_BN / 32;
constexpr int PP_THREADS = 512;
constexpr float PP_RESCALE_THR = 8.0f;                 // log2 units (same as fa_fwd_kernel)
#ifndef FA_PP_PF
#define FA_PP_PF 3                                     // operand fragments in flight ahead of their MFMA
#endif

template <int D> struct FwdPpSmem {
    static constexpr int TILE = PP_BN * D * 2;
    static constexpr int TOTAL = 4 * TILE;             // K0 K1 V0 V1
};

template <typename T, int D>
__global__ void __launch_bounds__(PP_THREADS, 2) fa_fwd_pp_kernel(const KArgs a) {
    using E = Elem<T>;
    constexpr int KSTEPS = D / 16;
    constexpr int DBLKS = D / 32;
    constexpr int CPR = D / 8;
    constexpr int CHUNKS = PP_BN * CPR / PP_THREADS;    // 1-KiB DMA instructions per wave, tensor and tile
    constexpr int TILE = FwdPpSmem<D>::TILE;
    constexpr int NQK = KSTEPS * PP_NKB;
    constexpr int NT = 2 * PP_NKB;
    constexpr int NPV = NT * DBLKS;
    static_assert(CHUNKS >= 1, "tile too small for 512 threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const fa_params& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const WorkItem w = a.flat_blocks
        ? decode_work_flat(blockIdx.x, a.flat_blocks, PP_BM, p.batch, p.nheads_q, p.nheads_k, p.cu_seqlens_q, lane)
        : decode_work(blockIdx.x, p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
    if (!w.valid) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gp = wave >> 2;                           // group 0 leads, group 1 runs one barrier behind
    const int l31 = lane & 31;
    const int g = lane >> 5;

    // ---- per-sequence geometry (as fa_fwd_kernel) ----
    int seqlen_q = p.seqlen_q, seqlen_k = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;
    int kv_b = w.b;
    if (p.cu_seqlens_q) {
        q_row0 = p.cu_seqlens_q[w.b];
        seqlen_q = p.cu_seqlens_q[w.b + 1] - (int)q_row0;
    }
    if (p.cu_seqlens_k) {
        const int k0 = p.cu_seqlens_k[w.b];
        seqlen_k = p.cu_seqlens_k[w.b + 1] - k0;
        k_row0 = k0;
    }
    if (a.seqlens_k) {
        const int su = a.seqlens_k[w.b] + a.seqlen_k_add;
        if (p.cu_seqlens_k) seqlen_k = su > 0 ? (su < seqlen_k ? su : seqlen_k) : 0;
        else seqlen_k = su;
    }
    if (a.kv_batch_idx) kv_b = a.kv_batch_idx[w.b];
    if (a.leftpad_k) k_row0 += a.leftpad_k[w.b];

    const int off = seqlen_k - seqlen_q;
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;
    const float c = a.scale_log2e;
    const int dv = valid_cols(p);

    // lane-constant LDS read addresses, pinned (slot, key block, k-step are immediate offsets)
    const lds_char* k_ptr[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) k_ptr[ks] = lds_pin(smem + swz_row_off<D>(l31, 32 * ks + 16 * g));
    const int v_rr = (lane & 15) >> 2;
    const int v_cb = (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);
    const lds_char* v_ptr[2][DBLKS];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int d = 0; d < DBLKS; ++d) v_ptr[h][d] = lds_pin(smem + 2 * TILE + swzt_row_off<D>(4 * g + v_rr + 8 * h, d * 64 + v_cb));

    const int n_pass = (a.pair_qblocks && (a.n_qblocks_total - 1 - w.qb) != w.qb) ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
    const int qb_cur = pass == 0 ? w.qb : a.n_qblocks_total - 1 - w.qb;
    const int m_block = qb_cur * PP_BM;
    if (m_block >= seqlen_q) continue;                  // (workgroup-uniform)
    int n_min = 0, n_max = (seqlen_k + PP_BN - 1) / PP_BN;
    {
        const int m_last = (m_block + PP_BM < seqlen_q ? m_block + PP_BM : seqlen_q) - 1;
        if (wr >= 0) {
            const int kmax = m_last + off + wr;
            const int t = kmax < 0 ? 0 : kmax / PP_BN + 1;
            n_max = t < n_max ? t : n_max;
        }
        if (wl >= 0) {
            const int kmin = m_block + off - wl;
            if (kmin > 0) n_min = kmin / PP_BN;
        }
    }
    const int J = n_max - n_min;                        // tiles of this block (workgroup-uniform)

    const int wave_row0 = m_block + wave * 32;
    const int my_row = wave_row0 + l31;
    int lo = 0, hi = seqlen_k - 1;
    if (wr >= 0) { const int h2 = my_row + off + wr; hi = h2 < hi ? h2 : hi; }
    if (wl >= 0) { const int l2 = my_row + off - wl; lo = l2 > lo ? l2 : lo; }
    const int wrow_last = wave_row0 + 31;
    int w_hi_min = seqlen_k - 1, w_hi_max = seqlen_k - 1, w_lo_max = 0;
    if (wr >= 0) {
        const int h0 = wave_row0 + off + wr, h1 = wrow_last + off + wr;
        w_hi_min = h0 < w_hi_min ? h0 : w_hi_min;
        w_hi_max = h1 < w_hi_max ? h1 : w_hi_max;
    }
    if (wl >= 0) { const int l1 = wrow_last + off - wl; w_lo_max = l1 > 0 ? l1 : 0; }
    const int w_lo_min = (wl >= 0 && wave_row0 + off - wl > 0) ? wave_row0 + off - wl : 0;

    const uint16_t* qp = reinterpret_cast<const uint16_t*>(p.q) + (p.cu_seqlens_q ? 0 : (int64_t)w.b * p.q_batch_stride)
                         + q_row0 * p.q_row_stride + (int64_t)w.h * p.q_head_stride;
    const uint16_t* kp = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)w.hk * p.k_head_stride
                         + (p.cu_seqlens_k ? 0 : (int64_t)kv_b * p.k_batch_stride) + k_row0 * p.k_row_stride;
    const uint16_t* vp = reinterpret_cast<const uint16_t*>(p.v) + (int64_t)w.hk * p.v_head_stride
                         + (p.cu_seqlens_k ? 0 : (int64_t)kv_b * p.v_batch_stride) + k_row0 * p.v_row_stride;

    // ---- Q fragments: B operand of S^T = K Q^T, lane holds Q[my_row][16ks + 8g .. +7] ----
    u32x4 qf[KSTEPS];
    {
        const bool ok = my_row < seqlen_q;
        const uint16_t* qrow = qp + (int64_t)my_row * p.q_row_stride + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            u32x4 z = {0, 0, 0, 0};
            qf[ks] = (ok && 16 * ks + 8 * g < dv) ? *reinterpret_cast<const u32x4*>(qrow + 16 * ks) : z;
        }
    }

    // ---- K / V staging by LDS-DMA (formats of fa_fwd_kernel: K swz, V swzt) ----
    constexpr int ROWS_PI = 64 / CPR;
    uint32_t k_voff[CHUNKS], v_voff[CHUNKS];
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int inst = wave * CHUNKS + i;
        const int row = inst * ROWS_PI + lane / CPR;
        const int slot = lane % CPR;
        const int k_cb = swz_row_off<D>(row, slot * 16) - row * D * 2;
        const int v_cb2 = swzt_row_off<D>(row, slot * 16) - row * D * 2;
        k_voff[i] = k_cb < dv * 2 ? (uint32_t)(row * p.k_row_stride * 2 + k_cb) : kOobVoff;
        v_voff[i] = v_cb2 < dv * 2 ? (uint32_t)(row * p.v_row_stride * 2 + v_cb2) : kOobVoff;
    }
    const __amdgpu_buffer_rsrc_t k_rsrc = make_rsrc(kp, p.k_row_stride, seqlen_k, dv);
    const __amdgpu_buffer_rsrc_t v_rsrc = make_rsrc(vp, p.v_row_stride, seqlen_k, dv);
    const uint32_t k_tile_bytes = (uint32_t)(PP_BN * p.k_row_stride * 2);
    const uint32_t v_tile_bytes = (uint32_t)(PP_BN * p.v_row_stride * 2);
    // tile j (relative to n_min) lives in slot j & 1 of its ring
    auto load_k = [&](int j, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
#ifdef FA_PP_KO_DMA
        return;
#endif
        const uint32_t so = (uint32_t)(n_min + j) * k_tile_bytes;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(k_rsrc, smem + slot * TILE + (wave * CHUNKS + i) * 1024, k_voff[i], so);
    };
    auto load_v = [&](int j, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
#ifdef FA_PP_KO_DMA
        return;
#endif
        const uint32_t so = (uint32_t)(n_min + j) * v_tile_bytes;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(v_rsrc, smem + (2 + slot) * TILE + (wave * CHUNKS + i) * 1024, v_voff[i], so);
    };
    auto bar = [&]() { __builtin_amdgcn_s_barrier(); };
    auto dma_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    f32x16 oacc[DBLKS];
#pragma unroll
    for (int d = 0; d < DBLKS; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 sacc[PP_NKB];
    u32x4 pf[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) pf[t] = u32x4{0, 0, 0, 0};

    auto active = [&](int j) { const int n0 = (n_min + j) * PP_BN; return (n0 <= w_hi_max) && (n0 + PP_BN - 1 >= w_lo_min); };

    // S(j) = K(j) Q^T from K slot `slot`
    auto qk = [&](int j, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        if (!active(j)) return;
#ifdef FA_PP_KO_MM
#pragma unroll
        for (int kb = 0; kb < PP_NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "=v"(sacc[kb][r]));       // opaque garbage scores
        return;
#endif
#pragma unroll
        for (int kb = 0; kb < PP_NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#ifdef FA_PP_KO_LDS
        auto kread = [&](int i) { u32x4 x; asm volatile("" : "=v"(x)); return x; };     // measurement build: no LDS reads
#else
        auto kread = [&](int i) { return lds_read_b128(k_ptr[i / PP_NKB] + (slot * TILE + (i % PP_NKB) * 32 * D * 2)); };
#endif
        u32x4 kk[NQK];
#pragma unroll
        for (int i = 0; i < FA_PP_PF; ++i) kk[i] = kread(i);
#pragma unroll
        for (int i = 0; i < NQK; ++i) {
            if (i + FA_PP_PF < NQK) kk[i + FA_PP_PF] = kread(i + FA_PP_PF);
            __builtin_amdgcn_sched_barrier(0);
            sacc[i % PP_NKB] = E::mfma(kk[i], qf[i / PP_NKB], sacc[i % PP_NKB]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // O += V(j)^T P(j) from V slot `slot`
    auto pv = [&](int j, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        if (!active(j)) return;
#ifdef FA_PP_KO_MM
        // measurement build: no MFMA / LDS reads, the softmax stays alive (P consumed by an empty asm)
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("" :: "v"(pf[t]));
        return;
#endif
        auto vread = [&](int i) {
#ifdef FA_PP_KO_LDS
            { u32x4 x; asm volatile("" : "=v"(x)); return x; }
#endif
            const int t = i / DBLKS, d = i % DBLKS;
            const u32x2 v0 = lds_read_tr16(v_ptr[0][d] + (slot * TILE + 16 * t * D * 2));
            const u32x2 v1 = lds_read_tr16(v_ptr[1][d] + (slot * TILE + 16 * t * D * 2));
            return u32x4{v0[0], v0[1], v1[0], v1[1]};
        };
        u32x4 vf[NPV];
#pragma unroll
        for (int i = 0; i < FA_PP_PF; ++i) vf[i] = vread(i);
#pragma unroll
        for (int i = 0; i < NPV; ++i) {
            if (i + FA_PP_PF < NPV) vf[i + FA_PP_PF] = vread(i + FA_PP_PF);
            __builtin_amdgcn_sched_barrier(0);
            oacc[i % DBLKS] = E::mfma(vf[i], pf[i / DBLKS], oacc[i % DBLKS]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // softmax of S(j): mask on edge tiles, running max with deferred rescale, P = exp2(S c - m) packed to 16 bit
    auto sm = [&](int j) {
        if (!active(j)) return;
#ifdef FA_PP_KO_SM
        // measurement build: no softmax arithmetic, the MFMAs stay alive (P = packed raw S)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) pf[t][e] = E::pack2(sacc[t >> 1][8 * (t & 1) + 2 * e], sacc[t >> 1][8 * (t & 1) + 2 * e + 1]);
        return;
#endif
        const int n0 = (n_min + j) * PP_BN;
        const bool need_mask = (n0 + PP_BN - 1 > w_hi_min) || (n0 < w_lo_max);
        if (need_mask) {
            const int lo_t = lo - n0 - 4 * g;
            const uint32_t width = (uint32_t)(hi - lo);
            const bool empty = hi < lo;
#pragma unroll
            for (int kb = 0; kb < PP_NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cpos = kb * 32 + (r & 3) + 8 * (r >> 2);
                    if (empty || (uint32_t)(cpos - lo_t) > width) sacc[kb][r] = -INFINITY;
                }
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
#pragma unroll
        for (int kb = 1; kb < PP_NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
        mx = xhalf_max(mx) * c;
        if (!__all(mx - m_run <= PP_RESCALE_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - m_use);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DBLKS; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }
        const float ms = (m_run == -INFINITY) ? 0.f : m_run;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < PP_NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = fast_exp2(fmaf(sacc[kb][r], c, -ms));
                sacc[kb][r] = e;
                psum += e;
            }
        l_run += psum;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) pf[t][e] = E::pack2(sacc[t >> 1][8 * (t & 1) + 2 * e], sacc[t >> 1][8 * (t & 1) + 2 * e + 1]);
    };

    // ---- prologue: K(0), V(0), K(1) ----
    bar();                                              // (second pass: everyone is done with the previous block's tiles)
    if (J > 0) {
        load_k(0, std::integral_constant<int, 0>{});
        load_v(0, std::integral_constant<int, 0>{});
        if (J > 1) load_k(1, std::integral_constant<int, 1>{});
    }
    dma_landed();
    bar();
    if (J > 0) {
        if (gp == 1) bar();                             // group 1 runs one interval behind
        qk(0, std::integral_constant<int, 0>{});
        bar();
        // steady state, two tiles per trip so that ring slots are compile-time constants
        auto step = [&](int j, auto par_c) {
            constexpr int par = decltype(par_c)::value;         // = j & 1
            // interval A (for this wave): SM(j).  Group 1 fetches its share of K(j+2), V(j+1) first.
            if (gp == 1) {
                if (j + 2 < J) load_k(j + 2, std::integral_constant<int, par>{});
                if (j + 1 < J) load_v(j + 1, std::integral_constant<int, par ^ 1>{});
            }
            sm(j);
            if (gp == 0) dma_landed();                  // group 0's share, issued one interval ago
            bar();
            // interval B: MM(j) = PV(j), then QK(j+1).  Group 0 fetches its share between the two
            // (after the transposing reads: hipcc drains vmcnt in front of those).
            pv(j, std::integral_constant<int, par>{});
            if (gp == 0) {
                if (j + 2 < J) load_k(j + 2, std::integral_constant<int, par>{});
                if (j + 1 < J) load_v(j + 1, std::integral_constant<int, par ^ 1>{});
            }
            if (j + 1 < J) qk(j + 1, std::integral_constant<int, par ^ 1>{});
            if (gp == 1) dma_landed();                  // group 1's share, issued one interval ago
            bar();
        };
        for (int j = 0; j < J; j += 2) {
            step(j, std::integral_constant<int, 0>{});
            if (j + 1 < J) step(j + 1, std::integral_constant<int, 1>{});
        }
        if (gp == 0) bar();                             // match group 1's extra leading barrier
    }

    // ---- epilogue: O / l, LSE ----
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
                if (d * 32 + 8 * rq + 4 * g < dv) *reinterpret_cast<u32x2*>(op + d * 32 + 8 * rq + 4 * g) = o2;
            }
        if (g == 0) {
            const float lse = l_tot > 0.f ? (m_run + fast_log2(l_tot)) * kLn2 : -INFINITY;
            p.lse[(int64_t)w.b * p.lse_batch_stride + (int64_t)w.h * p.lse_head_stride + q_row0 + my_row] = lse;
        }
    }
    }   // pass
}

// Returns 1 when it launched, 0 when the caller should use fa_fwd_kernel.
int launch_fwd_pp(const KArgs& a0, hipStream_t stream) {
    static const bool enabled = [] { const char* e = getenv("FA_FWD_PP"); return !(e && e[0] == '0'); }();
    const fa_params& p = a0.p;
    if (!enabled || p.block_table || p.p_dropout > 0.f || a0.has_bias || p.head_dim != 128) return 0;
    if (p.seqlen_q < 2 * PP_BM) return 0;              // short queries: 128-row workgroups fill the chip better
    KArgs a = a0;
    a.n_qblocks_total = (p.seqlen_q + PP_BM - 1) / PP_BM;
    int grid;
    if (a0.flat_blocks) {
        a.flat_blocks = p.total_q / PP_BM + p.batch;
        a.pair_qblocks = 0;
        a.n_qblocks = a.n_qblocks_total;
        grid = a.flat_blocks * p.nheads_q;
    } else {
        a.pair_qblocks = ((p.is_causal || p.window_right >= 0) && p.window_left < 0 && a.n_qblocks_total >= 2) ? 1 : 0;
        a.n_qblocks = a.pair_qblocks ? (a.n_qblocks_total + 1) / 2 : a.n_qblocks_total;
        grid = work_grid(p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
    }
    if (grid == 0) return 1;
    const size_t smem = FwdPpSmem<128>::TOTAL;
    if (p.dtype == FA_BF16) {
        auto kern = fa_fwd_pp_kernel<bf16_tag, 128>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PP_THREADS), smem, stream, a);
    } else {
        auto kern = fa_fwd_pp_kernel<fp16_tag, 128>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PP_THREADS), smem, stream, a);
    }
    return 1;
}

}  // namespace fa
