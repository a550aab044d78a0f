// fa_fwd64.hip - forward for the plain case (no bias, no dropout, contiguous K/V) with 64 query rows per wave.
//
// Same algorithm, tile formats and results as fa_fwd_kernel (fa_fwd.hip; reference kernel/fused_mha_forward.cu:25-224
// and fused_mha_forward_varlen.cu:25-275), different shape: a workgroup is 4 waves x 64 rows = 256 query rows and
// every wave owns the whole 512-register file of its SIMD (one wave per SIMD).  A wave carries TWO 32-row blocks,
// so each K / V fragment read from LDS feeds two MFMAs, and the two blocks run half a tile out of phase:
//
//     A  S0 = K Q0^T
//     B  S1 = K Q1^T   beside   P0 = exp2(S0 - m0), row sums, 16-bit packing
//     C  O0 += V^T P0  beside   P1 = exp2(S1 - m1), ...
//     D  O1 += V^T P1
//
// In B and C the matrix pipe and the VALU work of the OTHER row block sit in one basic block; scheduling
// fences (sched_group_barrier) place ~7 VALU between consecutive MFMAs, which is what one MFMA covers.  The row
// maximum and the (rare) deferred rescale are decided between the phases, so B and C are branch free.  Tiles that
// need masking add the mask between the phases (wave-uniform branches).
#include <cstdlib>
#include <type_traits>
#include "fa_common.h"

namespace fa {

constexpr int F64_BM = 256;
constexpr int F64_BN = 64;
constexpr int F64_NKB = F64_BN / 32;
constexpr int F64_RB = 2;                              // 32-row blocks per wave
constexpr int F64_THREADS = 256;
constexpr float F64_RESCALE_THR = 8.0f;                // log2 units (same as fa_fwd_kernel)
#ifndef FA_F64_PF
#define FA_F64_PF 3                                    // operand fragments in flight ahead of their MFMA
#endif
#ifndef FA_F64_VALU_PER_MFMA
#define FA_F64_VALU_PER_MFMA 7
#endif

template <int D> struct Fwd64Smem {
    static constexpr int TILE = F64_BN * D * 2;
    static constexpr int STAGE = 2 * TILE;
    static constexpr int TOTAL = 2 * STAGE;
};

// acc *= alpha for one 32x32 accumulator tile that lives in the accumulator half of the register file (VALU cannot
// touch it: read, multiply, write back).  The tile is pinned to a fixed accumulator range so that the instruction text
// can name its registers; slot = 0..7 picks a[16 slot : 16 slot + 15].
#define FA_RS1(n) "v_accvgpr_read_b32 %1, a" #n "\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a" #n ", %1\n\t"
#define FA_RS16(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15)                                   \
    FA_RS1(a0) FA_RS1(a1) FA_RS1(a2) FA_RS1(a3) FA_RS1(a4) FA_RS1(a5) FA_RS1(a6) FA_RS1(a7)                             \
    FA_RS1(a8) FA_RS1(a9) FA_RS1(a10) FA_RS1(a11) FA_RS1(a12) FA_RS1(a13) FA_RS1(a14) FA_RS1(a15)
#define FA_RS_CASE(slot, lo, hi, ...)                                                                                   \
    case slot: asm volatile(FA_RS16(__VA_ARGS__) : "+{a[" #lo ":" #hi "]}"(acc), "=&v"(tmp) : "v"(alpha)); break;
__device__ __forceinline__ void scale_acc(int slot, f32x16& acc, float alpha) {
    float tmp;
    switch (slot) {
        FA_RS_CASE(0, 0, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
        FA_RS_CASE(1, 16, 31, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31)
        FA_RS_CASE(2, 32, 47, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47)
        FA_RS_CASE(3, 48, 63, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63)
        FA_RS_CASE(4, 64, 79, 64, 65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79)
        FA_RS_CASE(5, 80, 95, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95)
        FA_RS_CASE(6, 96, 111, 96, 97, 98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111)
        FA_RS_CASE(7, 112, 127, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 126, 127)
        default: break;
    }
}

template <typename T, int D>
__global__ void __launch_bounds__(F64_THREADS, 1) fa_fwd64_kernel(const KArgs a) {
    using E = Elem<T>;
    constexpr int KSTEPS = D / 16;
    constexpr int DBLKS = D / 32;
    constexpr int CPR = D / 8;
    constexpr int CHUNKS = F64_BN * CPR / F64_THREADS;
    constexpr int TILE = Fwd64Smem<D>::TILE;
    constexpr int STAGE = Fwd64Smem<D>::STAGE;
    constexpr int NQK = KSTEPS * F64_NKB;              // K fragments per tile
    constexpr int NT = 2 * F64_NKB;                    // 16-key contraction steps of the PV product
    constexpr int NPV = NT * DBLKS;                    // V fragments per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const fa_params& p = a.p;
    const WorkItem w = decode_work(blockIdx.x, p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
    if (!w.valid) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    const int n_pass = (a.pair_qblocks && (a.n_qblocks_total - 1 - w.qb) != w.qb) ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
    const int qb_cur = pass == 0 ? w.qb : a.n_qblocks_total - 1 - w.qb;
    const int m_block = qb_cur * F64_BM;
    if (m_block >= seqlen_q) continue;
    int n_min = 0, n_max = (seqlen_k + F64_BN - 1) / F64_BN;
    {
        const int m_last = (m_block + F64_BM < seqlen_q ? m_block + F64_BM : seqlen_q) - 1;
        if (wr >= 0) {
            const int kmax = m_last + off + wr;
            const int t = kmax < 0 ? 0 : kmax / F64_BN + 1;
            n_max = t < n_max ? t : n_max;
        }
        if (wl >= 0) {
            const int kmin = m_block + off - wl;
            if (kmin > 0) n_min = kmin / F64_BN;
        }
    }

    const int wave_row0 = m_block + wave * 32 * F64_RB;
    int my_row[F64_RB], lo[F64_RB], hi[F64_RB];
    int b_hi_min[F64_RB], b_hi_max[F64_RB], b_lo_max[F64_RB], b_lo_min[F64_RB];   // wave-uniform, per row block
#pragma unroll
    for (int rb = 0; rb < F64_RB; ++rb) {
        const int r0 = wave_row0 + 32 * rb, r1 = r0 + 31;
        my_row[rb] = r0 + l31;
        lo[rb] = 0; hi[rb] = seqlen_k - 1;
        if (wr >= 0) { const int h2 = my_row[rb] + off + wr; hi[rb] = h2 < hi[rb] ? h2 : hi[rb]; }
        if (wl >= 0) { const int l2 = my_row[rb] + off - wl; lo[rb] = l2 > lo[rb] ? l2 : lo[rb]; }
        b_hi_min[rb] = seqlen_k - 1; b_hi_max[rb] = seqlen_k - 1; b_lo_max[rb] = 0;
        if (wr >= 0) {
            const int h0 = r0 + off + wr, h1 = r1 + off + wr;
            b_hi_min[rb] = h0 < b_hi_min[rb] ? h0 : b_hi_min[rb];
            b_hi_max[rb] = h1 < b_hi_max[rb] ? h1 : b_hi_max[rb];
        }
        if (wl >= 0) { const int l1 = r1 + off - wl; b_lo_max[rb] = l1 > 0 ? l1 : 0; }
        b_lo_min[rb] = (wl >= 0 && r0 + off - wl > 0) ? r0 + off - wl : 0;
    }

    const uint16_t* qp = reinterpret_cast<const uint16_t*>(p.q) + (p.cu_seqlens_q ? 0 : (int64_t)w.b * p.q_batch_stride)
                         + q_row0 * p.q_row_stride + (int64_t)w.h * p.q_head_stride;
    const uint16_t* kp = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)w.hk * p.k_head_stride
                         + (p.cu_seqlens_k ? 0 : (int64_t)kv_b * p.k_batch_stride) + k_row0 * p.k_row_stride;
    const uint16_t* vp = reinterpret_cast<const uint16_t*>(p.v) + (int64_t)w.hk * p.v_head_stride
                         + (p.cu_seqlens_k ? 0 : (int64_t)kv_b * p.v_batch_stride) + k_row0 * p.v_row_stride;

    // ---- Q fragments: B operand of S^T = K Q^T, lane holds Q[my_row][16ks + 8g .. +7] ----
    u32x4 qf[F64_RB][KSTEPS];
#pragma unroll
    for (int rb = 0; rb < F64_RB; ++rb) {
        const bool ok = my_row[rb] < seqlen_q;
        const uint16_t* qrow = qp + (int64_t)my_row[rb] * p.q_row_stride + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            u32x4 z = {0, 0, 0, 0};
            qf[rb][ks] = (ok && 16 * ks + 8 * g < dv) ? *reinterpret_cast<const u32x4*>(qrow + 16 * ks) : z;
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
        const int v_cb = swzt_row_off<D>(row, slot * 16) - row * D * 2;
        k_voff[i] = k_cb < dv * 2 ? (uint32_t)(row * p.k_row_stride * 2 + k_cb) : kOobVoff;
        v_voff[i] = v_cb < dv * 2 ? (uint32_t)(row * p.v_row_stride * 2 + v_cb) : kOobVoff;
    }
    const __amdgpu_buffer_rsrc_t k_rsrc = make_rsrc(kp, p.k_row_stride, seqlen_k, dv);
    const __amdgpu_buffer_rsrc_t v_rsrc = make_rsrc(vp, p.v_row_stride, seqlen_k, dv);
    const uint32_t k_tile_bytes = (uint32_t)(F64_BN * p.k_row_stride * 2);
    const uint32_t v_tile_bytes = (uint32_t)(F64_BN * p.v_row_stride * 2);
    auto load_tile = [&](int nb, auto stage_c) {
        constexpr int stage = decltype(stage_c)::value;
        char* base = smem + stage * STAGE;
        const uint32_t ks_off = (uint32_t)nb * k_tile_bytes, vs_off = (uint32_t)nb * v_tile_bytes;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(k_rsrc, base + (wave * CHUNKS + i) * 1024, k_voff[i], ks_off);
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(v_rsrc, base + TILE + (wave * CHUNKS + i) * 1024, v_voff[i], vs_off);
    };

    f32x16 oacc[F64_RB][DBLKS];
    float m_run[F64_RB], l_run[F64_RB];
#pragma unroll
    for (int rb = 0; rb < F64_RB; ++rb) {
        m_run[rb] = -INFINITY; l_run[rb] = 0.f;
#pragma unroll
        for (int d = 0; d < DBLKS; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[rb][d][r] = 0.f;
    }

    int k_rd[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) k_rd[ks] = swz_row_off<D>(l31, 32 * ks + 16 * g);
    const int v_rr = (lane & 15) >> 2;
    const int v_cb = (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);

    auto compute_tile = [&](auto stage_c, int nb) {
        constexpr int stage = decltype(stage_c)::value;
        const int n0 = nb * F64_BN;
        const char* sbase = smem + stage * STAGE;
        auto kread = [&](int i) { return lds_read_b128(sbase + k_rd[i / F64_NKB] + (i % F64_NKB) * 32 * D * 2); };
        auto vread = [&](int i) {      // fragment i = (t, d): keys 16 t .. +15 of the tile, 64-byte column block d
            const int t = i / DBLKS, d = i % DBLKS;
            const int row_a = 16 * t + 4 * g + v_rr;
            const u32x2 v0 = lds_read_tr16(sbase + TILE + swzt_row_off<D>(row_a, d * 64 + v_cb));
            const u32x2 v1 = lds_read_tr16(sbase + TILE + swzt_row_off<D>(row_a + 8, d * 64 + v_cb));
            return u32x4{v0[0], v0[1], v1[0], v1[1]};
        };
        f32x16 sacc[F64_RB][F64_NKB];
        u32x4 pf[F64_RB][NT];
        auto zero_s = [&](int rb) {
#pragma unroll
            for (int kb = 0; kb < F64_NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[rb][kb][r] = 0.f;
        };
        auto mask_s = [&](int rb) {
            const int lo_t = lo[rb] - n0 - 4 * g;
            const uint32_t width = (uint32_t)(hi[rb] - lo[rb]);
            const bool empty = hi[rb] < lo[rb];
#pragma unroll
            for (int kb = 0; kb < F64_NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cpos = kb * 32 + (r & 3) + 8 * (r >> 2);
                    if (empty || (uint32_t)(cpos - lo_t) > width) sacc[rb][kb][r] = -INFINITY;
                }
        };
        // row maximum of the tile and, rarely, the deferred rescale (a wave-uniform branch)
        auto max_rescale = [&](int rb) {
            float mx = sacc[rb][0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[rb][0][r]);
#pragma unroll
            for (int kb = 1; kb < F64_NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[rb][kb][r]);
            mx = xhalf_max(mx) * c;
            if (!__all(mx - m_run[rb] <= F64_RESCALE_THR)) {
                const float m_new = fmaxf(m_run[rb], mx);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = fast_exp2(m_run[rb] - m_use);
                m_run[rb] = m_new;
                l_run[rb] *= alpha;
                // The O accumulators live in the accumulator half of the register file.  Left to hipcc, this rare
                // multiply turns the loop-carried accumulators into VGPR values (128 copies per tile at the loop
                // head); done in place through "+a" operands they never leave the accumulator file.  The s_nops
                // cover the MFMA -> accvgpr_read and accvgpr_write -> MFMA wait states hipcc cannot see.
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
                for (int d = 0; d < DBLKS; ++d) scale_acc(rb * DBLKS + d, oacc[rb][d], alpha);
                asm volatile("s_nop 15" ::: "memory");
            }
        };
        // branch free: P = exp2(S c - m), row sum, 16-bit packing into the B operands of the PV product
        auto exp_pack = [&](int rb) {
            const float ms = (m_run[rb] == -INFINITY) ? 0.f : m_run[rb];
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < F64_NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = fast_exp2(fmaf(sacc[rb][kb][r], c, -ms));
                    sacc[rb][kb][r] = e;
                    psum += e;
                }
            l_run[rb] += psum;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    pf[rb][t][e] = E::pack2(sacc[rb][t >> 1][8 * (t & 1) + 2 * e], sacc[rb][t >> 1][8 * (t & 1) + 2 * e + 1]);
        };

        // P of row block rb must be finished HERE (the optimiser otherwise sinks exp_pack down to its first use,
        // out of the phase whose MFMAs are meant to cover it)
        auto pin_p = [&](int rb) {
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(pf[rb][t]));
            asm volatile("" : "+v"(l_run[rb]));
        };
        bool act[F64_RB], msk[F64_RB];
#pragma unroll
        for (int rb = 0; rb < F64_RB; ++rb) {
            act[rb] = (n0 <= b_hi_max[rb]) && (n0 + F64_BN - 1 >= b_lo_min[rb]);
            msk[rb] = (n0 + F64_BN - 1 > b_hi_min[rb]) || (n0 < b_lo_max[rb]);
        }
        if (!act[0] && !act[1]) return;                 // (wave-uniform) nothing visible for this wave in this tile
        // One code path for every tile: a row block without visible keys in this tile is masked to -inf like any
        // other (its P is 0); only the masking itself is conditional.
        zero_s(0); zero_s(1);
        // A
        {
            u32x4 kk[NQK];
#pragma unroll
            for (int i = 0; i < FA_F64_PF; ++i) kk[i] = kread(i);
#pragma unroll
            for (int i = 0; i < NQK; ++i) {
                if (i + FA_F64_PF < NQK) kk[i + FA_F64_PF] = kread(i + FA_F64_PF);
                __builtin_amdgcn_sched_barrier(0);
                sacc[0][i % F64_NKB] = E::mfma(kk[i], qf[0][i / F64_NKB], sacc[0][i % F64_NKB]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (msk[0]) mask_s(0);
        max_rescale(0);
        __builtin_amdgcn_sched_barrier(0);
        // B
        {
            u32x4 kk[NQK];
#pragma unroll
            for (int i = 0; i < NQK; ++i) kk[i] = kread(i);
#pragma unroll
            for (int i = 0; i < NQK; ++i) sacc[1][i % F64_NKB] = E::mfma(kk[i], qf[1][i / F64_NKB], sacc[1][i % F64_NKB]);
            exp_pack(0);
            __builtin_amdgcn_sched_group_barrier(0x100, FA_F64_PF, 0);
#pragma unroll
            for (int i = 0; i < NQK; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x402, FA_F64_VALU_PER_MFMA, 0);
            }
        }
        pin_p(0);
        __builtin_amdgcn_sched_barrier(0);
        if (msk[1]) mask_s(1);
        max_rescale(1);
        __builtin_amdgcn_sched_barrier(0);
        // C
        {
            u32x4 vv[NPV];
#pragma unroll
            for (int i = 0; i < NPV; ++i) vv[i] = vread(i);
#pragma unroll
            for (int i = 0; i < NPV; ++i) oacc[0][i % DBLKS] = E::mfma(vv[i], pf[0][i / DBLKS], oacc[0][i % DBLKS]);
            exp_pack(1);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * FA_F64_PF, 0);
#pragma unroll
            for (int i = 0; i < NPV; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x402, FA_F64_VALU_PER_MFMA, 0);
            }
        }
        pin_p(1);
        __builtin_amdgcn_sched_barrier(0);
        // D
        {
            u32x4 vv[NPV];
#pragma unroll
            for (int i = 0; i < FA_F64_PF; ++i) vv[i] = vread(i);
#pragma unroll
            for (int i = 0; i < NPV; ++i) {
                if (i + FA_F64_PF < NPV) vv[i + FA_F64_PF] = vread(i + FA_F64_PF);
                __builtin_amdgcn_sched_barrier(0);
                oacc[1][i % DBLKS] = E::mfma(vv[i], pf[1][i / DBLKS], oacc[1][i % DBLKS]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    auto tile_step = [&](auto stage_c, int nb) {
        constexpr int stage = decltype(stage_c)::value;
        if (nb + 1 < n_max) load_tile(nb + 1, std::integral_constant<int, stage ^ 1>{});
        compute_tile(stage_c, nb);
        __syncthreads();
    };
    if (n_min < n_max) load_tile(n_min, std::integral_constant<int, 0>{});
    __syncthreads();
    for (int nb = n_min; nb < n_max; nb += 2) {
        tile_step(std::integral_constant<int, 0>{}, nb);
        if (nb + 1 < n_max) tile_step(std::integral_constant<int, 1>{}, nb + 1);
    }

    // ---- epilogue: O / l, LSE ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // last MFMA -> accvgpr_read
#pragma unroll
    for (int rb = 0; rb < F64_RB; ++rb) {
        const float l_tot = xhalf_sum(l_run[rb]);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (my_row[rb] < seqlen_q) {
            uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + (p.cu_seqlens_q ? 0 : (int64_t)w.b * p.o_batch_stride)
                           + (q_row0 + my_row[rb]) * p.o_row_stride + (int64_t)w.h * p.o_head_stride;
#pragma unroll
            for (int d = 0; d < DBLKS; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    // (read through an "a" operand: a plain VALU use here makes hipcc copy all 128 accumulators
                    //  to VGPRs at the head of EVERY tile iteration)
                    float o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(o4[e]) : "a"(oacc[rb][d][4 * rq + e]));
                    u32x2 o2;
                    o2[0] = E::pack2(o4[0] * inv, o4[1] * inv);
                    o2[1] = E::pack2(o4[2] * inv, o4[3] * inv);
                    if (d * 32 + 8 * rq + 4 * g < dv) *reinterpret_cast<u32x2*>(op + d * 32 + 8 * rq + 4 * g) = o2;
                }
            if (g == 0) {
                const float lse = l_tot > 0.f ? (m_run[rb] + fast_log2(l_tot)) * kLn2 : -INFINITY;
                p.lse[(int64_t)w.b * p.lse_batch_stride + (int64_t)w.h * p.lse_head_stride + q_row0 + my_row[rb]] = lse;
            }
        }
    }
    }   // pass (the last tile step ended with a barrier: stage 0 is free again)
}

// Returns 1 when it launched, 0 when the caller should use fa_fwd_kernel.
int launch_fwd64(const KArgs& a0, hipStream_t stream) {
    static const bool enabled = [] { const char* e = getenv("FA_FWD64"); return !(e && e[0] == '0'); }();
    const fa_params& p = a0.p;
    if (!enabled || p.block_table || p.p_dropout > 0.f || a0.has_bias || p.head_dim != 128) return 0;
    if (p.seqlen_q < 2 * F64_BM) return 0;             // short query blocks: the 128-row kernel fills the chip better
    KArgs a = a0;
    a.n_qblocks_total = (p.seqlen_q + F64_BM - 1) / F64_BM;
    a.pair_qblocks = ((p.is_causal || p.window_right >= 0) && p.window_left < 0 && a.n_qblocks_total >= 2) ? 1 : 0;
    a.n_qblocks = a.pair_qblocks ? (a.n_qblocks_total + 1) / 2 : a.n_qblocks_total;
    const int grid = work_grid(p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
    if (grid == 0) return 1;
    const size_t smem = Fwd64Smem<128>::TOTAL;
    if (p.dtype == FA_BF16) {
        auto kern = fa_fwd64_kernel<bf16_tag, 128>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(F64_THREADS), smem, stream, a);
    } else {
        auto kern = fa_fwd64_kernel<fp16_tag, 128>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(F64_THREADS), smem, stream, a);
    }
    return 1;
}

}  // namespace fa
