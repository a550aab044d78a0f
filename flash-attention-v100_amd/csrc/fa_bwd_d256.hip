// fa_bwd_d256.hip - dK / dV of the backward at head dim 256 (and, DV = 192, head dims 129 .. 192) without bias / dropout:
// (softcap-only scores included) the two GEMM pairs of a 32 x 32 sub-tile on TWO waves.  Replaces the reference's uniform-in-D loop
// kernel/fused_mha_backward.cu:367-474 for D > 128 (everything else at that width stays on fa_bwd_dkdv_kernel in fa_bwd.hip).
#include <type_traits>
#include "fa_common.h"

namespace fa {

// A wave that owns 32 keys needs 2 x 128 accumulator registers (dK^T, dV^T) and 2 x 64 fragment registers (K, V) at
// D = 256: fa_bwd_dkdv_kernel<.., 256> therefore sweeps the query tiles TWICE (a 128-column half of dK / dV per sweep, S and
// dP recomputed: 96 MFMAs per 32 x 32 sub-tile instead of 64).  Here the work of a 32-key block is split by GEMM PAIR
// instead: eight waves per workgroup, wave kb (0..3) and wave kb + 4 share key block kb and sit on the same SIMD -
//     "P wave"  (role 0): K fragments in registers;  S = Q K^T, P = exp2(S c - lse2) -> LDS (fp32), dV^T += dO^T P
//     "dS wave" (role 1): V fragments in registers;  dP = dO V^T - D, dS = P o dP,                  dK^T += Q^T dS
// 128 accumulator + 64 fragment registers each, 32 MFMAs per wave and sub-tile, nothing recomputed; what crosses between
// the two is the 32 x 32 P tile (4 KiB of fp32 through LDS, the accumulator layout on both sides) behind the second of two
// barriers per stage.  Q / dO stages by LDS-DMA as in fa_bwd_dkdv2_kernel (all eight waves read the same stage: one LDS
// fragment per MFMA as everywhere else).  No bias, no dropout (those stay on fa_bwd_dkdv_kernel).
template <int D> struct DkvSplitSmem {
    static constexpr int BQ = 32;
    static constexpr int QT = BQ * D * 2;                // Q (or dO) stage
    static constexpr int STG = 2 * QT + 8 * BQ;          // Q, dO, row statistics: lse2[BQ] | -D[BQ]
    static constexpr int XB = 2 * STG;                   // P hand-off: 4 key blocks x [4 quads][64 lanes][16 B]
    static constexpr int TOTAL = XB + 4 * 4096;
};
constexpr int SPLIT_THREADS = 512;

// DV: columns that can be non-zero (192: head dims 129 .. 192 run on the 256-wide images - the DMA reads the missing columns as
// zeros - but skip the k-steps and the accumulator blocks that would only see them: 24 MFMAs per wave and sub-tile instead of 32)
// CAP: softcap without ALiBi (Gemma-2's head dim 256 form): the scores pass through cap tanh(s scale / cap) - only the P wave changes:
//      it keeps P for its dV and hands P (1 - tanh^2) to the dS wave (the chain-rule factor belongs to dS alone)
// PART: a launch smaller than the chip (fa_bwd.hip: dkv_split_factor): the workgroup walks a share of each pass's query tiles and
//       leaves an fp32 partial (its own instantiation: the full launch keeps its 9 spilled registers, this one has 13)
template <typename T, int D, int DV, bool CAP = false, bool PART = false>
__global__ void __launch_bounds__(SPLIT_THREADS, 1) fa_bwd_dkdv_split_kernel(const KArgs a) {
    using E = Elem<T>;
    static_assert(D == 256 && (DV == 256 || DV == 192), "two waves per key block: the D = 256 form");
    constexpr int KSTEPS = DV / 16;
    constexpr int DBLKS = DV / 32;
    constexpr int CPR = D / 8;
    constexpr int QT = DkvSplitSmem<D>::QT;
    constexpr int STG = DkvSplitSmem<D>::STG;
    constexpr int BQ = DkvSplitSmem<D>::BQ;
    constexpr int ROWS_PI = 64 / CPR;                        // rows per 1-KiB DMA instruction
    constexpr int Q_INSTS = BQ / ROWS_PI / 8;                // per wave, per tensor
    constexpr int SPLIT_PF = 2;                              // transposed fragments in flight ahead of their MFMA
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const stg_base = smem;
    char* const xb_base = smem + DkvSplitSmem<D>::XB;

    const fa_params& p = a.p;
    const int n_kblocks = (p.seqlen_k + DKV_BN - 1) / DKV_BN;
    const bool pair = a.pair_qblocks && n_kblocks >= 2 && !a.flat_kblocks;
    const int n_kb_grid = pair ? (n_kblocks + 1) / 2 : n_kblocks;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    int b, hk, nb0;
    const int nsplit = PART ? a.dkv_split : 1;               // dense launches smaller than the chip (fa_bwd.hip: dkv_split_factor)
    int split = 0;
    if (a.flat_kblocks) {
        int id = blockIdx.x;
        if (PART) { split = id % nsplit; id /= nsplit; }
        hk = id % p.nheads_k;
        flat_owner(id / p.nheads_k, DKV_BN, p.batch, p.cu_seqlens_k, lane, b, nb0);
        if (b < 0) return;
    } else {
        // (the splits of a key block are neighbours: same XCD, K / V from its L2)
        const UnitItem ui = decode_unit_item(blockIdx.x, p.batch * p.nheads_k, n_kb_grid * nsplit);
        if (!ui.valid) return;
        nb0 = ui.item / nsplit;
        split = ui.item - nb0 * nsplit;
        const int unit = ui.unit;
        b = unit / p.nheads_k; hk = unit - b * p.nheads_k;
    }
    const SeqGeom sg = seq_geom(p, b);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, kb = wave & 3;               // (waves kb and kb + 4: the same SIMD)
    const int group = p.nheads_q / p.nheads_k;
    const int off = sg.off;
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;
    const float c = a.scale_log2e;
    // softcap: cap tanh(s scale / cap) = cap (1 - 2 / (1 + exp2(s k1))) in log2 units (as in fa_bwd_dkdv2_kernel, BIAS == 3)
    const float cap_k1 = CAP ? 2.0f * a.scale_log2e / p.softcap : 0.f, cap_c2 = CAP ? p.softcap * kLog2e : 0.f;
    const int dv = valid_cols(p);

    uint32_t q_voff[Q_INSTS], do_voff[Q_INSTS];
#pragma unroll
    for (int i = 0; i < Q_INSTS; ++i) {
        const int row = (wave * Q_INSTS + i) * ROWS_PI + lane / CPR;
        const int cbs = swzt_row_off<D>(row, (lane % CPR) * 16) - row * D * 2;
        q_voff[i] = cbs < dv * 2 ? (uint32_t)(row * p.q_row_stride * 2 + cbs) : kOobVoff;
        do_voff[i] = cbs < dv * 2 ? (uint32_t)(row * p.do_row_stride * 2 + cbs) : kOobVoff;
    }
    const int64_t qb_off = p.cu_seqlens_q ? 0 : (int64_t)b * p.q_batch_stride;
    const int64_t dob_off = p.cu_seqlens_q ? 0 : (int64_t)b * p.do_batch_stride;
    const uint16_t* q_base = reinterpret_cast<const uint16_t*>(p.q) + qb_off + sg.q_row0 * p.q_row_stride;
    const uint16_t* do_base = reinterpret_cast<const uint16_t*>(p.dout) + dob_off + sg.q_row0 * p.do_row_stride;
    const float* lse_base = p.lse + (int64_t)b * p.lse_batch_stride + sg.q_row0;
    const float* dsum_base = p.softmax_d + (int64_t)b * p.lse_batch_stride + sg.q_row0;
    const int64_t kb_off = p.cu_seqlens_k ? 0 : (int64_t)b * p.k_batch_stride;
    const int64_t vb_off = p.cu_seqlens_k ? 0 : (int64_t)b * p.v_batch_stride;
    // the wave's fragment tensor: K (P wave) or V (dS wave)
    const uint16_t* f_head = role == 0
        ? reinterpret_cast<const uint16_t*>(p.k) + kb_off + sg.k_row0 * p.k_row_stride + (int64_t)hk * p.k_head_stride
        : reinterpret_cast<const uint16_t*>(p.v) + vb_off + sg.k_row0 * p.v_row_stride + (int64_t)hk * p.v_head_stride;
    const int64_t f_row_stride = role == 0 ? p.k_row_stride : p.v_row_stride;
    // lane-constant LDS read addresses (stage 0; the stage buffer, the upper 256 bytes of a row and the 16-row group of a
    // transposed read are immediates: the slot XOR of the `swzt` image touches byte bits 4..7 only).  The P wave reads Q by rows
    // and dO transposed, the dS wave dO by rows and Q transposed.
    const int x_tile = role == 0 ? 0 : QT, t_tile = role == 0 ? QT : 0;
    const lds_char* x_rp[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) x_rp[ks] = lds_pin(stg_base + x_tile + swzt_row_off<D>(l31, 32 * ks + 16 * g));
    const int rr = (lane & 15) >> 2;
    const int cb = (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);
    const lds_char* t_rp[2][4];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int d = 0; d < 4; ++d) t_rp[h2][d] = lds_pin(stg_base + t_tile + swzt_row_off<D>(4 * g + rr + 8 * h2, d * 64 + cb));
    // statistics in the accumulator layout: lse2[8 i + 4 g ..] (P wave), -D 4 BQ bytes further (dS wave)
    const lds_char* st_rp = lds_pin(stg_base + 2 * QT + 16 * g + (role == 0 ? 0 : 4 * BQ));
    char* const xb = xb_base + kb * 4096 + lane * 16;        // P hand-off: quad i at + 1024 i

    const int n_pass = (pair && (n_kblocks - 1 - nb0) != nb0) ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
    const int nb = pass == 0 ? nb0 : n_kblocks - 1 - nb0;
    const int n0 = nb * DKV_BN;
    if (n0 >= sg.seqlen_k) continue;

    const int kw0 = n0 + kb * 32;
    const int my_key = kw0 + l31;
    int qlo = 0, qhi = sg.seqlen_q - 1;
    if (wr >= 0) { const int t = my_key - off - wr; qlo = t > qlo ? t : qlo; }
    if (wl >= 0) { const int t = my_key - off + wl; qhi = t < qhi ? t : qhi; }
    if (my_key >= sg.seqlen_k) { qlo = 0x7fffffff; qhi = -1; }
    const int kw_last = (kw0 + 31 < sg.seqlen_k ? kw0 + 31 : sg.seqlen_k - 1);
    int w_qlo_min = 0, w_qlo_max = 0, w_qhi_min = sg.seqlen_q - 1, w_qhi_max = sg.seqlen_q - 1;
    if (wr >= 0) {
        const int t0 = kw0 - off - wr, t1 = kw_last - off - wr;
        w_qlo_min = t0 > 0 ? t0 : 0; w_qlo_max = t1 > 0 ? t1 : 0;
    }
    if (wl >= 0) {
        const int t0 = kw0 - off + wl, t1 = kw_last - off + wl;
        w_qhi_min = t0 < w_qhi_min ? t0 : w_qhi_min; w_qhi_max = t1 < w_qhi_max ? t1 : w_qhi_max;
    }
    const bool wave_has_keys = kw0 < sg.seqlen_k;
    const bool key_tail = kw0 + 31 >= sg.seqlen_k;
    int m_lo = 0, m_hi = sg.seqlen_q;
    {
        const int n_last = (n0 + DKV_BN < sg.seqlen_k ? n0 + DKV_BN : sg.seqlen_k) - 1;
        if (wr >= 0) { const int t = n0 - off - wr; m_lo = t > 0 ? t : 0; }
        if (wl >= 0) { const int t = n_last - off + wl + 1; m_hi = t < m_hi ? t : m_hi; }
    }
    int mt0 = m_lo / BQ;
    int mt1 = m_hi > m_lo ? (m_hi + BQ - 1) / BQ : mt0;
    if (PART) dkv_split_range(split, nsplit, mt0, mt1);               // this split's share of the pass's query tiles
    const int n_tiles = mt1 - mt0;
    const int n_iter = n_tiles * group;
    const bool empty = qhi < qlo;
    const int lo_l = empty ? 0x3fffffff : qlo - 4 * g;
    const uint32_t width = empty ? 0u : (uint32_t)(qhi - qlo);

    __syncthreads();                                         // the previous pass is done with the LDS
    u32x4 frag[KSTEPS];                                      // B operand of S (K) or dP (V): lane = key, 8 columns at 16 ks + 8 g
    {
        const uint16_t* fr = f_head + (int64_t)my_key * f_row_stride + 8 * g;
        const bool ok = my_key < sg.seqlen_k;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            u32x4 z = {0, 0, 0, 0};
            frag[ks] = (ok && 16 * ks + 8 * g < dv) ? *reinterpret_cast<const u32x4*>(fr + 16 * ks) : z;
        }
    }
    f32x16 acc[DBLKS];                                       // dV^T (P wave) or dK^T (dS wave)
#pragma unroll
    for (int d = 0; d < DBLKS; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

    __amdgpu_buffer_rsrc_t q_rsrc, do_rsrc;
    const float* stat_row = nullptr;
    // row statistics of a stage: wave 0 fetches lse, wave 4 softmax_d (a wave-uniform base each), lanes 0..31 publish
    const bool stat_wave = kb == 0;
    const bool stat_is_d = role == 1;
    auto set_head = [&](int gq) {
        const int h = hk * group + gq;
        q_rsrc = make_rsrc(q_base + (int64_t)h * p.q_head_stride, p.q_row_stride, sg.seqlen_q, dv);
        do_rsrc = make_rsrc(do_base + (int64_t)h * p.do_head_stride, p.do_row_stride, sg.seqlen_q, dv);
        if (stat_wave) stat_row = (stat_is_d ? dsum_base : lse_base) + (int64_t)h * p.lse_head_stride;
    };
    const uint32_t q_step = (uint32_t)(BQ * p.q_row_stride * 2), do_step = (uint32_t)(BQ * p.do_row_stride * 2);
    int gq_n = 0, mt_n = mt0;
    uint32_t q_soff = (uint32_t)mt0 * q_step, do_soff = (uint32_t)mt0 * do_step;
    float stat_next = 0.f;
    auto issue_stage = [&](auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        char* qd = stg_base + PAR * STG;
#pragma unroll
        for (int i = 0; i < Q_INSTS; ++i) {
            buf_load_lds_b128(q_rsrc, qd + (wave * Q_INSTS + i) * 1024, q_voff[i], q_soff);
            buf_load_lds_b128(do_rsrc, qd + QT + (wave * Q_INSTS + i) * 1024, do_voff[i], do_soff);
        }
        if (stat_wave) {
            const int qi = mt_n * BQ + l31;
            const int qc = qi < sg.seqlen_q ? qi : (sg.seqlen_q > 0 ? sg.seqlen_q - 1 : 0);
            stat_next = stat_row[qc];                          // (the raw value: see fa_bwd_dkdv2_kernel)
        }
    };
    auto publish_stats = [&](auto par_c, int mt_x) {
        constexpr int PAR = decltype(par_c)::value;
        if (stat_wave) {
            const float x = mt_x * BQ + l31 < sg.seqlen_q ? (stat_is_d ? -stat_next : stat_next * kLog2e) : 0.f;
            if (g == 0) reinterpret_cast<float*>(stg_base + PAR * STG + 2 * QT)[(stat_is_d ? BQ : 0) + l31] = x;
        }
    };
    auto advance_n = [&]() {
        ++mt_n; q_soff += q_step; do_soff += do_step;
        if (mt_n == mt1) {
            mt_n = mt0; ++gq_n;
            q_soff = (uint32_t)mt0 * q_step; do_soff = (uint32_t)mt0 * do_step;
            if (gq_n < group) set_head(gq_n);
        }
    };
    int mt = mt0;
    if (n_iter > 0) {
        set_head(0);
        issue_stage(std::integral_constant<int, 0>{});
        publish_stats(std::integral_constant<int, 0>{}, mt_n);
    }

    auto stage = [&](auto par_c, int it) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr int SB = PAR * STG;
        const int q0 = mt * BQ;
        __syncthreads();                                     // stage it landed, everyone left stage it - 1 (and its P tiles)
        advance_n();
        const bool has_next = it + 1 < n_iter;
        if (has_next) issue_stage(std::integral_constant<int, PAR ^ 1>{});
        const int mt_pub = mt_n;
        if (++mt == mt1) mt = mt0;
        const bool active = wave_has_keys && (q0 <= w_qhi_max) && (q0 + 31 >= w_qlo_min);
        f32x16 x;                                            // S (P wave) or dP - D (dS wave)
        u32x4 bf[2];                                         // packed P or dS: B operand of the second GEMM
        if (active) {
            // (the dS wave's accumulator starts from the -D quads; the P wave fetches its lse2 quads behind the GEMM)
            if (role == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 d4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(st_rp + (SB + 32 * i));
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[4 * i + e] = d4[e];
                }
            }
            {
                // row fragments two ahead of their MFMA, in this order (left alone hipcc keeps five in flight: 20 registers the
                // kernel does not have)
                u32x4 xa[KSTEPS];
                xa[0] = lds_read_b128(x_rp[0] + SB);
                xa[1] = lds_read_b128(x_rp[1] + SB);
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    if (ks + 2 < KSTEPS) xa[ks + 2] = lds_read_b128(x_rp[(ks + 2) & 7] + (SB + ((ks + 2) >> 3) * 256));
                    __builtin_amdgcn_sched_barrier(0);
                    x = E::mfma(xa[ks], frag[ks], x);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (role == 0) {
                const bool need_mask = key_tail || (q0 < w_qlo_max) || (q0 + 31 > w_qhi_min);
                // P (kept in x: this wave's dV operand) and what the dS wave multiplies its dP - D with (-> LDS).  Edge tiles: each wave
                // masks its own product below (both waves own the same keys: the same lane constants)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 l4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(st_rp + (SB + 32 * i));
                    f32x4 w4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * i + e;
                        float pr, chain = 1.0f;
                        if (CAP) {
                            const float rr1 = fast_rcp(1.0f + fast_exp2(x[r] * cap_k1));
                            const float t = fmaf(rr1, -2.0f, 1.0f);
                            pr = fast_exp2(fmaf(rr1, -2.0f * cap_c2, cap_c2) - l4[e]);
                            chain = fmaf(-t, t, 1.0f);
                        } else {
                            pr = fast_exp2(fmaf(x[r], c, -l4[e]));
                        }
                        x[r] = pr;
                        w4[e] = CAP ? pr * chain : pr;
                    }
                    *reinterpret_cast<__attribute__((address_space(3))) f32x4*>((lds_char*)(xb + 1024 * i)) = w4;
                }
                if (need_mask) {
                    const int lo_t = lo_l - q0;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((uint32_t)((r & 3) + 8 * (r >> 2) - lo_t) > width) x[r] = 0.f;
                }
            }
        }
        // the P tiles are written: wait for the LDS only (the next stage's DMA stays in flight), then the barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (active) {
            if (role == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((const lds_char*)(xb + 1024 * i));
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[4 * i + e] *= v4[e];
                }
                if (key_tail || (q0 < w_qlo_max) || (q0 + 31 > w_qhi_min)) {      // (the P wave hands its tile over unmasked)
                    const int lo_t = lo_l - q0;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((uint32_t)((r & 3) + 8 * (r >> 2) - lo_t) > width) x[r] = 0.f;
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) bf[t][w2] = E::pack2(x[8 * t + 2 * w2], x[8 * t + 2 * w2 + 1]);
            // ---- dV^T += dO^T P  /  dK^T += Q^T dS: transposed fragments SPLIT_PF ahead of their MFMA ----
            constexpr int NBK = 2 * DBLKS;
            auto tread = [&](int i) {
                const int t = i / DBLKS, d = i % DBLKS;
                const int o = SB + 16 * t * D * 2 + (d >> 2) * 256;
                const u32x2 a0 = lds_read_tr16_nw(t_rp[0][d & 3], o);
                const u32x2 a1 = lds_read_tr16_nw(t_rp[1][d & 3], o);
                return u32x4{a0[0], a0[1], a1[0], a1[1]};
            };
            u32x4 tf[NBK];
#pragma unroll
            for (int i = 0; i < SPLIT_PF && i < NBK; ++i) tf[i] = tread(i);
#pragma unroll
            for (int i = 0; i < NBK; ++i) {
                if (i + SPLIT_PF < NBK) tf[i + SPLIT_PF] = tread(i + SPLIT_PF);
                lds_tr_wait(tf[i], 2 * ((NBK - 1 - i) < SPLIT_PF ? (NBK - 1 - i) : SPLIT_PF));
                __builtin_amdgcn_sched_barrier(0);
                acc[i % DBLKS] = E::mfma(tf[i], bf[i / DBLKS], acc[i % DBLKS]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (has_next) publish_stats(std::integral_constant<int, PAR ^ 1>{}, mt_pub);
    };
#pragma unroll 1
    for (int it = 0; it < n_iter; it += 2) {
        stage(std::integral_constant<int, 0>{}, it);
        if (it + 1 < n_iter) stage(std::integral_constant<int, 1>{}, it + 1);
    }

    if (my_key < sg.seqlen_k) {
        // the P wave stores dV, the dS wave dK * softmax_scale
        // (addresses from opaque copies of the lane's key and half: nothing below is hoisted over the stage loop and parked in scratch)
        int key_e = my_key, g_e = g;
        asm volatile("" : "+v"(key_e), "+v"(g_e));
        const int64_t ob = p.cu_seqlens_k ? 0 : (int64_t)b * (role == 0 ? p.dv_batch_stride : p.dk_batch_stride);
        uint16_t* op = reinterpret_cast<uint16_t*>(role == 0 ? p.dv : p.dk) + ob +
                       (sg.k_row0 + key_e) * (role == 0 ? p.dv_row_stride : p.dk_row_stride) +
                       (int64_t)hk * (role == 0 ? p.dv_head_stride : p.dk_head_stride);
        const float sc = role == 0 ? 1.0f : p.softmax_scale;
        if (PART) {
            // fp32 partial of this split, [dK | dV][split][B][Sk][Hk][D]: dkv_reduce_kernel adds the splits and rounds once
            const int64_t row = (int64_t)p.nheads_k * D;
            const int64_t slab = (p.cu_seqlens_k ? (int64_t)p.total_k : (int64_t)p.batch * p.seqlen_k) * row;
            const int64_t krow = p.cu_seqlens_k ? sg.k_row0 + key_e : (int64_t)b * p.seqlen_k + key_e;
            float* pp = reinterpret_cast<float*>(a.dkv_part) + ((role == 0 ? nsplit : 0) + split) * slab + krow * row + (int64_t)hk * D;
#pragma unroll
            for (int d = 0; d < DBLKS; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 o4 = {acc[d][4 * rq + 0] * sc, acc[d][4 * rq + 1] * sc, acc[d][4 * rq + 2] * sc, acc[d][4 * rq + 3] * sc};
                    if (d * 32 + 8 * rq + 4 * g_e < dv) *reinterpret_cast<f32x4*>(pp + d * 32 + 8 * rq + 4 * g_e) = o4;
                }
        } else
#pragma unroll
        for (int d = 0; d < DBLKS; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u32x2 o2;
                o2[0] = E::pack2(acc[d][4 * rq + 0] * sc, acc[d][4 * rq + 1] * sc);
                o2[1] = E::pack2(acc[d][4 * rq + 2] * sc, acc[d][4 * rq + 3] * sc);
                if (d * 32 + 8 * rq + 4 * g_e < dv) *reinterpret_cast<u32x2*>(op + d * 32 + 8 * rq + 4 * g_e) = o2;
            }
    }
    }   // pass
}


template <typename T>
static int launch_split_t(const KArgs& a, int grid, hipStream_t stream) {
    constexpr int D = 256;
    const size_t smem = DkvSplitSmem<D>::TOTAL;
#define FA_LAUNCH_SPLIT(DV_, CAP_)                                                              \
    do {                                                                                        \
        if (a.dkv_split > 1) {                                                                  \
            auto kern = fa_bwd_dkdv_split_kernel<T, D, DV_, CAP_, true>;                        \
            FA_SET_LDS_ONCE(kern, smem);                                                        \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(SPLIT_THREADS), smem, stream, a);         \
        } else {                                                                                \
            auto kern = fa_bwd_dkdv_split_kernel<T, D, DV_, CAP_>;                              \
            FA_SET_LDS_ONCE(kern, smem);                                                        \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(SPLIT_THREADS), smem, stream, a);         \
        }                                                                                       \
    } while (0)
    const bool cap = a.p.softcap > 0.f;
    if (valid_cols(a.p) <= 192) { if (cap) FA_LAUNCH_SPLIT(192, true); else FA_LAUNCH_SPLIT(192, false); }
    else                        { if (cap) FA_LAUNCH_SPLIT(256, true); else FA_LAUNCH_SPLIT(256, false); }
#undef FA_LAUNCH_SPLIT
    return 0;
}

// grid: the dK/dV kernels' (8 x ceil(units / 8) x key blocks, or the flat list of key blocks of packed sequences)
int launch_bwd_dkdv_split(const KArgs& a, int grid, hipStream_t stream) {
    if (grid <= 0) return 0;
    return a.p.dtype == FA_BF16 ? launch_split_t<bf16_tag>(a, grid, stream) : launch_split_t<fp16_tag>(a, grid, stream);
}

}  // namespace fa
