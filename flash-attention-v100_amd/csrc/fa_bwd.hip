// fa_bwd.hip - fused attention backward for gfx950 (dense + varlen).
//
// Replaces kernel/fused_mha_backward.cu:26-489 and kernel/fused_mha_backward_varlen.cu:26-540
// of the reference.  Like the reference the backward is atomic-free and deterministic and
// is split by the output it owns (reference: blockIdx.y == 0 -> dQ, == 1 -> dK/dV,
// fused_mha_backward.cu:58,257); here those are three launches on one stream:
//   1. bwd_preprocess   softmax_d[b,h,i] = sum_d O[i,d] dO[i,d]   (include/product.h:72-94;
//                       computed ONCE, the reference recomputes it per Q tile in the dKV phase)
//   2. bwd_dkdv         a workgroup owns 128 keys (32 per wave, K/V fragments in registers),
//                       streams Q/dO tiles through LDS, accumulates dK^T, dV^T in registers
//                       across the q-heads of its kv-head (GQA sum in-kernel, like
//                       fused_mha_backward.cu:351)
//   3. bwd_dq           a workgroup owns 128 query rows (Q/dO fragments in registers),
//                       streams K/V tiles through LDS, accumulates dQ^T in registers.
// Math (include/softmax.h:282-314):  P = exp(S - LSE), dP = dO V^T,
//   dS = P o (dP - D) * scale  [softcap: * (1 - (S/c)^2)],  dV = P^T dO, dK = dS^T Q, dQ = dS K.
// Every GEMM runs on v_mfma_f32_32x32x16; P and dS are rounded to the 16-bit io type before
// their GEMMs (as the reference does) and the MFMA C-layout is reused as the next B operand.
#include <cstdlib>
#include <type_traits>
#include <cstring>
#include "fa_common.h"

namespace fa {

constexpr int BWD_THREADS = 256;

// ---------------------------------------------------------------------------------------------
// 1. preprocess: D_i = rowsum(O o dO)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) bwd_preprocess_kernel(const KArgs a) {
    using E = Elem<T>;
    const fa_params& p = a.p;
    const int cpr = p.head_dim / 8;                     // lanes per row (8 or 16)
    const int rows_per_block = 256 / cpr;
    const int64_t total_rows = p.cu_seqlens_q ? (int64_t)p.total_q : (int64_t)p.batch * p.seqlen_q;
    const int64_t row = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / cpr;
    const int h = blockIdx.y;
    const int cc = threadIdx.x % cpr;
    float acc = 0.f;
    int64_t b = 0, i = row;
    if (!p.cu_seqlens_q) { b = row / p.seqlen_q; i = row - b * p.seqlen_q; }
    if (row < total_rows && cc * 8 < valid_cols(p)) {
        const uint16_t* op = reinterpret_cast<const uint16_t*>(p.o) + b * p.o_batch_stride + i * p.o_row_stride +
                             (int64_t)h * p.o_head_stride + cc * 8;
        const uint16_t* dp = reinterpret_cast<const uint16_t*>(p.dout) + b * p.do_batch_stride + i * p.do_row_stride +
                             (int64_t)h * p.do_head_stride + cc * 8;
        const u32x4 ov = *reinterpret_cast<const u32x4*>(op);
        const u32x4 dv = *reinterpret_cast<const u32x4*>(dp);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc = fmaf(E::lo(ov[j]), E::lo(dv[j]), acc);
            acc = fmaf(E::hi(ov[j]), E::hi(dv[j]), acc);
        }
    }
    for (int m = cpr >> 1; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (row < total_rows && cc == 0) {
        p.softmax_d[b * p.lse_batch_stride + (int64_t)h * p.lse_head_stride + i] = acc;
        if (a.stats_ws) {           // dense only: the asm dK/dV kernel streams both statistics from one workspace
            const float lse = p.lse[b * p.lse_batch_stride + (int64_t)h * p.lse_head_stride + i];
            const int64_t plane = (int64_t)p.batch * p.nheads_q * p.seqlen_q;
            const int64_t at = (b * p.nheads_q + h) * p.seqlen_q + i;
            a.stats_ws[at] = lse == -INFINITY ? INFINITY : lse * kLog2e;      // P = exp2(S c - lse2) = 0 for rows without keys
            a.stats_ws[plane + at] = -acc;                                    // the dP accumulators start from -D
        }
    }
}

// ---------------------------------------------------------------------------------------------
// shared geometry
// ---------------------------------------------------------------------------------------------
// (SeqGeom / seq_geom / DKV_BN: fa_common.h - shared with fa_bwd_d256.hip)

// ---------------------------------------------------------------------------------------------
// 2. dK / dV
// ---------------------------------------------------------------------------------------------
#ifdef FA_TIMERS
// development aid (variant builds only): per-phase s_memtime sums of wave 0 of workgroup FA_TIMERS
__device__ unsigned long long g_timers[16];
#define TMR_NOW() __builtin_readcyclecounter()
#define TMR_ADD(i, t0) do { const unsigned long long t1_ = TMR_NOW(); tmr[i] += t1_ - (t0); (t0) = t1_; } while (0)
#else
#define TMR_NOW() 0ull
#define TMR_ADD(i, t0) do { } while (0)
#endif
#ifndef FA_DKV_BQ
#define FA_DKV_BQ 64
#endif
constexpr int DKV_BQ = FA_DKV_BQ;      // query rows per LDS stage

template <int D> struct DkvSmem {
    // D = 256: 32-row stages leave 64 KiB in which the waves park their K fragments (see kpark in the kernel)
    static constexpr int BQ = D > 128 ? 32 : DKV_BQ;
    static constexpr int TILE = BQ * D * 2;             // one Q (or dO) tile
    static constexpr int STATS = BQ * 4 * 2;            // lse2 + D, fp32
    static constexpr int STAGE = 2 * TILE + STATS;
    static constexpr int KPARK = D > 128 ? DKV_BN * D * 2 : 0;
    static constexpr int TOTAL = 2 * STAGE + KPARK;
};

// BIAS: 0 none, 1 general (ALiBi / softcap per element), 2 causal ALiBi through the matrix pipe (fa_common.h)
template <typename T, int D, int BIAS, bool DROPOUT>
__global__ void __launch_bounds__(BWD_THREADS, 1) fa_bwd_dkdv_kernel(const KArgs a) {
    using E = Elem<T>;
    constexpr int KSTEPS = D / 16;
    constexpr int DBLKS = D / 32;
    // D = 256: dK / dV are produced in two 128-column halves (two sweeps over the query tiles, S and dP
    // recomputed) - 256 accumulator registers + 128 K/V fragment registers do not fit beside the rest
    // (the one-sweep build spilled 400-900 registers: 10 ms -> see DESIGN.md)
#ifndef FA_DKV_NDH256
#define FA_DKV_NDH256 2
#endif
    constexpr int NDH = D > 128 ? FA_DKV_NDH256 : 1;
    constexpr int ADB = DBLKS / NDH;                     // accumulated 32-column blocks per sweep
    constexpr int CPR = D / 8;
    constexpr int BQ = DkvSmem<D>::BQ;
    constexpr int CHUNKS = BQ * CPR / BWD_THREADS;
    constexpr int TILE = DkvSmem<D>::TILE;
    constexpr int STAGE = DkvSmem<D>::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const fa_params& p = a.p;
    unsigned long long tmr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)tmr;
    const int n_kblocks = (p.seqlen_k + DKV_BN - 1) / DKV_BN;
    // causal load balance: key block i is paired with its mirror (heavy + light = constant)
    const bool pair = a.pair_qblocks && n_kblocks >= 2;
    const int n_kb_grid = pair ? (n_kblocks + 1) / 2 : n_kblocks;
    int b, hk, nb0;
    {
        const UnitItem ui = decode_unit_item(blockIdx.x, p.batch * p.nheads_k, n_kb_grid);
        if (!ui.valid) return;
        nb0 = ui.item;
        const int unit = ui.unit;
        b = unit / p.nheads_k; hk = unit - b * p.nheads_k;
    }
    const SeqGeom sg = seq_geom(p, b);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = p.nheads_q / p.nheads_k;
    const int off = sg.off;
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;
    const float c = a.scale_log2e;
    // softcap-only variant (BIAS == 3): cap tanh(s scale / cap) = cap (1 - 2 / (1 + exp2(s k1))), log2 units
    const float cap_k1 = p.softcap > 0.f ? 2.0f * a.scale_log2e / p.softcap : 0.f, cap_c2 = p.softcap * kLog2e;
    (void)cap_k1; (void)cap_c2;
    DropCtx dc = {0, 0, 0, 0};
    if (DROPOUT) {
        dc.k0 = (uint32_t)p.philox_seed; dc.k1 = (uint32_t)(p.philox_seed >> 32);
        dc.thr = a.drop_thr; dc.offset = p.philox_offset;
    }
    const uint64_t drop_n_glob = (uint64_t)p.seqlen_k;

    const int dv = valid_cols(p);
    // loop-invariant staging geometry
#ifndef FA_DKV_DMA
    // Q / dO tiles staged through registers (buffer_load -> ds_write after the MFMAs): measured
    // 13 % faster than LDS-DMA for this one-wave-per-SIMD kernel (1.54 vs 1.74 ms), while
    // LDS-DMA wins in the two-wave kernels (fwd, dQ).  D = 256 has no registers to stage through.
#ifndef FA_DKV_DMA256
#define FA_DKV_DMA256 1
#endif
    constexpr bool DMA = D > 128 && FA_DKV_DMA256;
#else
    // Q / dO tiles by LDS-DMA (see fa_fwd.hip): instruction `inst` = wave*CHUNKS + i covers
    // ROWS_PI rows, lane -> (row, physical slot); the source offset carries the swizzle.
    constexpr bool DMA = true;
#endif
    uint32_t q_voff[CHUNKS], do_voff[CHUNKS];
    int t_lds[CHUNKS];
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        if (DMA) {
            constexpr int ROWS_PI = 64 / CPR;
            const int inst = wave * CHUNKS + i;
            const int row = inst * ROWS_PI + lane / CPR;
            const int cbs = swzt_row_off<D>(row, (lane % CPR) * 16) - row * D * 2;
            q_voff[i] = cbs < dv * 2 ? (uint32_t)(row * p.q_row_stride * 2 + cbs) : kOobVoff;
            do_voff[i] = cbs < dv * 2 ? (uint32_t)(row * p.do_row_stride * 2 + cbs) : kOobVoff;
            t_lds[i] = inst * 1024;
        } else {
            const int cidx = tid + i * BWD_THREADS;
            const int row = cidx / CPR, cc = cidx % CPR;
            q_voff[i] = cc * 8 < dv ? (uint32_t)(row * p.q_row_stride + cc * 8) * 2u : kOobVoff;
            do_voff[i] = cc * 8 < dv ? (uint32_t)(row * p.do_row_stride + cc * 8) * 2u : kOobVoff;
            t_lds[i] = swzt_row_off<D>(row, cc * 16);
        }
    }
    const int64_t qb_off = p.cu_seqlens_q ? 0 : (int64_t)b * p.q_batch_stride;
    const int64_t dob_off = p.cu_seqlens_q ? 0 : (int64_t)b * p.do_batch_stride;
    const uint16_t* q_base = reinterpret_cast<const uint16_t*>(p.q) + qb_off + sg.q_row0 * p.q_row_stride;
    const uint16_t* do_base = reinterpret_cast<const uint16_t*>(p.dout) + dob_off + sg.q_row0 * p.do_row_stride;
    const float* lse_base = p.lse + (int64_t)b * p.lse_batch_stride + sg.q_row0;
    const float* dsum_base = p.softmax_d + (int64_t)b * p.lse_batch_stride + sg.q_row0;
    // lane-constant LDS read offsets
    int a_rd[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) a_rd[ks] = swzt_row_off<D>(l31, 32 * ks + 16 * g);
    const int rr = (lane & 15) >> 2;
    const int cb = (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);

    const int n_pass = (pair && (n_kblocks - 1 - nb0) != nb0) ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
    const int nb = pass == 0 ? nb0 : n_kblocks - 1 - nb0;
    const int n0 = nb * DKV_BN;
    if (n0 >= sg.seqlen_k) continue;

    const int kw0 = n0 + wave * 32;
    const int my_key = kw0 + l31;
    // visible queries of my key: qlo <= i <= qhi      (j <= i + off + wr ; j >= i + off - wl)
    int qlo = 0, qhi = sg.seqlen_q - 1;
    if (wr >= 0) { const int t = my_key - off - wr; qlo = t > qlo ? t : qlo; }
    if (wl >= 0) { const int t = my_key - off + wl; qhi = t < qhi ? t : qhi; }
    if (my_key >= sg.seqlen_k) { qlo = 0x7fffffff; qhi = -1; }
    // wave-uniform query bounds for skipping / mask elision
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
    const bool key_tail = kw0 + 31 >= sg.seqlen_k;       // some lanes of this wave hold no key

    // query-tile range of the whole 128-key block
    int m_lo = 0, m_hi = sg.seqlen_q;                    // [m_lo, m_hi)
    {
        const int n_last = (n0 + DKV_BN < sg.seqlen_k ? n0 + DKV_BN : sg.seqlen_k) - 1;
        if (wr >= 0) { const int t = n0 - off - wr; m_lo = t > 0 ? t : 0; }
        if (wl >= 0) { const int t = n_last - off + wl + 1; m_hi = t < m_hi ? t : m_hi; }
    }
    const int mt0 = m_lo / BQ;
    const int mt1 = m_hi > m_lo ? (m_hi + BQ - 1) / BQ : mt0;
    const int n_tiles = mt1 - mt0;
    const int n_iter = n_tiles * group;

    // ---- K, V fragments of my 32 keys: B operands, lane holds X[my_key][16ks + 8g .. +7] ----
    // D = 256: 128 K + V fragment registers beside 128 accumulators spilled ~80 registers to scratch.  The K fragments
    // are parked in LDS instead, in exactly the register image (instruction ks = one contiguous 1-KiB line, lane l at
    // 16 l: conflict-free, no swizzle); only the owning wave touches its 16 KiB, so no barrier is involved.
    constexpr bool KPARK = DkvSmem<D>::KPARK > 0;
    u32x4 kf[KPARK ? 1 : KSTEPS], vf[KSTEPS];
    char* kpark = smem + 2 * STAGE + wave * (32 * D * 2) + lane * 16;
    (void)kpark;
    {
        const int64_t kb_off = p.cu_seqlens_k ? 0 : (int64_t)b * p.k_batch_stride;
        const int64_t vb_off = p.cu_seqlens_k ? 0 : (int64_t)b * p.v_batch_stride;
        const uint16_t* kr = reinterpret_cast<const uint16_t*>(p.k) + kb_off + (sg.k_row0 + my_key) * p.k_row_stride +
                             (int64_t)hk * p.k_head_stride + 8 * g;
        const uint16_t* vr = reinterpret_cast<const uint16_t*>(p.v) + vb_off + (sg.k_row0 + my_key) * p.v_row_stride +
                             (int64_t)hk * p.v_head_stride + 8 * g;
        const bool ok = my_key < sg.seqlen_k;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            u32x4 z = {0, 0, 0, 0};
            const bool okc = ok && 16 * ks + 8 * g < dv;
            const u32x4 kx = okc ? *reinterpret_cast<const u32x4*>(kr + 16 * ks) : z;
            if constexpr (KPARK) lds_write_b128(kpark + ks * 1024, kx);
            else kf[ks] = kx;
            vf[ks] = okc ? *reinterpret_cast<const u32x4*>(vr + 16 * ks) : z;
        }
    }

    // ---- staging of Q / dO / lse / D tiles (rows past seqlen_q read as zero) ----
    u32x4 qreg[CHUNKS], doreg[CHUNKS];
    float statreg = 0.f;
    // (measured with s_memtime, tools/read_timers.py: the 8 buffer_loads cost the wave ~900 issue
    //  cycles per step; spreading them in slices between the MFMA phases made the kernel 25 %
    //  SLOWER - the extra control flow costs more registers than this kernel has - so they stay
    //  one burst at the top of the step.)
    auto load_tile = [&](int it, auto stage_c) {
        constexpr int stage = decltype(stage_c)::value;
        const int gq = group == 1 ? 0 : it / n_tiles;     // (a scalar division costs ~150 cycles per step)
        const int m0 = (mt0 + it - gq * n_tiles) * BQ;
        const int h = hk * group + gq;
        const __amdgpu_buffer_rsrc_t q_rsrc = make_rsrc(q_base + (int64_t)h * p.q_head_stride, p.q_row_stride, sg.seqlen_q, dv);
        const __amdgpu_buffer_rsrc_t do_rsrc = make_rsrc(do_base + (int64_t)h * p.do_head_stride, p.do_row_stride, sg.seqlen_q, dv);
        const uint32_t q_soff = (uint32_t)(m0 * p.q_row_stride * 2);
        const uint32_t do_soff = (uint32_t)(m0 * p.do_row_stride * 2);
        if (DMA) {
            char* qs = smem + stage * STAGE;
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(q_rsrc, qs + t_lds[i], q_voff[i], q_soff);
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(do_rsrc, qs + TILE + t_lds[i], do_voff[i], do_soff);
        } else {
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) qreg[i] = buf_load_b128(q_rsrc, q_voff[i], q_soff);
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) doreg[i] = buf_load_b128(do_rsrc, do_voff[i], do_soff);
        }
        if (tid < 2 * BQ) {
            const int r = tid & (BQ - 1);
            const int qi = m0 + r;
            statreg = 0.f;
            if (qi < sg.seqlen_q) {
                // (no arithmetic on the loaded value here: a use would put an s_waitcnt vmcnt(0) - the full latency of the
                //  tile loads issued just above - at the top of the step; the log2(e) factor is applied in store_tile)
                statreg = (tid < BQ ? lse_base : dsum_base)[(int64_t)h * p.lse_head_stride + qi];
            }
        }
    };
    auto store_tile = [&](auto stage_c) {
        constexpr int stage = decltype(stage_c)::value;
        char* qs = smem + stage * STAGE;
        char* dos = qs + TILE;
        float* st = reinterpret_cast<float*>(dos + TILE);
        if (!DMA) {
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) {
                lds_write_b128(qs + t_lds[i], qreg[i]);
                lds_write_b128(dos + t_lds[i], doreg[i]);
            }
        }
        if (tid < 2 * BQ) st[tid] = tid < BQ ? statreg * kLog2e : statreg;     // [0,BQ): lse2, [BQ,2BQ): D
    };

    f32x16 dk_acc[ADB], dv_acc[ADB];
    int dh_off = 0;                                      // byte offset of the sweep's columns in a Q / dO row

    float slope = 0.f;

    // ---- the three phases of one 32-query x 32-key sub-tile of this wave ----
    // sd: S = Q K^T, dP = dO V^T : acc[r] = X[q0 + row(r,g)][my_key]
    u32x4 alibi_b = {0, 0, 0, 0};                          // BIAS == 2: per sub-tile, set in compute()
    const u32x4 alibi_a = alibi_pos_operand<T>(lane);      // A side = query rows: position of the register
    auto sd = [&](const char* qs, const char* dos, int sub, f32x16& s_acc, f32x16& dp_acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s_acc[r] = 0.f; dp_acc[r] = 0.f; }
        if (BIAS == 2) s_acc = E::mfma(alibi_a, alibi_b, s_acc);
        // (measured: alternating the S and dP chains is 6 % SLOWER here - one wave per SIMD -
        //  while it is 10 % faster in the two-wave dQ kernel)
#ifndef FA_DKV_NO_PREFETCH
        if constexpr (D <= 128) {
        // one wave per SIMD: nothing else hides the LDS latency, so all Q fragments are fetched
        // up front and the dO fragments stream in behind the S MFMAs
        u32x4 qa[KSTEPS], da[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) qa[ks] = lds_read_b128(qs + a_rd[ks] + sub * 32 * D * 2);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            da[ks] = lds_read_b128(dos + a_rd[ks] + sub * 32 * D * 2);
            s_acc = E::mfma(qa[ks], kf[ks], s_acc);
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) dp_acc = E::mfma(da[ks], vf[ks], dp_acc);
        __builtin_amdgcn_sched_group_barrier(0x100, KSTEPS, 0);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, KSTEPS, 0);
        } else
#endif
        {
        // D = 256: the compiler otherwise reuses one register quad per operand and waits for every LDS read in front of its
        // MFMA (read, lgkmcnt(0), MFMA: ~70 cycles per MFMA).  A ring of PF operand pairs keeps PF reads in flight.
#ifndef FA_DKV_PF256
#define FA_DKV_PF256 4
#endif
        constexpr int PF = FA_DKV_PF256, NJ = 2 * KSTEPS;
        u32x4 ra[PF], rb[PF];
        auto fetch = [&](int j, u32x4& a_, u32x4& b_) {          // j < KSTEPS: S = Q K^T steps, then dP = dO V^T steps
            const int ks = j % KSTEPS;
            a_ = lds_read_b128((j < KSTEPS ? qs : dos) + a_rd[ks] + sub * 32 * D * 2);
            if (j < KSTEPS) {
                if constexpr (KPARK) b_ = lds_read_b128(kpark + ks * 1024);
                else b_ = kf[ks];
            }
        };
#pragma unroll
        for (int j = 0; j < PF; ++j) fetch(j, ra[j], rb[j]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j < KSTEPS) s_acc = E::mfma(ra[j % PF], rb[j % PF], s_acc);
            else dp_acc = E::mfma(ra[j % PF], vf[j - KSTEPS], dp_acc);
            if (j + PF < NJ) fetch(j + PF, ra[j % PF], rb[j % PF]);
        }
        if (KPARK) __builtin_amdgcn_sched_group_barrier(0x100, 2 * PF, 0);
        else __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (KPARK && j + PF < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            else if (j + PF < NJ) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        }
    };
    // Dropout keep bits of this lane's 16 accumulator registers.  The four lanes of a quad hold four
    // consecutive keys = ONE Philox counter per query row (include/softmax.h:97-104), so each lane draws the
    // rows r = 4 i + (lane & 3) only (keep bits of all four keys) and the quad exchanges them with DPP:
    // 4 Philox calls per lane and sub-tile instead of 16.  Returns bit r = keep(register r).
    auto drop_bits = [&](int q0) -> uint32_t {
        uint32_t mine = 0;
        const int lq = lane & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qi = q0 + lq + 8 * i + 4 * g;          // register r = 4 i + lq
            mine |= dropout_keep4(dc, (uint64_t)(sg.q_row0 + qi) * drop_n_glob + (uint64_t)(my_key & ~3)) << (4 * i);
        }
        const int kq = my_key & 3;
        uint32_t bits = 0;
        const uint32_t a0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x00, 0xf, 0xf, false);   // quad_perm [0,0,0,0]
        const uint32_t a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x55, 0xf, 0xf, false);   // [1,1,1,1]
        const uint32_t a2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xaa, 0xf, 0xf, false);   // [2,2,2,2]
        const uint32_t a3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xff, 0xf, 0xf, false);   // [3,3,3,3]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bits |= ((a0 >> (4 * i + kq)) & 1u) << (4 * i + 0);
            bits |= ((a1 >> (4 * i + kq)) & 1u) << (4 * i + 1);
            bits |= ((a2 >> (4 * i + kq)) & 1u) << (4 * i + 2);
            bits |= ((a3 >> (4 * i + kq)) & 1u) << (4 * i + 3);
        }
        return bits;
    };
    // row statistics for q = q0 + 8 i + 4 g + (0..3)
    auto sm_stats = [&](const float* st, int sub, f32x4 (&lse2)[4], f32x4 (&dsum)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lse2[i] = *reinterpret_cast<const f32x4*>(st + sub * 32 + 8 * i + 4 * g);
            dsum[i] = *reinterpret_cast<const f32x4*>(st + BQ + sub * 32 + 8 * i + 4 * g);
        }
    };
    // sm: P = exp2(S c - lse2), dS = P (dP - D), rounded to 16 bit as the B operands of phase bk
    auto sm = [&](int q0, bool need_mask, const f32x4 (&lse2)[4], const f32x4 (&dsum)[4],
                  const f32x16& s_acc, const f32x16& dp_acc, u32x4 (&pf)[2], u32x4 (&dsf)[2]) {
        float pv[16], dsv[16];
        const uint32_t kbits = DROPOUT ? drop_bits(q0) : 0xffffu;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float l2 = lse2[r >> 2][r & 3];
            const int qi = q0 + (r & 3) + 8 * (r >> 2) + 4 * g;
            float dpe = dp_acc[r];
            bool keep = true;
            if (DROPOUT) {
                // dS = P (keep rp dP - D); dV accumulates keep P (rp applied in the epilogue)
                keep = (kbits >> r) & 1u;
                dpe = keep ? dpe * a.rp_dropout : 0.f;
            }
            float pr, dsr;
            if (BIAS == 1) {
                float sv = s_acc[r] * p.softmax_scale;
                sv = fmaf(-slope, fabsf((float)(qi + off - my_key)), sv);
                float chain = 1.f;
                if (p.softcap > 0.f) {
                    const float t = fast_tanh(sv / p.softcap);
                    sv = p.softcap * t;
                    chain = 1.f - t * t;
                }
                pr = fast_exp2(fmaf(sv, kLog2e, -l2));
                dsr = pr * (dpe - dsum[r >> 2][r & 3]) * chain;
            } else if (BIAS == 3) {
                const float rr1 = fast_rcp(1.0f + fast_exp2(s_acc[r] * cap_k1));
                const float t = fmaf(rr1, -2.0f, 1.0f);                      // tanh(s scale / cap)
                pr = fast_exp2(fmaf(rr1, -2.0f * cap_c2, cap_c2) - l2);
                dsr = pr * (dpe - dsum[r >> 2][r & 3]) * fmaf(-t, t, 1.0f);
            } else {
                pr = fast_exp2(fmaf(s_acc[r], c, -l2));
                dsr = pr * (dpe - dsum[r >> 2][r & 3]);
            }
            pv[r] = (DROPOUT && !keep) ? 0.f : pr;
            dsv[r] = dsr;
        }
        if (need_mask) {
            const bool empty = qhi < qlo;                  // folded into the operands (see fa_fwd.hip)
            const int lo_t = empty ? 0x3fffffff : qlo - q0 - 4 * g;
            const uint32_t width = empty ? 0u : (uint32_t)(qhi - qlo);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cpos = (r & 3) + 8 * (r >> 2);
                if ((uint32_t)(cpos - lo_t) > width) { pv[r] = 0.f; dsv[r] = 0.f; }
            }
        }
        // k-step t of phase bk covers regs 8t .. 8t+7
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                pf[t][w2] = E::pack2(pv[8 * t + 2 * w2], pv[8 * t + 2 * w2 + 1]);
                dsf[t][w2] = E::pack2(dsv[8 * t + 2 * w2], dsv[8 * t + 2 * w2 + 1]);
            }
    };
    // sm for ONE group of four accumulator registers (rows 8 i + 4 g + 0..3): the register-lean form
    uint32_t kbits_rows = 0xffffu;                         // set per sub-tile by the caller when DROPOUT
    auto sm_rows = [&](int i, int q0, bool need_mask, const f32x4& l2v, const f32x4& dsm, const f32x16& s_acc,
                       const f32x16& dp_acc, u32x4 (&pf)[2], u32x4 (&dsf)[2]) {
        float pv[4], dsv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * i + e;
            const int qi = q0 + e + 8 * i + 4 * g;
            float dpe = dp_acc[r];
            bool keep = true;
            if (DROPOUT) {
                keep = (kbits_rows >> r) & 1u;
                dpe = keep ? dpe * a.rp_dropout : 0.f;
            }
            float pr, dsr;
            if (BIAS == 1) {
                float sv = s_acc[r] * p.softmax_scale;
                sv = fmaf(-slope, fabsf((float)(qi + off - my_key)), sv);
                float chain = 1.f;
                if (p.softcap > 0.f) {
                    const float t = fast_tanh(sv / p.softcap);
                    sv = p.softcap * t;
                    chain = 1.f - t * t;
                }
                pr = fast_exp2(fmaf(sv, kLog2e, -l2v[e]));
                dsr = pr * (dpe - dsm[e]) * chain;
            } else if (BIAS == 3) {
                const float rr1 = fast_rcp(1.0f + fast_exp2(s_acc[r] * cap_k1));
                const float t = fmaf(rr1, -2.0f, 1.0f);
                pr = fast_exp2(fmaf(rr1, -2.0f * cap_c2, cap_c2) - l2v[e]);
                dsr = pr * (dpe - dsm[e]) * fmaf(-t, t, 1.0f);
            } else {
                pr = fast_exp2(fmaf(s_acc[r], c, -l2v[e]));
                dsr = pr * (dpe - dsm[e]);
            }
            float pk = (DROPOUT && !keep) ? 0.f : pr;
            if (need_mask && (qi < qlo || qi > qhi)) { pk = 0.f; dsr = 0.f; }
            pv[e] = pk; dsv[e] = dsr;
        }
        pf[i >> 1][2 * (i & 1)] = E::pack2(pv[0], pv[1]);
        pf[i >> 1][2 * (i & 1) + 1] = E::pack2(pv[2], pv[3]);
        dsf[i >> 1][2 * (i & 1)] = E::pack2(dsv[0], dsv[1]);
        dsf[i >> 1][2 * (i & 1) + 1] = E::pack2(dsv[2], dsv[3]);
    };
    // bk: dV^T += dO^T P,  dK^T += Q^T dS
    auto bk = [&](const char* qs, const char* dos, int sub, const u32x4 (&pf)[2], const u32x4 (&dsf)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // rows sub*32 + 16 t + 8 hf + 4 g + rr ; cols 32 d + 16 ((lane>>4)&1) + 4 (lane&3)
            const int row_a = sub * 32 + 16 * t + 4 * g + rr;
#ifndef FA_DKV_BKPF256
#define FA_DKV_BKPF256 1
#endif
#ifndef FA_DKV_NO_PREFETCH
            if constexpr (D <= 128 || FA_DKV_BKPF256) {
            u32x4 af[ADB], bfr[ADB];
#pragma unroll
            for (int d = 0; d < ADB; ++d) {
                const u32x2 a0 = lds_read_tr16(dos + swzt_row_off<D>(row_a, d * 64 + cb));
                const u32x2 a1 = lds_read_tr16(dos + swzt_row_off<D>(row_a + 8, d * 64 + cb));
                af[d] = u32x4{a0[0], a0[1], a1[0], a1[1]};
            }
#pragma unroll
            for (int d = 0; d < ADB; ++d) {
                const u32x2 b0 = lds_read_tr16(qs + swzt_row_off<D>(row_a, d * 64 + cb));
                const u32x2 b1 = lds_read_tr16(qs + swzt_row_off<D>(row_a + 8, d * 64 + cb));
                bfr[d] = u32x4{b0[0], b0[1], b1[0], b1[1]};
                dv_acc[d] = E::mfma(af[d], pf[t], dv_acc[d]);
            }
#pragma unroll
            for (int d = 0; d < ADB; ++d) dk_acc[d] = E::mfma(bfr[d], dsf[t], dk_acc[d]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * ADB, 0);
#pragma unroll
            for (int d = 0; d < ADB; ++d) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, ADB, 0);
            } else
#endif
            {
#pragma unroll
            for (int d = 0; d < ADB; ++d) {
                const u32x2 a0 = lds_read_tr16(dos + swzt_row_off<D>(row_a, d * 64 + cb));
                const u32x2 a1 = lds_read_tr16(dos + swzt_row_off<D>(row_a + 8, d * 64 + cb));
                u32x4 af = {a0[0], a0[1], a1[0], a1[1]};
                dv_acc[d] = E::mfma(af, pf[t], dv_acc[d]);
                const u32x2 b0 = lds_read_tr16(qs + swzt_row_off<D>(row_a, d * 64 + cb));
                const u32x2 b1 = lds_read_tr16(qs + swzt_row_off<D>(row_a + 8, d * 64 + cb));
                u32x4 bfr = {b0[0], b0[1], b1[0], b1[1]};
                dk_acc[d] = E::mfma(bfr, dsf[t], dk_acc[d]);
            }
            }
        }
    };

    // dS tiles produced in this step, stored after the step's LDS writes (see compute)
    u32x4 ds_chunk[BQ / 32][2];
    char* ds_tile[BQ / 32];
    uint32_t ds_pending = 0;
    auto flush_ds = [&]() {
#pragma unroll
        for (int sub = 0; sub < BQ / 32; ++sub)
            if (ds_pending & (1u << sub)) {
                *reinterpret_cast<u32x4*>(ds_tile[sub]) = ds_chunk[sub][0];
                *reinterpret_cast<u32x4*>(ds_tile[sub] + 1024) = ds_chunk[sub][1];
            }
        ds_pending = 0;
    };
    auto compute = [&](auto stage_c, int it) {
        constexpr int stage = decltype(stage_c)::value;
        constexpr int NSUB = BQ / 32;
        const int gq = group == 1 ? 0 : it / n_tiles;     // (a scalar division costs ~150 cycles per step)
        const int m0 = (mt0 + it - gq * n_tiles) * BQ;
        if (BIAS && p.alibi_slopes) slope = p.alibi_slopes[b * p.alibi_batch_stride + hk * group + gq];
        const char* qs = smem + stage * STAGE;
        const char* dos = qs + TILE;
        const float* st = reinterpret_cast<const float*>(dos + TILE);
        // any visible (query, key) pair for this wave in rows [q0, q0+31]?
        auto is_active = [&](int q0) { return wave_has_keys && (q0 <= w_qhi_max) && (q0 + 31 >= w_qlo_min); };
        auto needs_mask = [&](int q0) { return key_tail || (q0 < w_qlo_max) || (q0 + 31 > w_qhi_min); };
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int q0 = m0 + sub * 32;
            if (!is_active(q0)) continue;
            f32x16 s_acc, dp_acc;
            u32x4 pf[2], dsf[2];
            f32x4 lse2[4], dsum[4];
            unsigned long long t0 = TMR_NOW();
            (void)t0;
            if (BIAS == 2)      // bias = slope ((kw0 - off - q0) + key_pos - row_pos)
                alibi_b = alibi_lane_operand<T>(lane, slope / p.softmax_scale, -1.f, (float)l31, (float)(kw0 - off - q0));
            sd(qs, dos, sub, s_acc, dp_acc);
#ifdef FA_TIMERS
            asm volatile("s_nop 0" :: "v"(s_acc[0]), "v"(dp_acc[0]));
#endif
            TMR_ADD(0, t0);
#ifndef FA_DKV_SMROWS256
#define FA_DKV_SMROWS256 0      // the register-lean row-group form (sched_barrier between groups): 1223 vs 641 cycles per step
#endif
            if constexpr (D <= 128 || !FA_DKV_SMROWS256) {
                sm_stats(st, sub, lse2, dsum);
                sm(q0, needs_mask(q0), lse2, dsum, s_acc, dp_acc, pf, dsf);
            } else {
                // D = 256 is register-starved: one row group's statistics at a time
                const bool nm = needs_mask(q0);
                if (DROPOUT) kbits_rows = drop_bits(q0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 l4[4], d4[4];
                    l4[i] = *reinterpret_cast<const f32x4*>(st + sub * 32 + 8 * i + 4 * g);
                    d4[i] = *reinterpret_cast<const f32x4*>(st + BQ + sub * 32 + 8 * i + 4 * g);
                    sm_rows(i, q0, nm, l4[i], d4[i], s_acc, dp_acc, pf, dsf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#ifdef FA_TIMERS
            asm volatile("s_nop 0" :: "v"(pf[1][3]), "v"(dsf[1][3]), "v"(pf[0][0]), "v"(dsf[0][0]));
#endif
            TMR_ADD(1, t0);
            if (a.ds_ws && dh_off == 0) {
                // hand dS to the dQ kernel: after one half-exchange per register pair a lane holds 8
                // consecutive query rows of its key (16 bytes) -> two coalesced 1-KiB stores per sub-tile.
                // Tile layout: [t = 16-row group][key][row-octet g][8 rows].  The stores are issued at
                // the END of the step (flush_ds): vmcnt counts stores too, and stores issued before the
                // wait on the next tile's loads would put their latency on the critical path.
                const int h = hk * group + gq;
                ds_tile[sub] = reinterpret_cast<char*>(a.ds_ws) +
                               ((((int64_t)b * p.nheads_q + h) * a.ds_nqb + (q0 >> 5)) * a.ds_nkb + (kw0 >> 5)) * 2048 +
                               l31 * 32 + g * 16;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    uint32_t x0 = dsf[t][0], x1 = dsf[t][1], y0 = dsf[t][2], y1 = dsf[t][3];
                    permlane32_swap(x0, y0);
                    permlane32_swap(x1, y1);
                    ds_chunk[sub][t] = u32x4{x0, x1, y0, y1};
                }
                ds_pending |= 1u << sub;
            }
            bk(qs + dh_off, dos + dh_off, sub, pf, dsf);
#ifdef FA_TIMERS
            asm volatile("s_nop 0" :: "v"(dk_acc[ADB - 1][0]), "v"(dv_acc[ADB - 1][0]));
#endif
            TMR_ADD(2, t0);
        }
    };
    auto step = [&](auto stage_c, int it) {
        constexpr int stage = decltype(stage_c)::value;
        const bool has_next = it + 1 < n_iter;
        unsigned long long t0 = TMR_NOW();
        (void)t0;
        if (has_next) load_tile(it + 1, std::integral_constant<int, stage ^ 1>{});
        TMR_ADD(3, t0);
        compute(stage_c, it);
        t0 = TMR_NOW();
        if (has_next) store_tile(std::integral_constant<int, stage ^ 1>{});
        flush_ds();
        TMR_ADD(4, t0);
        __syncthreads();
        TMR_ADD(5, t0);
        tmr[7] += 1;
    };

#pragma unroll 1
    for (int dh = 0; dh < NDH; ++dh) {
    dh_off = dh * ADB * 64;
#pragma unroll
    for (int d = 0; d < ADB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk_acc[d][r] = 0.f; dv_acc[d][r] = 0.f; }
    if (n_iter > 0) { load_tile(0, std::integral_constant<int, 0>{}); store_tile(std::integral_constant<int, 0>{}); }
    __syncthreads();
    for (int it = 0; it < n_iter; it += 2) {
        step(std::integral_constant<int, 0>{}, it);
        if (it + 1 < n_iter) step(std::integral_constant<int, 1>{}, it + 1);
    }

#ifdef FA_TIMERS
    if (blockIdx.x == FA_TIMERS && tid == 0) {
        for (int i = 0; i < 8; ++i) g_timers[pass * 8 + i] = tmr[i];
        for (int i = 0; i < 8; ++i) tmr[i] = 0;
    }
#endif
    // ---- epilogue: lane (key = l31, g) holds dX[my_key][32 d + 8 rq + 4 g + (0..3)] ----
    if (my_key < sg.seqlen_k) {
        const int64_t dkb = p.cu_seqlens_k ? 0 : (int64_t)b * p.dk_batch_stride;
        const int64_t dvb = p.cu_seqlens_k ? 0 : (int64_t)b * p.dv_batch_stride;
        uint16_t* dkp = reinterpret_cast<uint16_t*>(p.dk) + dkb + (sg.k_row0 + my_key) * p.dk_row_stride + (int64_t)hk * p.dk_head_stride;
        uint16_t* dvp = reinterpret_cast<uint16_t*>(p.dv) + dvb + (sg.k_row0 + my_key) * p.dv_row_stride + (int64_t)hk * p.dv_head_stride;
        const float sc = p.softmax_scale;
#pragma unroll
        for (int d = 0; d < ADB; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u32x2 k2, v2;
                k2[0] = E::pack2(dk_acc[d][4 * rq + 0] * sc, dk_acc[d][4 * rq + 1] * sc);
                k2[1] = E::pack2(dk_acc[d][4 * rq + 2] * sc, dk_acc[d][4 * rq + 3] * sc);
                const float rp = DROPOUT ? a.rp_dropout : 1.0f;
                v2[0] = E::pack2(dv_acc[d][4 * rq + 0] * rp, dv_acc[d][4 * rq + 1] * rp);
                v2[1] = E::pack2(dv_acc[d][4 * rq + 2] * rp, dv_acc[d][4 * rq + 3] * rp);
                if ((dh * ADB + d) * 32 + 8 * rq + 4 * g < dv) {
                    *reinterpret_cast<u32x2*>(dkp + (dh * ADB + d) * 32 + 8 * rq + 4 * g) = k2;
                    *reinterpret_cast<u32x2*>(dvp + (dh * ADB + d) * 32 + 8 * rq + 4 * g) = v2;
                }
            }
    }
    }   // dh (column sweep)
    }   // pass
}

// ---------------------------------------------------------------------------------------------
// 2b. dK / dV, two workgroups per CU
// ---------------------------------------------------------------------------------------------
// The one-wave-per-SIMD kernel above cannot overlap anything (knock-out: loads 0.39 + S/dP 0.31 + VALU
// 0.26 + dV/dK 0.32 + rest 0.26 ms add up to its 1.54 ms).  This variant fits 256 registers (D = 64: 168) and
// 80 KiB of LDS so that TWO (D = 64: THREE) independent workgroups share a CU and one computes while another
// loads or does its exp2 / dS arithmetic (the mechanism that carries the forward and dQ kernels):
//   * the 128 keys' K tile lives in LDS (its B fragments are re-read per use), the V fragments in registers;
//   * Q / dO arrive in 32-row double-buffered stages by LDS-DMA - no staging registers; the DMA of stage
//     it + 1 is issued right behind the barrier that opens stage it (one barrier per stage);
//   * the 64 row statistics of a stage (lse log2 e, -D) are fetched by wave 0 a stage ahead and published in
//     LDS behind the stage; every wave reads them as f32x4 in the accumulator layout, the -D quads straight
//     into the dP accumulator (dP - D costs no instruction).
// Round 5: the loop's instruction stream was put on a diet - at D = 64 (BASELINE config 3) the loop is bound by
// the waves' own VALU / scalar issue, not by the matrix pipe: every LDS read address is a pinned lane constant
// plus an immediate (the two stage buffers are two unrolled copies of the body), the per-head buffer
// descriptors and statistics pointers are rebuilt only when the q-head changes, only wave 0 fetches statistics.
#ifndef FA_DKV2_OCC64
#define FA_DKV2_OCC64 3
#endif
template <int D> struct Dkv2Smem {
    // rows per stage: 64 at D <= 64 (two 32-row sub-tiles per barrier / DMA issue / bookkeeping round - at BASELINE config 3 that
    // skeleton was 1211 of a stage's 2725 cycles, profiles/r05_config3_backward.txt), 32 at D = 128 (LDS: two workgroups per CU)
    static constexpr int BQ = D <= 64 ? 64 : 32;
    static constexpr int KT = DKV_BN * D * 2;            // K tile
    static constexpr int QT = BQ * D * 2;                // Q (or dO) stage
    static constexpr int STG = 2 * QT + 8 * BQ;          // Q, dO, row statistics: lse2[BQ] | -D[BQ]
    static constexpr int TOTAL = KT + 2 * STG;           // K tile + two stages (V fragments in registers)
};

#ifndef FA_DKV2_PF
#define FA_DKV2_PF 3                                    // transposed dO / Q fragments in flight ahead of their MFMA
#endif
// DV: columns that can be non-zero (D = 128 only: head dims 65 .. 96 skip the k-steps and accumulator blocks of the zero columns)
// PART: the launch is a split one (dkv_split_factor): this workgroup walks a share of each pass's query tiles and leaves an fp32
//       partial dK / dV (its own instantiation - the unsplit kernel keeps its registers: one more live value costs a spill at D = 64)
template <typename T, int D, int BIAS, bool DROPOUT, int DV = D, bool PART = false>
__global__ void __launch_bounds__(BWD_THREADS, D <= 64 ? FA_DKV2_OCC64 : 2) fa_bwd_dkdv2_kernel(const KArgs a) {
    using E = Elem<T>;
    static_assert(DV == D || (D == 128 && DV == 96), "narrow form: 96 of 128 columns");
    constexpr int KSTEPS = DV / 16;
    constexpr int DBLKS = DV / 32;
    constexpr int CPR = D / 8;
    constexpr int KT = Dkv2Smem<D>::KT;
    constexpr int QT = Dkv2Smem<D>::QT;
    constexpr int STG = Dkv2Smem<D>::STG;
    constexpr int ROWS_PI = 64 / CPR;                        // rows per 1-KiB DMA instruction
    constexpr int K_INSTS = DKV_BN / ROWS_PI / 4;            // per wave, per tensor
    constexpr int BQ = Dkv2Smem<D>::BQ;
    constexpr int NSUB = BQ / 32;
    constexpr int Q_INSTS = BQ / ROWS_PI / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ks_base = smem;                              // K tile   [128][D]  swzt (row reads)
    char* const stg_base = smem + KT;                        // two stages: Q [32][D] swzt (row + transposed reads), dO, statistics

    const fa_params& p = a.p;
    const int n_kblocks = (p.seqlen_k + DKV_BN - 1) / DKV_BN;
    const bool pair = a.pair_qblocks && n_kblocks >= 2 && !a.flat_kblocks;
    const int n_kb_grid = pair ? (n_kblocks + 1) / 2 : n_kblocks;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    int b, hk, nb0;
    const int nsplit = PART ? a.dkv_split : 1;               // dense launches only (dkv_split_factor)
    int split = 0;
    if (a.flat_kblocks) {
        // varlen flat work list over key blocks (fa_common.h: flat_owner), early (for causal masks: heavy) blocks first
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
    const int group = p.nheads_q / p.nheads_k;
    const int off = sg.off;
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;
    const float c = a.scale_log2e;
    // softcap-only variant (BIAS == 3): cap tanh(s scale / cap) = cap (1 - 2 / (1 + exp2(s k1))), log2 units
    const float cap_k1 = p.softcap > 0.f ? 2.0f * a.scale_log2e / p.softcap : 0.f, cap_c2 = p.softcap * kLog2e;
    (void)cap_k1; (void)cap_c2;
    const int dv = valid_cols(p);
    DropCtx dc = {0, 0, 0, 0};
    if (DROPOUT) {
        dc.k0 = (uint32_t)p.philox_seed; dc.k1 = (uint32_t)(p.philox_seed >> 32);
        dc.thr = a.drop_thr; dc.offset = p.philox_offset;
    }
    const uint64_t drop_n_glob = (uint64_t)p.seqlen_k;

    // DMA geometry: instruction `inst` covers ROWS_PI rows; lane -> (row, physical 16-byte slot); the source
    // offset carries the swizzle (and the out-of-range trick for columns past head_dim_v)
    uint32_t k_voff[K_INSTS], q_voff[Q_INSTS], do_voff[Q_INSTS];
#pragma unroll
    for (int i = 0; i < K_INSTS; ++i) {
        const int row = (wave * K_INSTS + i) * ROWS_PI + lane / CPR;
        const int cbs = swzt_row_off<D>(row, (lane % CPR) * 16) - row * D * 2;
        k_voff[i] = cbs < dv * 2 ? (uint32_t)(row * p.k_row_stride * 2 + cbs) : kOobVoff;
    }
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
    const uint16_t* k_head = reinterpret_cast<const uint16_t*>(p.k) + kb_off + sg.k_row0 * p.k_row_stride + (int64_t)hk * p.k_head_stride;
    const uint16_t* v_head = reinterpret_cast<const uint16_t*>(p.v) + vb_off + sg.k_row0 * p.v_row_stride + (int64_t)hk * p.v_head_stride;
    const __amdgpu_buffer_rsrc_t k_rsrc = make_rsrc(k_head, p.k_row_stride, sg.seqlen_k, dv);
    // lane-constant LDS read addresses, pinned in registers (lds_pin): the stage buffer, the tensor, the 16-row group of
    // a transposed read are immediate offsets of the read instruction.
    // (the K tile uses the same swizzle as the stages, so a key-row read is the stage's row address + a wave-uniform offset)
    const lds_char* q_rp[KSTEPS];                            // row reads of stage 0's Q tile (dO: + QT)
    const lds_char* k_rp[KSTEPS];                            // row reads of the wave's 32 keys in the K tile
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const int a_rd = swzt_row_off<D>(l31, 32 * ks + 16 * g);
        q_rp[ks] = lds_pin(stg_base + a_rd);
        k_rp[ks] = lds_pin(ks_base + wave * 32 * D * 2 + a_rd);
    }
    const int rr = (lane & 15) >> 2;
    const int cb = (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);
    const lds_char* t_rp[2][DBLKS];                          // transposed reads: rows 4 g + rr (+ 8), 64-byte column group d
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int d = 0; d < DBLKS; ++d) t_rp[h2][d] = lds_pin(stg_base + swzt_row_off<D>(4 * g + rr + 8 * h2, d * 64 + cb));
    const lds_char* st_rp = lds_pin(stg_base + 2 * QT + 16 * g);     // statistics: lse2[8 i + 4 g ..], -D 4 BQ bytes further
    const u32x4 alibi_a = alibi_pos_operand<T>(lane);

    const int n_pass = (pair && (n_kblocks - 1 - nb0) != nb0) ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
    const int nb = pass == 0 ? nb0 : n_kblocks - 1 - nb0;
    const int n0 = nb * DKV_BN;
    if (n0 >= sg.seqlen_k) continue;

    const int kw0 = n0 + wave * 32;
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
    // the mask of a sub-tile as two lane constants (see fa_fwd.hip): masked <=> (cpos - lo_t) >u width, lo_t = lo_l - q0
    const bool empty = qhi < qlo;
    const int lo_l = empty ? 0x3fffffff : qlo - 4 * g;
    const uint32_t width = empty ? 0u : (uint32_t)(qhi - qlo);

    // ---- K tile of the 128 keys -> LDS, V fragments of the wave's 32 keys -> registers ----
    __syncthreads();                                         // the previous pass is done with the LDS
    {
        const uint32_t ksoff = (uint32_t)(n0 * p.k_row_stride * 2);
#pragma unroll
        for (int i = 0; i < K_INSTS; ++i) buf_load_lds_b128(k_rsrc, ks_base + (wave * K_INSTS + i) * 1024, k_voff[i], ksoff);
    }
    u32x4 vf[KSTEPS];
    {
        const uint16_t* vr = v_head + (int64_t)my_key * p.v_row_stride + 8 * g;
        const bool ok = my_key < sg.seqlen_k;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            u32x4 z = {0, 0, 0, 0};
            vf[ks] = (ok && 16 * ks + 8 * g < dv) ? *reinterpret_cast<const u32x4*>(vr + 16 * ks) : z;
        }
    }

    f32x16 dk_acc[DBLKS], dv_acc[DBLKS];
#pragma unroll
    for (int d = 0; d < DBLKS; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk_acc[d][r] = 0.f; dv_acc[d][r] = 0.f; }

    // Per-q-head state of the stage stream (rebuilt when the stream wraps to the next head of a GQA group): the Q / dO
    // descriptors and, on the waves that fetch statistics, the lane's statistics row.  32-row stages: wave 0, lanes 0..31 lse and
    // 32..63 softmax_d; 64-row stages: wave 0 lse and wave 1 softmax_d, one row per lane
    __amdgpu_buffer_rsrc_t q_rsrc, do_rsrc;
    const float* stat_row = nullptr;
    const bool stat_wave = BQ == 64 ? wave < 2 : wave == 0;
    const bool stat_is_d = BQ == 64 ? wave == 1 : g == 1;
    const int stat_lrow = BQ == 64 ? lane : l31;
    auto set_head = [&](int gq) {
        const int h = hk * group + gq;
        q_rsrc = make_rsrc(q_base + (int64_t)h * p.q_head_stride, p.q_row_stride, sg.seqlen_q, dv);
        do_rsrc = make_rsrc(do_base + (int64_t)h * p.do_head_stride, p.do_row_stride, sg.seqlen_q, dv);
        if (stat_wave) stat_row = (stat_is_d ? dsum_base : lse_base) + (int64_t)h * p.lse_head_stride;
    };
    const uint32_t q_step = (uint32_t)(BQ * p.q_row_stride * 2), do_step = (uint32_t)(BQ * p.do_row_stride * 2);
    // stage stream being fetched: tile mt_n of head gq_n
    int gq_n = 0, mt_n = mt0;
    uint32_t q_soff = (uint32_t)mt0 * q_step, do_soff = (uint32_t)mt0 * do_step;
    float stat_next = 0.f;
    auto issue_stage = [&](auto par_c) {                      // DMA of stage (gq_n, mt_n) into buffer PAR; wave 0: its statistics
        constexpr int PAR = decltype(par_c)::value;
        char* qd = stg_base + PAR * STG;
#pragma unroll
        for (int i = 0; i < Q_INSTS; ++i) {
            buf_load_lds_b128(q_rsrc, qd + (wave * Q_INSTS + i) * 1024, q_voff[i], q_soff);
            buf_load_lds_b128(do_rsrc, qd + QT + (wave * Q_INSTS + i) * 1024, do_voff[i], do_soff);
        }
        if (stat_wave) {
            const int qi = mt_n * BQ + stat_lrow;
            const int qc = qi < sg.seqlen_q ? qi : (sg.seqlen_q > 0 ? sg.seqlen_q - 1 : 0);
            // the raw value: any arithmetic on it here would put an s_waitcnt vmcnt(0) - the latency of the tile loads issued
            // just above - at the top of the stage; publish_stats() fixes it up where it is consumed, a stage later
            stat_next = stat_row[qc];
        }
    };
    // statistics of the stage that was just fetched -> LDS behind its buffer: lse log2 e (rows past the sequence: 0) and -D
    auto publish_stats = [&](auto par_c, int mt_x) {
        constexpr int PAR = decltype(par_c)::value;
        if (stat_wave) {
            const float x = mt_x * BQ + stat_lrow < sg.seqlen_q ? (stat_is_d ? -stat_next : stat_next * kLog2e) : 0.f;
            reinterpret_cast<float*>(stg_base + PAR * STG + 2 * QT)[(stat_is_d ? BQ : 0) + stat_lrow] = x;
        }
    };
    auto advance_n = [&]() {
        ++mt_n; q_soff += q_step; do_soff += do_step;
        if (mt_n == mt1) {                                    // next q-head of the group, first tile
            mt_n = mt0; ++gq_n;
            q_soff = (uint32_t)mt0 * q_step; do_soff = (uint32_t)mt0 * do_step;
            if (gq_n < group) set_head(gq_n);
        }
    };
    int gq = 0, mt = mt0;                                  // stage it = (q-head gq of the group, 32-row tile mt)
    if (n_iter > 0) {
        set_head(0);
        issue_stage(std::integral_constant<int, 0>{});
        publish_stats(std::integral_constant<int, 0>{}, mt_n);
    }

    // ---- one stage: buffer PAR holds stage it; stage it + 1 is fetched into the other buffer ----
    auto stage = [&](auto par_c, int it) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr int SB = PAR * STG;                          // byte offset of the stage buffer: an immediate of every read
        const int q0_stage = mt * BQ;
        const int h = hk * group + gq;
        (void)h;
        __syncthreads();                                     // stage it landed (vmcnt(0) before the barrier) and
        advance_n();                                         // everyone left stage it-1: its buffer is re-filled
        const bool has_next = it + 1 < n_iter;
        if (has_next) issue_stage(std::integral_constant<int, PAR ^ 1>{});
        const int mt_pub = mt_n;
        if (++mt == mt1) { mt = mt0; ++gq; }

#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {               // the stage's 32-row sub-tiles, one after the other (same registers)
        const int q0 = q0_stage + 32 * sub;
        const int SO = SB + sub * 32 * D * 2;                // (an immediate after unrolling)
        const int ST = SB + sub * 128;
        const bool active = wave_has_keys && (q0 <= w_qhi_max) && (q0 + 31 >= w_qlo_min);
        if (active) {
        // ---- S = Q K^T, dP = dO V^T - D ----
        f32x16 s_acc, dp_acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 d4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(st_rp + (ST + 4 * BQ + 32 * i));
#pragma unroll
            for (int e = 0; e < 4; ++e) dp_acc[4 * i + e] = DROPOUT ? 0.f : d4[e];       // the accumulator starts from -D
        }
        f32x4 dneg[DROPOUT ? 4 : 1];
        if (DROPOUT) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dneg[i] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(st_rp + (ST + 4 * BQ + 32 * i));
        }
        if (BIAS == 2) {
            const float slope = p.alibi_slopes[b * p.alibi_batch_stride + h];
            const u32x4 ab = alibi_lane_operand<T>(lane, slope / p.softmax_scale, -1.f, (float)l31, (float)(kw0 - off - q0));
            s_acc = E::mfma(alibi_a, ab, s_acc);
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const u32x4 qa = lds_read_b128(q_rp[ks] + SO);
            const u32x4 kb2 = lds_read_b128(k_rp[ks]);
            const u32x4 da = lds_read_b128(q_rp[ks] + (SO + QT));
            s_acc = E::mfma(qa, kb2, s_acc);
            dp_acc = E::mfma(da, vf[ks], dp_acc);
        }
        // ---- P, dS ----
        const bool need_mask = key_tail || (q0 < w_qlo_max) || (q0 + 31 > w_qhi_min);
        u32x4 pf[2], dsf[2];
        float pv[16], dsv[16];
        uint32_t kbits = 0xffffu;
        if (DROPOUT) {
            // the quad (4 consecutive keys = one Philox counter per row) draws each row once: see drop_bits
            // in fa_bwd_dkdv_kernel
            uint32_t mine = 0;
            const int lq = lane & 3;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                mine |= dropout_keep4(dc, (uint64_t)(sg.q_row0 + q0 + lq + 8 * i + 4 * g) * drop_n_glob + (uint64_t)(my_key & ~3)) << (4 * i);
            const int kq = my_key & 3;
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x00, 0xf, 0xf, false);
            const uint32_t a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x55, 0xf, 0xf, false);
            const uint32_t a2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xaa, 0xf, 0xf, false);
            const uint32_t a3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xff, 0xf, 0xf, false);
            kbits = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kbits |= ((a0 >> (4 * i + kq)) & 1u) << (4 * i + 0);
                kbits |= ((a1 >> (4 * i + kq)) & 1u) << (4 * i + 1);
                kbits |= ((a2 >> (4 * i + kq)) & 1u) << (4 * i + 2);
                kbits |= ((a3 >> (4 * i + kq)) & 1u) << (4 * i + 3);
            }
        }
        f32x4 l4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) l4[i] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(st_rp + (ST + 32 * i));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float l2 = l4[r >> 2][r & 3];
            float pr, chain = 1.0f;
            if (BIAS == 3) {
                const float rr1 = fast_rcp(1.0f + fast_exp2(s_acc[r] * cap_k1));
                const float t = fmaf(rr1, -2.0f, 1.0f);
                pr = fast_exp2(fmaf(rr1, -2.0f * cap_c2, cap_c2) - l2);
                chain = fmaf(-t, t, 1.0f);
            } else {
                pr = fast_exp2(fmaf(s_acc[r], c, -l2));
            }
            if (DROPOUT) {       // dS = P (keep rp dP - D); dV accumulates keep P (rp applied in the epilogue)
                const bool keep = (kbits >> r) & 1u;
                pv[r] = keep ? pr : 0.f;
                dsv[r] = pr * ((keep ? dp_acc[r] * a.rp_dropout : 0.f) + dneg[DROPOUT ? (r >> 2) : 0][r & 3]);
            } else if (BIAS == 3) {
                pv[r] = pr;
                dsv[r] = pr * dp_acc[r] * chain;
            } else {
                pv[r] = pr;
                dsv[r] = pr * dp_acc[r];
            }
        }
        if (need_mask) {                                     // wave-uniform branch: interior tiles skip all of it
            const int lo_t = lo_l - q0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cpos = (r & 3) + 8 * (r >> 2);
                if ((uint32_t)(cpos - lo_t) > width) { pv[r] = 0.f; dsv[r] = 0.f; }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                pf[t][w2] = E::pack2(pv[8 * t + 2 * w2], pv[8 * t + 2 * w2 + 1]);
                dsf[t][w2] = E::pack2(dsv[8 * t + 2 * w2], dsv[8 * t + 2 * w2 + 1]);
            }
#ifdef FA_MEASURE
        if (a.ds_ws) {
            // dS hand-off to the one-GEMM dQ kernel (see fa_bwd_dkdv_kernel): two coalesced 1-KiB stores
            char* tile = reinterpret_cast<char*>(a.ds_ws) +
                         ((((int64_t)b * p.nheads_q + h) * a.ds_nqb + (q0 >> 5)) * a.ds_nkb + (kw0 >> 5)) * 2048 + l31 * 32 + g * 16;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint32_t x0 = dsf[t][0], x1 = dsf[t][1], y0 = dsf[t][2], y1 = dsf[t][3];
                permlane32_swap(x0, y0);
                permlane32_swap(x1, y1);
                *reinterpret_cast<u32x4*>(tile + t * 1024) = u32x4{x0, x1, y0, y1};
            }
        }
#endif
        // ---- dV^T += dO^T P,  dK^T += Q^T dS ----
        {
            // MFMA i = (t, d, dO | Q); fences keep the transposed operand of MFMA i + PF in flight ahead of MFMA i
            // (S / dP are dead here, so the extra fragments cost no registers; hipcc alone serialises
            // read -> wait -> MFMA through one temporary)
            constexpr int NBK = 4 * DBLKS;
            auto tread = [&](int i) {
                const int t = i / (2 * DBLKS), d = (i >> 1) % DBLKS;
                const int o = SO + ((i & 1) ? 0 : QT) + 16 * t * D * 2;
                const u32x2 a0 = lds_read_tr16_nw(t_rp[0][d], o);
                const u32x2 a1 = lds_read_tr16_nw(t_rp[1][d], o);
                return u32x4{a0[0], a0[1], a1[0], a1[1]};
            };
            u32x4 tf[NBK];
#pragma unroll
            for (int i = 0; i < FA_DKV2_PF && i < NBK; ++i) tf[i] = tread(i);
#pragma unroll
            for (int i = 0; i < NBK; ++i) {
                if (i + FA_DKV2_PF < NBK) tf[i + FA_DKV2_PF] = tread(i + FA_DKV2_PF);
                lds_tr_wait(tf[i], 2 * ((NBK - 1 - i) < FA_DKV2_PF ? (NBK - 1 - i) : FA_DKV2_PF));
                __builtin_amdgcn_sched_barrier(0);
                const int t = i / (2 * DBLKS), d = (i >> 1) % DBLKS;
                if (i & 1) dk_acc[d] = E::mfma(tf[i], dsf[t], dk_acc[d]);
                else dv_acc[d] = E::mfma(tf[i], pf[t], dv_acc[d]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }   // active
        }   // sub
        if (has_next) publish_stats(std::integral_constant<int, PAR ^ 1>{}, mt_pub);
    };
#pragma unroll 1
    for (int it = 0; it < n_iter; it += 2) {
        stage(std::integral_constant<int, 0>{}, it);
        if (it + 1 < n_iter) stage(std::integral_constant<int, 1>{}, it + 1);
    }

    if (my_key < sg.seqlen_k) {
        // (the output addresses are formed HERE: left visible, hipcc computes the two 64-bit lane pointers in front of the stage loop
        //  and, at three workgroups per CU - 168 registers -, parks them in scratch across it; the lane's key index passes through an
        //  opaque asm so that nothing below can be hoisted)
        int my_key_e = my_key, g_e = g;
        asm volatile("" : "+v"(my_key_e), "+v"(g_e));
        const int64_t dkb = p.cu_seqlens_k ? 0 : (int64_t)b * p.dk_batch_stride;
        const int64_t dvb = p.cu_seqlens_k ? 0 : (int64_t)b * p.dv_batch_stride;
        uint16_t* dkp = reinterpret_cast<uint16_t*>(p.dk) + dkb + (sg.k_row0 + my_key_e) * p.dk_row_stride + (int64_t)hk * p.dk_head_stride;
        uint16_t* dvp = reinterpret_cast<uint16_t*>(p.dv) + dvb + (sg.k_row0 + my_key_e) * p.dv_row_stride + (int64_t)hk * p.dv_head_stride;
        if (PART) {
            // partial dK / dV of this split, fp32 [dK | dV][split][B][Sk][Hk][D]: the accumulators as they are (dK scaled, dV
            // with the dropout factor) - dkv_reduce_kernel adds the splits and rounds once
            const int64_t row = (int64_t)p.nheads_k * D;
            const int64_t slab = (p.cu_seqlens_k ? (int64_t)p.total_k : (int64_t)p.batch * p.seqlen_k) * row;
            const int64_t krow = p.cu_seqlens_k ? sg.k_row0 + my_key : (int64_t)b * p.seqlen_k + my_key;
            float* pk = reinterpret_cast<float*>(a.dkv_part) + split * slab + krow * row + (int64_t)hk * D;
            float* pv = pk + nsplit * slab;
            const float sc = p.softmax_scale, rp = DROPOUT ? a.rp_dropout : 1.0f;
#pragma unroll
            for (int d = 0; d < DBLKS; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    if (d * 32 + 8 * rq + 4 * g_e < dv) {
                        const f32x4 k4 = {dk_acc[d][4 * rq + 0] * sc, dk_acc[d][4 * rq + 1] * sc, dk_acc[d][4 * rq + 2] * sc, dk_acc[d][4 * rq + 3] * sc};
                        const f32x4 v4 = {dv_acc[d][4 * rq + 0] * rp, dv_acc[d][4 * rq + 1] * rp, dv_acc[d][4 * rq + 2] * rp, dv_acc[d][4 * rq + 3] * rp};
                        *reinterpret_cast<f32x4*>(pk + d * 32 + 8 * rq + 4 * g_e) = k4;
                        *reinterpret_cast<f32x4*>(pv + d * 32 + 8 * rq + 4 * g_e) = v4;
                    }
                }
        } else {
        const float sc = p.softmax_scale;
#pragma unroll
        for (int d = 0; d < DBLKS; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u32x2 k2, v2;
                k2[0] = E::pack2(dk_acc[d][4 * rq + 0] * sc, dk_acc[d][4 * rq + 1] * sc);
                k2[1] = E::pack2(dk_acc[d][4 * rq + 2] * sc, dk_acc[d][4 * rq + 3] * sc);
                const float rp = DROPOUT ? a.rp_dropout : 1.0f;
                v2[0] = E::pack2(dv_acc[d][4 * rq + 0] * rp, dv_acc[d][4 * rq + 1] * rp);
                v2[1] = E::pack2(dv_acc[d][4 * rq + 2] * rp, dv_acc[d][4 * rq + 3] * rp);
                if (d * 32 + 8 * rq + 4 * g_e < dv) {
                    *reinterpret_cast<u32x2*>(dkp + d * 32 + 8 * rq + 4 * g_e) = k2;
                    *reinterpret_cast<u32x2*>(dvp + d * 32 + 8 * rq + 4 * g_e) = v2;
                }
            }
        }
    }
    }   // pass
}

// ---------------------------------------------------------------------------------------------
// 3. dQ
// ---------------------------------------------------------------------------------------------
#ifndef FA_DQ_OCC64
#define FA_DQ_OCC64 2                                   // waves per SIMD of the dQ kernel at head dim 64
#endif
#ifndef FA_DQ_PFS
#define FA_DQ_PFS 2                                     // K / V fragments in flight ahead of their S / dP MFMA
#endif
#ifndef FA_DQ_PFT
#define FA_DQ_PFT 2                                     // transposed K fragments in flight ahead of their dQ MFMA
#endif
constexpr int DQ_BM = 128;
constexpr int DQ_BN = 64;

template <int D> struct DqSmem {
    static constexpr int TILE = DQ_BN * D * 2;
    static constexpr int STAGE = 2 * TILE;
    static constexpr int TOTAL = 2 * STAGE;
};

// DV: columns that can be non-zero (D = 256 only: head dims 129 .. 192 skip the k-steps and accumulator blocks of the zero columns)
template <typename T, int D, int BIAS, int OCC, bool DROPOUT, int DV = D>
__global__ void __launch_bounds__(BWD_THREADS, OCC) fa_bwd_dq_kernel(const KArgs a) {
    using E = Elem<T>;
    static_assert(DV == D || (D == 256 && DV == 192) || (D == 128 && DV == 96), "narrow forms: 192 of 256, 96 of 128 columns");
    constexpr int KSTEPS = DV / 16;
    constexpr int DBLKS = DV / 32;
    constexpr int CPR = D / 8;
    constexpr int CHUNKS = DQ_BN * CPR / BWD_THREADS;
    constexpr int TILE = DqSmem<D>::TILE;
    constexpr int STAGE = DqSmem<D>::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const fa_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const WorkItem w = a.flat_blocks
        ? decode_work_flat(blockIdx.x, a.flat_blocks, DQ_BM, p.batch, p.nheads_q, p.nheads_k, p.cu_seqlens_q, lane)
        : decode_work(blockIdx.x, p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
    if (!w.valid) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SeqGeom sg = seq_geom(p, w.b);
    const int off = sg.off;
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;
    const float c = a.scale_log2e;
    // softcap-only variant (BIAS == 3): cap tanh(s scale / cap) = cap (1 - 2 / (1 + exp2(s k1))), log2 units
    const float cap_k1 = p.softcap > 0.f ? 2.0f * a.scale_log2e / p.softcap : 0.f, cap_c2 = p.softcap * kLog2e;
    (void)cap_k1; (void)cap_c2;

    const int64_t kb_off = p.cu_seqlens_k ? 0 : (int64_t)w.b * p.k_batch_stride;
    const int64_t vb_off = p.cu_seqlens_k ? 0 : (int64_t)w.b * p.v_batch_stride;
    const uint16_t* kp = reinterpret_cast<const uint16_t*>(p.k) + kb_off + sg.k_row0 * p.k_row_stride + (int64_t)w.hk * p.k_head_stride;
    const uint16_t* vp = reinterpret_cast<const uint16_t*>(p.v) + vb_off + sg.k_row0 * p.v_row_stride + (int64_t)w.hk * p.v_head_stride;
    const int dv = valid_cols(p);
    const __amdgpu_buffer_rsrc_t k_rsrc = make_rsrc(kp, p.k_row_stride, sg.seqlen_k, dv);
    const __amdgpu_buffer_rsrc_t v_rsrc = make_rsrc(vp, p.v_row_stride, sg.seqlen_k, dv);
    const uint32_t k_tile_bytes = (uint32_t)(DQ_BN * p.k_row_stride * 2);
    const uint32_t v_tile_bytes = (uint32_t)(DQ_BN * p.v_row_stride * 2);
    // LDS-DMA staging (see fa_fwd.hip): instruction `inst` = wave*CHUNKS + i covers ROWS_PI rows,
    // lane -> (row, physical slot); the source offset carries the swizzle.
    constexpr int ROWS_PI = 64 / CPR;
    uint32_t k_voff[CHUNKS], v_voff[CHUNKS];
    int k_lds[CHUNKS], v_lds[CHUNKS];
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int inst = wave * CHUNKS + i;
        const int row = inst * ROWS_PI + lane / CPR;
        const int slot = lane % CPR;
        const int k_cb = swzt_row_off<D>(row, slot * 16) - row * D * 2;
        const int v_cb = swz_row_off<D>(row, slot * 16) - row * D * 2;
        k_voff[i] = k_cb < dv * 2 ? (uint32_t)(row * p.k_row_stride * 2 + k_cb) : kOobVoff;
        v_voff[i] = v_cb < dv * 2 ? (uint32_t)(row * p.v_row_stride * 2 + v_cb) : kOobVoff;
        k_lds[i] = inst * 1024;
        v_lds[i] = TILE + inst * 1024;
    }
    int k_rd[KSTEPS], v_rd[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        k_rd[ks] = swzt_row_off<D>(l31, 32 * ks + 16 * g);
        v_rd[ks] = TILE + swz_row_off<D>(l31, 32 * ks + 16 * g);
    }
    const int rr = (lane & 15) >> 2;
    const int cb = (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);
#if FA_DQ_PFS > 0
    // pinned lane-constant read addresses (stage and key block are immediate offsets): see lds_pin
    const lds_char* k_ptr[KSTEPS];
    const lds_char* v_ptr[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) { k_ptr[ks] = lds_pin(smem + k_rd[ks]); v_ptr[ks] = lds_pin(smem + v_rd[ks]); }
#endif
#if FA_DQ_PFT > 0
    const lds_char* t_ptr[2][DBLKS];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int d = 0; d < DBLKS; ++d) t_ptr[h][d] = lds_pin(smem + swzt_row_off<D>(4 * g + rr + 8 * h, d * 64 + cb));
#endif
    float slope = 0.f;
    if (BIAS && p.alibi_slopes) slope = p.alibi_slopes[w.b * p.alibi_batch_stride + w.h];
    const u32x4 alibi_a = alibi_pos_operand<T>(lane);
    DropCtx dc = {0, 0, 0, 0};
    if (DROPOUT) {
        dc.k0 = (uint32_t)p.philox_seed; dc.k1 = (uint32_t)(p.philox_seed >> 32);
        dc.thr = a.drop_thr; dc.offset = p.philox_offset;
    }
    const uint64_t drop_n_glob = (uint64_t)p.seqlen_k;

    const int n_pass = (a.pair_qblocks && (a.n_qblocks_total - 1 - w.qb) != w.qb) ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
    const int qb_cur = pass == 0 ? w.qb : a.n_qblocks_total - 1 - w.qb;
    const int m_block = qb_cur * DQ_BM;
    if (m_block >= sg.seqlen_q) continue;
    int n_min = 0, n_max = (sg.seqlen_k + DQ_BN - 1) / DQ_BN;
    {
        const int m_last = (m_block + DQ_BM < sg.seqlen_q ? m_block + DQ_BM : sg.seqlen_q) - 1;
        if (wr >= 0) {
            const int kmax = m_last + off + wr;
            const int t = kmax < 0 ? 0 : kmax / DQ_BN + 1;
            n_max = t < n_max ? t : n_max;
        }
        if (wl >= 0) { const int kmin = m_block + off - wl; if (kmin > 0) n_min = kmin / DQ_BN; }
    }
    const int wave_row0 = m_block + wave * 32;
    const int my_row = wave_row0 + l31;
    int lo = 0, hi = sg.seqlen_k - 1;
    if (wr >= 0) { const int h2 = my_row + off + wr; hi = h2 < hi ? h2 : hi; }
    if (wl >= 0) { const int l2 = my_row + off - wl; lo = l2 > lo ? l2 : lo; }
    if (my_row >= sg.seqlen_q) { lo = 0x7fffffff; hi = -1; }
    const int wrow_last = wave_row0 + 31;
    int w_hi_min = sg.seqlen_k - 1, w_hi_max = sg.seqlen_k - 1, w_lo_max = 0;
    if (wr >= 0) {
        const int h0 = wave_row0 + off + wr, h1 = wrow_last + off + wr;
        w_hi_min = h0 < w_hi_min ? h0 : w_hi_min; w_hi_max = h1 < w_hi_max ? h1 : w_hi_max;
    }
    if (wl >= 0) { const int l1 = wrow_last + off - wl; w_lo_max = l1 > 0 ? l1 : 0; }
    const int w_lo_min = (wl >= 0 && wave_row0 + off - wl > 0) ? wave_row0 + off - wl : 0;
    const bool row_tail = wrow_last >= sg.seqlen_q;

    // ---- Q, dO fragments (B operands) + row statistics ----
    u32x4 qf[KSTEPS], dof[KSTEPS];
    float lse2 = 0.f, dsum = 0.f;
    {
        const bool ok = my_row < sg.seqlen_q;
        const int64_t qb_off = p.cu_seqlens_q ? 0 : (int64_t)w.b * p.q_batch_stride;
        const int64_t dob_off = p.cu_seqlens_q ? 0 : (int64_t)w.b * p.do_batch_stride;
        const uint16_t* qrow = reinterpret_cast<const uint16_t*>(p.q) + qb_off + (sg.q_row0 + my_row) * p.q_row_stride +
                               (int64_t)w.h * p.q_head_stride + 8 * g;
        const uint16_t* dorow = reinterpret_cast<const uint16_t*>(p.dout) + dob_off + (sg.q_row0 + my_row) * p.do_row_stride +
                                (int64_t)w.h * p.do_head_stride + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            u32x4 z = {0, 0, 0, 0};
            const bool okc = ok && 16 * ks + 8 * g < dv;
            qf[ks] = okc ? *reinterpret_cast<const u32x4*>(qrow + 16 * ks) : z;
            dof[ks] = okc ? *reinterpret_cast<const u32x4*>(dorow + 16 * ks) : z;
        }
        const int64_t so = (int64_t)w.b * p.lse_batch_stride + (int64_t)w.h * p.lse_head_stride + sg.q_row0 + my_row;
        if (a.fuse_pre) {
            // D = rowsum(dO o O) here instead of in a preprocess launch (dense asm path: this kernel runs FIRST and
            // leaves softmax_d and the dK/dV kernel's statistics behind): the lane holds half of its row's dO already
            const int64_t ob_off = p.cu_seqlens_q ? 0 : (int64_t)w.b * p.o_batch_stride;
            const uint16_t* orow = reinterpret_cast<const uint16_t*>(p.o) + ob_off + (sg.q_row0 + my_row) * p.o_row_stride +
                                   (int64_t)w.h * p.o_head_stride + 8 * g;
            float acc = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                u32x4 ov = {0, 0, 0, 0};
                if (ok && 16 * ks + 8 * g < dv) ov = *reinterpret_cast<const u32x4*>(orow + 16 * ks);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc = fmaf(E::lo(ov[j]), E::lo(dof[ks][j]), acc);
                    acc = fmaf(E::hi(ov[j]), E::hi(dof[ks][j]), acc);
                }
            }
            acc += __shfl_xor(acc, 32, 64);
            dsum = acc;
            if (ok) {
                const float lse = p.lse[so];
                lse2 = lse * kLog2e;
                if (g == 0) {
                    p.softmax_d[so] = acc;
                    // dense: [B][H][Sq] planes; packed sequences: [H][total_q]
                    const int64_t plane = p.cu_seqlens_q ? (int64_t)p.nheads_q * p.total_q : (int64_t)p.batch * p.nheads_q * p.seqlen_q;
                    const int64_t at = p.cu_seqlens_q ? (int64_t)w.h * p.total_q + sg.q_row0 + my_row
                                                      : ((int64_t)w.b * p.nheads_q + w.h) * p.seqlen_q + my_row;
                    if (a.stats_ws) {                         // the asm dK/dV kernel's statistics planes
                        a.stats_ws[at] = lse == -INFINITY ? INFINITY : lse2;
                        a.stats_ws[plane + at] = -acc;
                    }
                }
            }
        } else if (ok) {
            lse2 = p.lse[so] * kLog2e;
            dsum = p.softmax_d[so];
        }
    }

    auto load_tile = [&](int nb, auto stage_c) {
        constexpr int stage = decltype(stage_c)::value;
        char* base = smem + stage * STAGE;
        const uint32_t ks_off = (uint32_t)nb * k_tile_bytes;
        const uint32_t vs_off = (uint32_t)nb * v_tile_bytes;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(k_rsrc, base + k_lds[i], k_voff[i], ks_off);
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(v_rsrc, base + v_lds[i], v_voff[i], vs_off);
    };

    f32x16 dq_acc[DBLKS];
#pragma unroll
    for (int d = 0; d < DBLKS; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq_acc[d][r] = 0.f;

    auto compute = [&](auto stage_c, int nb) {
        constexpr int stage = decltype(stage_c)::value;
        const int n0 = nb * DQ_BN;
        const char* sbase = smem + stage * STAGE;
        const bool need_mask = row_tail || (n0 + DQ_BN - 1 > w_hi_min) || (n0 < w_lo_max);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            // S^T, dP^T : acc[r] = X[my_row][n0 + 32 kb + row(r,g)]
            f32x16 s_acc, dp_acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s_acc[r] = 0.f; dp_acc[r] = 0.f; }
            if (BIAS == 2) {    // A side = key rows: position of the register; bias = slope ((n0 + 32 kb - off - row0) + key_pos - row_pos)
                const u32x4 ab = alibi_lane_operand<T>(lane, slope / p.softmax_scale, 1.f, -(float)l31,
                                                       (float)(n0 + kb * 32 - off - wave_row0));
                s_acc = E::mfma(alibi_a, ab, s_acc);
            }
#if FA_DQ_PFS > 0
            {
                // two alternating accumulator chains; fences keep the K / V fragment of MFMA i + PFS in flight ahead of
                // MFMA i (hipcc alone serialises read -> wait -> MFMA through one temporary)
                constexpr int NSD = 2 * KSTEPS;
                auto fread = [&](int i) { return lds_read_b128(((i & 1) ? v_ptr[i >> 1] : k_ptr[i >> 1]) + (stage * STAGE + kb * 32 * D * 2)); };
                u32x4 fr[NSD];
#pragma unroll
                for (int i = 0; i < FA_DQ_PFS && i < NSD; ++i) fr[i] = fread(i);
#pragma unroll
                for (int i = 0; i < NSD; ++i) {
                    if (i + FA_DQ_PFS < NSD) fr[i + FA_DQ_PFS] = fread(i + FA_DQ_PFS);
                    __builtin_amdgcn_sched_barrier(0);
                    if (i & 1) dp_acc = E::mfma(fr[i], dof[i >> 1], dp_acc);
                    else s_acc = E::mfma(fr[i], qf[i >> 1], s_acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#else
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {                // two alternating accumulator chains
                const u32x4 ka = lds_read_b128(sbase + k_rd[ks] + kb * 32 * D * 2);
                const u32x4 va = lds_read_b128(sbase + v_rd[ks] + kb * 32 * D * 2);
                s_acc = E::mfma(ka, qf[ks], s_acc);
                dp_acc = E::mfma(va, dof[ks], dp_acc);
            }
#endif
            float dsv[16];
            uint32_t kbits = 0xffffu;
            if (DROPOUT) {
                kbits = 0;
                const uint64_t row_g = (uint64_t)(sg.q_row0 + my_row);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int j4 = n0 + kb * 32 + 8 * rg + 4 * g;
                    kbits |= dropout_keep4(dc, row_g * drop_n_glob + (uint64_t)j4) << (4 * rg);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float dpe = dp_acc[r];
                if (DROPOUT) dpe = ((kbits >> r) & 1u) ? dpe * a.rp_dropout : 0.f;   // dS = P (keep rp dP - D)
                float pr, dsr;
                if (BIAS == 1) {
                    const int j = n0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                    float s = s_acc[r] * p.softmax_scale;
                    s = fmaf(-slope, fabsf((float)(my_row + off - j)), s);
                    float chain = 1.f;
                    if (p.softcap > 0.f) {
                        const float t = fast_tanh(s / p.softcap);
                        s = p.softcap * t;
                        chain = 1.f - t * t;
                    }
                    pr = fast_exp2(fmaf(s, kLog2e, -lse2));
                    dsr = pr * (dpe - dsum) * chain;
                } else if (BIAS == 3) {
                    const float rr1 = fast_rcp(1.0f + fast_exp2(s_acc[r] * cap_k1));
                    const float t = fmaf(rr1, -2.0f, 1.0f);
                    pr = fast_exp2(fmaf(rr1, -2.0f * cap_c2, cap_c2) - lse2);
                    dsr = pr * (dpe - dsum) * fmaf(-t, t, 1.0f);
                } else {
                    pr = fast_exp2(fmaf(s_acc[r], c, -lse2));
                    dsr = pr * (dpe - dsum);
                }
                dsv[r] = dsr;
            }
            if (need_mask) {
                const bool empty = hi < lo;                // folded into the operands (see fa_fwd.hip)
                const int lo_t = empty ? 0x3fffffff : lo - n0 - kb * 32 - 4 * g;
                const uint32_t width = empty ? 0u : (uint32_t)(hi - lo);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cpos = (r & 3) + 8 * (r >> 2);
                    if ((uint32_t)(cpos - lo_t) > width) dsv[r] = 0.f;
                }
            }
            // dQ^T += K^T dS^T
#if FA_DQ_PFT > 0
            {
                constexpr int NDQ = 2 * DBLKS;
                u32x4 dsf[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) dsf[t][w2] = E::pack2(dsv[8 * t + 2 * w2], dsv[8 * t + 2 * w2 + 1]);
                auto tread = [&](int i) {           // (asm form: no vmcnt wait for the DMA of the next tile, see lds_read_tr16_nw)
                    const int t = i / DBLKS, d = i % DBLKS;
                    const u32x2 a0 = lds_read_tr16_nw(t_ptr[0][d], stage * STAGE + (kb * 32 + 16 * t) * D * 2);
                    const u32x2 a1 = lds_read_tr16_nw(t_ptr[1][d], stage * STAGE + (kb * 32 + 16 * t) * D * 2);
                    return u32x4{a0[0], a0[1], a1[0], a1[1]};
                };
                u32x4 tf[NDQ];
#pragma unroll
                for (int i = 0; i < FA_DQ_PFT && i < NDQ; ++i) tf[i] = tread(i);
#pragma unroll
                for (int i = 0; i < NDQ; ++i) {
                    if (i + FA_DQ_PFT < NDQ) tf[i + FA_DQ_PFT] = tread(i + FA_DQ_PFT);
                    lds_tr_wait(tf[i], 2 * ((NDQ - 1 - i) < FA_DQ_PFT ? (NDQ - 1 - i) : FA_DQ_PFT));
                    __builtin_amdgcn_sched_barrier(0);
                    dq_acc[i % DBLKS] = E::mfma(tf[i], dsf[i / DBLKS], dq_acc[i % DBLKS]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#else
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 dsf;
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) dsf[w2] = E::pack2(dsv[8 * t + 2 * w2], dsv[8 * t + 2 * w2 + 1]);
                const int row_a = kb * 32 + 16 * t + 4 * g + rr;
#pragma unroll
                for (int d = 0; d < DBLKS; ++d) {
                    const u32x2 a0 = lds_read_tr16(sbase + swzt_row_off<D>(row_a, d * 64 + cb));
                    const u32x2 a1 = lds_read_tr16(sbase + swzt_row_off<D>(row_a + 8, d * 64 + cb));
                    u32x4 af = {a0[0], a0[1], a1[0], a1[1]};
                    dq_acc[d] = E::mfma(af, dsf, dq_acc[d]);
                }
            }
#endif
        }
    };
    auto step = [&](auto stage_c, int nb) {
        constexpr int stage = decltype(stage_c)::value;
        const bool has_next = nb + 1 < n_max;
        if (has_next) load_tile(nb + 1, std::integral_constant<int, stage ^ 1>{});
        const int n0 = nb * DQ_BN;
        const bool wave_active = (n0 <= w_hi_max) && (n0 + DQ_BN - 1 >= w_lo_min) && (wave_row0 < sg.seqlen_q);
        if (wave_active) compute(stage_c, nb);
        __syncthreads();
    };

    if (n_min < n_max) load_tile(n_min, std::integral_constant<int, 0>{});
    __syncthreads();
    for (int nb = n_min; nb < n_max; nb += 2) {
        step(std::integral_constant<int, 0>{}, nb);
        if (nb + 1 < n_max) step(std::integral_constant<int, 1>{}, nb + 1);
    }

    if (my_row < sg.seqlen_q) {
        const int64_t dqb = p.cu_seqlens_q ? 0 : (int64_t)w.b * p.dq_batch_stride;
        uint16_t* dqp = reinterpret_cast<uint16_t*>(p.dq) + dqb + (sg.q_row0 + my_row) * p.dq_row_stride + (int64_t)w.h * p.dq_head_stride;
        const float sc = p.softmax_scale;
#pragma unroll
        for (int d = 0; d < DBLKS; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u32x2 o2;
                o2[0] = E::pack2(dq_acc[d][4 * rq + 0] * sc, dq_acc[d][4 * rq + 1] * sc);
                o2[1] = E::pack2(dq_acc[d][4 * rq + 2] * sc, dq_acc[d][4 * rq + 3] * sc);
                if (d * 32 + 8 * rq + 4 * g < dv) *reinterpret_cast<u32x2*>(dqp + d * 32 + 8 * rq + 4 * g) = o2;
            }
    }
    }   // pass
}

// ---------------------------------------------------------------------------------------------
// 3b. dQ from the dS tiles written by the dK/dV kernel:  dQ = scale * dS K   (one GEMM, no exp)
// ---------------------------------------------------------------------------------------------
// Workgroup = 128 query rows (32 per wave).  K tiles (64 keys) are shared through LDS (LDS-DMA,
// transposable swizzle); every wave DMA-loads its own two 2-KiB dS tiles per K tile into a private
// LDS area and reads them back with ds_read_b64_tr_b16 as the B operand of dQ^T += K^T dS^T.
// It reads B*Hq*pairs*2 bytes of dS instead of recomputing S and dP (16 of the 24 MFMAs per
// sub-tile of fa_bwd_dq_kernel): HBM-bound where the other one is matrix-pipe bound.
template <int D> struct DqdsSmem {
    static constexpr int KTILE = DQ_BN * D * 2;
    static constexpr int DS = 4 * 2 * 2048;              // 4 waves x 2 key blocks
    static constexpr int STAGE = KTILE + DS;
    static constexpr int TOTAL = 2 * STAGE;
};

template <typename T, int D>
__global__ void __launch_bounds__(BWD_THREADS, 2) fa_bwd_dq_from_ds_kernel(const KArgs a) {
    using E = Elem<T>;
    constexpr int DBLKS = D / 32;
    constexpr int CPR = D / 8;
    constexpr int CHUNKS = DQ_BN * CPR / BWD_THREADS;
    constexpr int KTILE = DqdsSmem<D>::KTILE;
    constexpr int STAGE = DqdsSmem<D>::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const fa_params& p = a.p;
    const WorkItem w = decode_work(blockIdx.x, p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
    if (!w.valid) return;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SeqGeom sg = seq_geom(p, w.b);
    const int off = sg.off;
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;

    const int64_t kb_off = p.cu_seqlens_k ? 0 : (int64_t)w.b * p.k_batch_stride;
    const uint16_t* kp = reinterpret_cast<const uint16_t*>(p.k) + kb_off + sg.k_row0 * p.k_row_stride + (int64_t)w.hk * p.k_head_stride;
    const int dv = valid_cols(p);
    const __amdgpu_buffer_rsrc_t k_rsrc = make_rsrc(kp, p.k_row_stride, sg.seqlen_k, dv);
    const uint32_t k_tile_bytes = (uint32_t)(DQ_BN * p.k_row_stride * 2);
    constexpr int ROWS_PI = 64 / CPR;
    uint32_t k_voff[CHUNKS];
    int k_lds[CHUNKS];
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int inst = wave * CHUNKS + i;
        const int row = inst * ROWS_PI + lane / CPR;
        const int k_cb = swzt_row_off<D>(row, (lane % CPR) * 16) - row * D * 2;
        k_voff[i] = k_cb < dv * 2 ? (uint32_t)(row * p.k_row_stride * 2 + k_cb) : kOobVoff;
        k_lds[i] = inst * 1024;
    }
    // dS tiles of this (batch, head): one descriptor, tile offsets travel in the scalar offset.
    // LDS image of a tile: [key][64 B = 32 query rows]; DMA lane L of instruction j lands on row
    // 16 j + L/4, 16-byte chunk c = L & 3 (row octet) = global piece (t = c >> 1, g = c & 1).
    const int64_t ds_head = ((int64_t)w.b * p.nheads_q + w.h) * a.ds_nqb * (int64_t)a.ds_nkb * 2048;
    const char* ds_base = reinterpret_cast<const char*>(a.ds_ws) + ds_head;
    const uint32_t ds_lim = (uint32_t)((int64_t)a.ds_nqb * a.ds_nkb * 2048 > 0xffffffffll ? 0xffffffffll
                                                                                          : (int64_t)a.ds_nqb * a.ds_nkb * 2048);
    const __amdgpu_buffer_rsrc_t ds_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(ds_base), 0, (int)__builtin_amdgcn_readfirstlane(ds_lim), 0x00020000);
    uint32_t ds_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 16 * j + (lane >> 2), cch = lane & 3;
        ds_voff[j] = (uint32_t)((cch >> 1) * 1024 + row * 32 + (cch & 1) * 16);
    }
    const int rr = (lane & 15) >> 2;
    const int cb = (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);

    const int n_pass = (a.pair_qblocks && (a.n_qblocks_total - 1 - w.qb) != w.qb) ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
    const int qb_cur = pass == 0 ? w.qb : a.n_qblocks_total - 1 - w.qb;
    const int m_block = qb_cur * DQ_BM;
    if (m_block >= sg.seqlen_q) continue;
    int n_min = 0, n_max = (sg.seqlen_k + DQ_BN - 1) / DQ_BN;
    {
        const int m_last = (m_block + DQ_BM < sg.seqlen_q ? m_block + DQ_BM : sg.seqlen_q) - 1;
        if (wr >= 0) {
            const int kmax = m_last + off + wr;
            const int t = kmax < 0 ? 0 : kmax / DQ_BN + 1;
            n_max = t < n_max ? t : n_max;
        }
        if (wl >= 0) { const int kmin = m_block + off - wl; if (kmin > 0) n_min = kmin / DQ_BN; }
    }
    const int wave_row0 = m_block + wave * 32;
    const int my_row = wave_row0 + l31;
    const int64_t qrow_tiles = (int64_t)(wave_row0 >> 5) * a.ds_nkb;     // tile index of (my 32 rows, key block 0)

    auto active = [&](int nb, int kb) {
        return wave_row0 < sg.seqlen_q && subtile_active(wave_row0, nb * DQ_BN + kb * 32, sg.seqlen_q, sg.seqlen_k, off, wl, wr);
    };
    auto load_tile = [&](int nb, auto stage_c) {
        constexpr int stage = decltype(stage_c)::value;
        char* base = smem + stage * STAGE;
        const uint32_t ks_off = (uint32_t)nb * k_tile_bytes;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(k_rsrc, base + k_lds[i], k_voff[i], ks_off);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (!active(nb, kb)) continue;
            const int64_t tile_off = (qrow_tiles + nb * 2 + kb) * 2048;
            char* dst = base + KTILE + (wave * 2 + kb) * 2048;
#pragma unroll
            for (int j = 0; j < 2; ++j) buf_load_lds_b128(ds_rsrc, dst + j * 1024, ds_voff[j], (uint32_t)tile_off);
        }
    };

    f32x16 dq_acc[DBLKS];
#pragma unroll
    for (int d = 0; d < DBLKS; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq_acc[d][r] = 0.f;

    auto compute = [&](auto stage_c, int nb) {
        constexpr int stage = decltype(stage_c)::value;
        const char* sbase = smem + stage * STAGE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (!active(nb, kb)) continue;
            const char* dsb = sbase + KTILE + (wave * 2 + kb) * 2048;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // B = dS^T: lane (query l31, g) takes keys 16 t + 8 g + (0..7) of its column
                const u32x2 b0 = lds_read_tr16(dsb + (16 * t + 8 * g + rr) * 64 + cb);
                const u32x2 b1 = lds_read_tr16(dsb + (16 * t + 8 * g + 4 + rr) * 64 + cb);
                const u32x4 dsf = {b0[0], b0[1], b1[0], b1[1]};
                // A = K^T in the same key order
                const int row_a = kb * 32 + 16 * t + 8 * g + rr;
#pragma unroll
                for (int d = 0; d < DBLKS; ++d) {
                    const u32x2 a0 = lds_read_tr16(sbase + swzt_row_off<D>(row_a, d * 64 + cb));
                    const u32x2 a1 = lds_read_tr16(sbase + swzt_row_off<D>(row_a + 4, d * 64 + cb));
                    u32x4 af = {a0[0], a0[1], a1[0], a1[1]};
                    dq_acc[d] = E::mfma(af, dsf, dq_acc[d]);
                }
            }
        }
    };
    auto step = [&](auto stage_c, int nb) {
        constexpr int stage = decltype(stage_c)::value;
        if (nb + 1 < n_max) load_tile(nb + 1, std::integral_constant<int, stage ^ 1>{});
        compute(stage_c, nb);
        __syncthreads();
    };

    if (n_min < n_max) load_tile(n_min, std::integral_constant<int, 0>{});
    __syncthreads();
    for (int nb = n_min; nb < n_max; nb += 2) {
        step(std::integral_constant<int, 0>{}, nb);
        if (nb + 1 < n_max) step(std::integral_constant<int, 1>{}, nb + 1);
    }

    if (my_row < sg.seqlen_q) {
        const int64_t dqb = p.cu_seqlens_q ? 0 : (int64_t)w.b * p.dq_batch_stride;
        uint16_t* dqp = reinterpret_cast<uint16_t*>(p.dq) + dqb + (sg.q_row0 + my_row) * p.dq_row_stride + (int64_t)w.h * p.dq_head_stride;
        const float sc = p.softmax_scale;
#pragma unroll
        for (int d = 0; d < DBLKS; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u32x2 o2;
                o2[0] = E::pack2(dq_acc[d][4 * rq + 0] * sc, dq_acc[d][4 * rq + 1] * sc);
                o2[1] = E::pack2(dq_acc[d][4 * rq + 2] * sc, dq_acc[d][4 * rq + 3] * sc);
                if (d * 32 + 8 * rq + 4 * g < dv) *reinterpret_cast<u32x2*>(dqp + d * 32 + 8 * rq + 4 * g) = o2;
            }
    }
    }   // pass
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
// dS hand-off workspace: B * Hq * ceil(Sq/32) * ceil(Sk/32) tiles of 2 KiB (0 = not used: the dQ
// kernel then recomputes S and dP).  OPT-IN through FA_BWD_DS_MAX_GB=<limit>: measured at config 2
// the dQ kernel drops from 0.81 to 0.53 ms, but writing the tiles costs the one-wave-per-SIMD dK/dV
// kernel +0.29 ms (1.55 -> 1.84 ms; store issue is fully exposed there) - a wash that costs 4.3 GB.
int launch_bwd_dkdv_split(const KArgs& a, int grid, hipStream_t stream);
bool bwd_ds2_applicable(const fa_params& p);             // fa_bwd_dq_ds.hip: dS hand-off between the generated dK/dV kernel and a one-GEMM dQ kernel
size_t bwd_ds2_bytes(const fa_params& p);
int launch_bwd_dq_ds(const KArgs& a, hipStream_t stream);
bool bwd_asm_applicable(const KArgs& a);
size_t bwd_asm_workspace_bytes(const fa_params& p);
int launch_bwd_dkdv_asm(const KArgs& a, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// dK/dV launches smaller than the chip.  A dK/dV workgroup owns 128 keys of one (batch, kv-head) and walks every query
// tile of every q-head of the GQA group: batch x kv-heads x key blocks workgroups - 128 of them for a Llama-3 layer at
// micro-batch 1 and 4 k tokens (H 32/8, causal pairs) on 256 CUs, 64 for a cross-attention layer with 77 keys.  Such
// launches divide the query tiles of each pass over `dkv_split` workgroups; every split leaves its fp32 accumulators (dK
// scaled) in the workspace and dkv_reduce_kernel adds them in split order and rounds once: deterministic, and the same
// value as the one-workgroup sum up to the order of the fp32 additions (the reference accumulates a GQA group in one
// block too: kernel/fused_mha_backward.cu:351-474).
// ---------------------------------------------------------------------------------------------
constexpr int DKV_SPLIT_MAX = 8;
constexpr int DKV_SPLIT_MIN_STAGES = 8;                  // query stages x q-heads a split should keep

// slots_per_cu: workgroups of the kernel a CU holds at once; stage_rows: query rows per stage of that kernel; pass_stages: what a
// pass costs besides its stages (work item, K / V tiles, first stage's latency, epilogue), in stage times: 9.2 k + 3.4 k cycles
// against 2.7 k per 64-row stage at D = 64 (phase stamps, profiles/r05_config3_backward.txt section 4) -> 4; the same latencies
// against a 32-row stage of ~1.3-1.5 k cycles in the D = 128 / 256 kernels -> 8.
// Model: the workgroups of a dense launch take equal time (mirrored pairs under a causal mask), so the launch takes
// rounds(workgroups / slots) x (stages / split + passes x pass_stages); the split with the smallest product wins if it beats the
// unsplit launch by 15 % (the partial slabs and the reduction launch are not in the model).  The model's choices against forced
// splits of 2 / 3 / 4 / 8 (build.py --variant ... FA_DKV_SPLIT_FORCE=n, tools/split_factor_sweep.py): it picks the measured best or
// second best on every underfilled shape tried (128 workgroups: 2; 64: 4; D 64, 192 workgroups of 768 slots: 4).
// Packed sequences (flat list of key blocks, no mirrored pairs: the host does not see the lengths): the same model on the AVERAGE
// pass - total_k / 128 + batch key blocks per kv-head, total_q / batch rows per sequence, half of them under a causal-like mask.
static int dkv_split_factor(const fa_params& p, int pair, int slots_per_cu, int stage_rows, int pass_stages) {
    if (p.seqlen_q < 1 || p.seqlen_k < 1 || p.nheads_k < 1 || p.batch < 1) return 1;
    if (p.flags & FA_FLAG_NO_DKV_SPLIT) return 1;
    const bool varlen = p.cu_seqlens_q || p.cu_seqlens_k;
    if (varlen && (!p.cu_seqlens_q || !p.cu_seqlens_k || p.total_q < 1 || p.total_k < 1)) return 1;
    const int64_t n_kblocks = (p.seqlen_k + DKV_BN - 1) / DKV_BN;
    const bool paired = !varlen && pair && n_kblocks >= 2;
    const int64_t wgs = varlen ? ((int64_t)p.total_k / DKV_BN + p.batch) * p.nheads_k
                               : (int64_t)p.batch * p.nheads_k * (paired ? (n_kblocks + 1) / 2 : n_kblocks);
    const int64_t slots = (int64_t)fa_device_cu_count() * slots_per_cu;
    // Only launches that leave slots EMPTY split.  Past one full wave of workgroups the "rounds" of this model do not show on the
    // clock: the socket runs at its power limit, a half-empty second round runs at a higher clock, and the split's own prologues and
    // partials are pure cost - forced splits measured 13-27 % SLOWER at 384 equal workgroups (B 3 x 8 heads, S 4096) and level at best
    // elsewhere (profiles/r05_small_grid.txt section 8).  Packed launches get 1.5 waves: their heavy key blocks come first and
    // a 264-296 workgroup list still ends with a few long passes (+9-13 % there, section 7), at 536 the dispatcher has balanced them.
    if (wgs < 1 || wgs >= (varlen ? 3 * slots / 2 : slots)) return 1;
#ifdef FA_DKV_SPLIT_FORCE                                  // measurement builds (build.py --variant): calibrate the model below
    return FA_DKV_SPLIT_FORCE;
#endif
    // stages of a workgroup (a causal pair walks about one full sequence in its two passes)
    int64_t stages = (int64_t)((p.seqlen_q + stage_rows - 1) / stage_rows) * (p.nheads_q / p.nheads_k);
    if (varlen) {
        const int64_t avg_q = (p.total_q + p.batch - 1) / p.batch;
        stages = ((avg_q + stage_rows - 1) / stage_rows) * (p.nheads_q / p.nheads_k);
        if (pair) stages = (stages + 1) / 2;
    }
    const int64_t fixed = (int64_t)pass_stages * (paired ? 2 : 1);
    // the partial slabs are addressed like dk / dv: 31-bit byte offsets inside one (batch) slice
    const int64_t slice_rows = varlen ? (int64_t)p.total_k : (int64_t)p.seqlen_k;
    if ((slice_rows + DKV_BN) * p.nheads_k * p.head_dim * 4 >= ((int64_t)1 << 31)) return 1;
    int best = 1;
    double best_cost = 0.0, cost1 = 0.0;
    for (int s = 1; s <= DKV_SPLIT_MAX; ++s) {
        if (s > 1 && stages / s < DKV_SPLIT_MIN_STAGES) break;
        const double cost = (double)((wgs * s + slots - 1) / slots) * ((double)stages / s + (double)fixed);
        if (s == 1) { cost1 = best_cost = cost; continue; }
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best_cost <= 0.85 * cost1 ? best : 1;
}
static size_t dkv_split_bytes(const fa_params& p, int split) {
    const size_t rows = p.cu_seqlens_k ? (size_t)p.total_k : (size_t)p.batch * p.seqlen_k;
    return split > 1 ? (size_t)2 * split * rows * p.nheads_k * p.head_dim * sizeof(float) : 0;
}

// dk[b, key, hk, :] = sum over the splits of the fp32 partial rows (split order), rounded once; the same for dv; 8 columns per thread
template <typename T>
__global__ void __launch_bounds__(256) dkv_reduce_kernel(const KArgs a) {
    using E = Elem<T>;
    const fa_params& p = a.p;
    const int D = p.head_dim, cpr = D / 8, dv = valid_cols(p);
    const bool varlen = p.cu_seqlens_k != nullptr;
    const int64_t rows = (varlen ? (int64_t)p.total_k : (int64_t)p.batch * p.seqlen_k) * p.nheads_k;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = idx / cpr;
    const int cc = (int)(idx - row * cpr);
    if (row >= rows || cc * 8 >= dv) return;
    // packed sequences: rows past the last sequence belong to nobody (no workgroup wrote their partials; dk / dv keep what they hold)
    if (varlen && row / p.nheads_k >= p.cu_seqlens_k[p.batch]) return;
    const int which = blockIdx.y;                            // 0: dK, 1: dV
    const int64_t slab = rows * D;
    const float* src = reinterpret_cast<const float*>(a.dkv_part) + (int64_t)which * a.dkv_split * slab + row * D + cc * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < a.dkv_split; ++s) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(src + (int64_t)s * slab);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(src + (int64_t)s * slab + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i] += x0[i]; acc[4 + i] += x1[i]; }
    }
    const int hk = (int)(row % p.nheads_k);
    const int64_t bk = row / p.nheads_k;
    const int64_t key = varlen ? bk : bk % p.seqlen_k;          // (packed: the row of the [total_k, Hk, D] tensor)
    const int64_t b = varlen ? 0 : bk / p.seqlen_k;
    uint16_t* dst = which == 0
        ? reinterpret_cast<uint16_t*>(p.dk) + b * p.dk_batch_stride + key * p.dk_row_stride + (int64_t)hk * p.dk_head_stride
        : reinterpret_cast<uint16_t*>(p.dv) + b * p.dv_batch_stride + key * p.dv_row_stride + (int64_t)hk * p.dv_head_stride;
    u32x4 o4;
#pragma unroll
    for (int i = 0; i < 4; ++i) o4[i] = E::pack2(acc[2 * i], acc[2 * i + 1]);
    *reinterpret_cast<u32x4*>(dst + cc * 8) = o4;
}
template <typename T>
static void launch_dkv_reduce(const KArgs& a, hipStream_t stream) {
    const fa_params& p = a.p;
    const int64_t rows = p.cu_seqlens_k ? (int64_t)p.total_k : (int64_t)p.batch * p.seqlen_k;
    const int64_t total = rows * p.nheads_k * (p.head_dim / 8);
    hipLaunchKernelGGL(dkv_reduce_kernel<T>, dim3((unsigned)((total + 255) / 256), 2), dim3(256), 0, stream, a);
}

static size_t bwd_ds_workspace_bytes(const fa_params& p) {
#ifdef FA_MEASURE
    static const double max_gb = getenv("FA_BWD_DS_MAX_GB") ? atof(getenv("FA_BWD_DS_MAX_GB")) : 0.0;
#else
    constexpr double max_gb = 0.0;                       // the hand-off is a measured net loss: measurement builds only
#endif
    if (p.head_dim > 128 || max_gb <= 0.0) return 0;
    const int64_t nqb = (p.seqlen_q + 31) / 32, nkb = (p.seqlen_k + 31) / 32;
    const int64_t per_head = nqb * nkb * 2048;
    if (per_head <= 0 || per_head >= (int64_t)0xffffffffll) return 0;       // tile offsets are 32-bit
    const double total = (double)per_head * p.batch * p.nheads_q;
    if (total > max_gb * 1073741824.0) return 0;
    return (size_t)total;
}
// workspace of the backward ops: the dS hand-off (measurement builds) or the statistics planes of the asm dK/dV kernel
static KArgs bwd_probe_args(const fa_params& p) {
    KArgs a;
    memset(&a, 0, sizeof(a));
    a.p = p;
    a.has_bias = (p.alibi_slopes != nullptr) || (p.softcap > 0.f);
    a.flat_blocks = p.cu_seqlens_q ? 1 : 0;           // (packed sequences run through the flat work lists unless FA_VARLEN_GRID=1)
    return a;
}
// which dK/dV kernel a dense call takes decides the slots per CU and the stage height behind dkv_split_factor
static int bwd_dkv_split_for(const KArgs& a, bool asm_kernel) {
    const fa_params& p = a.p;
    const bool drop = p.p_dropout > 0.f;
    const bool lin_alibi = p.alibi_slopes && p.softcap <= 0.f && (p.is_causal || p.window_right == 0);
    const bool cap_only = p.softcap > 0.f && !p.alibi_slopes;
    // the predicate of fa_api.hip: make_args (block_m = 128) - the kernels pair key blocks only when a.pair_qblocks is set: a
    // causal call with seqlen_q <= 128 over long keys is NOT paired (the model would otherwise see half its workgroups)
    const int pair = ((p.is_causal || p.window_right >= 0) && p.window_left < 0 && (p.seqlen_q + 127) / 128 >= 2) ? 1 : 0;
    if (asm_kernel) return p.alibi_slopes ? 1 : dkv_split_factor(p, pair, 1, 32, 8);     // (the ALiBi bodies have no partial epilogue)
    if (a.ds_ws) return 1;
    if (p.head_dim > 128)                                  // two waves per key block (fa_bwd_d256.hip): one workgroup per CU
        return ((!a.has_bias || cap_only) && !drop) ? dkv_split_factor(p, pair, 1, 32, 8) : 1;
    if (a.has_bias || drop) return 1;                      // the split (PART) instantiations: the plain scores
    return p.head_dim <= 64 ? dkv_split_factor(p, pair, FA_DKV2_OCC64, 64, 4) : dkv_split_factor(p, pair, 2, 32, 8);
}
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// dS hand-off (fa_bwd_dq_ds.hip): where both generated kernels would run, the launch fills the chip (no split) and the caller asked
// for all three gradients
static bool bwd_takes_ds2(const KArgs& a, bool asm_kernel, int split) {
    return asm_kernel && split <= 1 && bwd_ds2_applicable(a.p);
}
size_t bwd_workspace_bytes(const fa_params& p) {
    const size_t ds = bwd_ds_workspace_bytes(p);
    if (ds > 0) return ds;
    const KArgs a = bwd_probe_args(p);
    const bool asm_kernel = bwd_asm_applicable(a);
    const size_t stats = asm_kernel ? bwd_asm_workspace_bytes(p) : 0;
    const int split = bwd_dkv_split_for(a, asm_kernel);
    if (bwd_takes_ds2(a, asm_kernel, split)) return align256(stats) + bwd_ds2_bytes(p);
    const size_t part = dkv_split_bytes(p, split);
    return part ? align256(stats) + part : stats;
}
#ifdef FA_TIMERS
extern "C" int fa_debug_read_timers(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timers), (size_t)n * 8);
}
#endif


bool bwd_dq_asm_applicable(const KArgs& a);
int launch_bwd_dq_asm(const KArgs& a, hipStream_t stream);

template <typename T, int D>
static int launch_bwd_td(const KArgs& a, hipStream_t stream) {
    const fa_params& p = a.p;
    // outputs the caller did not ask for are not computed (include/fa_mi355.h: dq == NULL, dk == dv == NULL)
    const int g_bwd_phase_mask = 1 | (p.dk ? 2 : 0) | (p.dq ? 4 : 0);
    // no preprocess launch when the dQ kernel recomputes S / dP (every path but the dS hand-off): it computes D in its
    // prologue, runs first and leaves softmax_d (+ the asm dK/dV kernel's statistics planes) behind
    const bool fused_pre = a.fuse_pre != 0;
    // 1. preprocess
    if ((g_bwd_phase_mask & 1) && !fused_pre) {
        const int cpr = D / 8, rows_per_block = 256 / cpr;
        const int64_t total_rows = p.cu_seqlens_q ? (int64_t)p.total_q : (int64_t)p.batch * p.seqlen_q;
        if (total_rows > 0) {
            dim3 grid((unsigned)((total_rows + rows_per_block - 1) / rows_per_block), p.nheads_q);
            hipLaunchKernelGGL(bwd_preprocess_kernel<T>, grid, dim3(256), 0, stream, a);
        }
    }
    const bool drop = p.p_dropout > 0.f;
    // causal ALiBi without softcap: the bias rides on one extra MFMA per sub-tile (BIAS = 2)
    const bool lin_alibi = p.alibi_slopes && p.softcap <= 0.f && (p.is_causal || p.window_right == 0);
    // 2. dK/dV
    auto launch_dkdv = [&]() {
    if (g_bwd_phase_mask & 2) {
        const int n_kblocks = (p.seqlen_k + DKV_BN - 1) / DKV_BN;
        const int n_kb_grid = (a.pair_qblocks && n_kblocks >= 2) ? (n_kblocks + 1) / 2 : n_kblocks;
        const int units = p.batch * p.nheads_k;
        const int grid = unit_grid(units, n_kb_grid);
        const size_t smem = DkvSmem<D>::TOTAL;
#define FA_LAUNCH_DKV(BIAS, DROP)                                                                                 \
        do {                                                                                                      \
            auto kern = fa_bwd_dkdv_kernel<T, D, BIAS, DROP>;                                                     \
            FA_SET_LDS_ONCE(kern, smem); \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(BWD_THREADS), smem, stream, a);                             \
        } while (0)
        // two-workgroups-per-CU kernel where it applies (6 % faster at config 2); FA_DKDV1 forces the other one
#ifdef FA_MEASURE
        static const bool dkv2_env = getenv("FA_DKDV1") == nullptr;
#else
        constexpr bool dkv2_env = true;
#endif
        bool done = false;
        if constexpr (D == 128) {
            if (a.stats_ws && grid > 0) {
                launch_bwd_dkdv_asm(a, stream);             // (grid x a.dkv_split workgroups)
                if (a.dkv_split > 1) launch_dkv_reduce<T>(a, stream);
                done = true;
            }
        }
        if constexpr (D <= 128) {
            if (done) {} else {
            const bool cap_only = p.softcap > 0.f && !p.alibi_slopes;
            if (dkv2_env && (!a.has_bias || ((lin_alibi || cap_only) && !drop)) && grid > 0) {
                const size_t smem2 = Dkv2Smem<D>::TOTAL;
                KArgs a2 = a;                     // varlen: flat list of key blocks for this kernel
                int grid2 = grid;
                if (a.flat_blocks && p.cu_seqlens_k && p.total_k > 0) {
                    a2.flat_kblocks = p.total_k / DKV_BN + p.batch;
                    grid2 = a2.flat_kblocks * p.nheads_k * (a.dkv_split > 1 ? a.dkv_split : 1);
                } else if (a.dkv_split > 1) {
                    grid2 = unit_grid(units, n_kb_grid * a.dkv_split);   // the query tiles of a pass over dkv_split workgroups (dkv_split_factor)
                }
#define FA_LAUNCH_DKV2(BIAS, DROP)                                                                                \
                do {                                                                                              \
                    if (BIAS == 0 && !(DROP) && a.dkv_split > 1) {         /* split launch: the PART instantiations */ \
                        if (D == 128 && valid_cols(p) <= 96) {                                                    \
                            auto kern = fa_bwd_dkdv2_kernel<T, D, 0, false, (D == 128 ? 96 : D), true>;           \
                            FA_SET_LDS_ONCE(kern, smem2);                                                         \
                            hipLaunchKernelGGL(kern, dim3(grid2), dim3(BWD_THREADS), smem2, stream, a2);          \
                        } else {                                                                                  \
                            auto kern = fa_bwd_dkdv2_kernel<T, D, 0, false, D, true>;                             \
                            FA_SET_LDS_ONCE(kern, smem2);                                                         \
                            hipLaunchKernelGGL(kern, dim3(grid2), dim3(BWD_THREADS), smem2, stream, a2);          \
                        }                                                                                         \
                    } else if (D == 128 && valid_cols(p) <= 96) {                                                 \
                        auto kern = fa_bwd_dkdv2_kernel<T, D, BIAS, DROP, (D == 128 ? 96 : D)>;                   \
                        FA_SET_LDS_ONCE(kern, smem2);                                                             \
                        hipLaunchKernelGGL(kern, dim3(grid2), dim3(BWD_THREADS), smem2, stream, a2);              \
                    } else {                                                                                      \
                        auto kern = fa_bwd_dkdv2_kernel<T, D, BIAS, DROP>;                                        \
                        FA_SET_LDS_ONCE(kern, smem2);                                                             \
                        hipLaunchKernelGGL(kern, dim3(grid2), dim3(BWD_THREADS), smem2, stream, a2);              \
                    }                                                                                             \
                } while (0)
                if (a.has_bias && lin_alibi) FA_LAUNCH_DKV2(2, false);
                else if (a.has_bias) FA_LAUNCH_DKV2(3, false);
                else if (drop) FA_LAUNCH_DKV2(0, true);
                else FA_LAUNCH_DKV2(0, false);
#undef FA_LAUNCH_DKV2
                if (a.dkv_split > 1) launch_dkv_reduce<T>(a, stream);
                done = true;
            }
            }
        }
        if constexpr (D == 256) {
            // no bias or softcap only, no dropout: two waves per key block (fa_bwd_dkdv_split_kernel), one sweep instead of two
            if ((!a.has_bias || (p.softcap > 0.f && !p.alibi_slopes)) && !drop && !a.ds_ws && grid > 0) {
                KArgs a2 = a;
                int grid2 = grid;
                if (a.flat_blocks && p.cu_seqlens_k && p.total_k > 0) {
                    a2.flat_kblocks = p.total_k / DKV_BN + p.batch;
                    grid2 = a2.flat_kblocks * p.nheads_k * (a.dkv_split > 1 ? a.dkv_split : 1);
                } else if (a.dkv_split > 1) {
                    grid2 = unit_grid(units, n_kb_grid * a.dkv_split);
                }
                launch_bwd_dkdv_split(a2, grid2, stream);         // fa_bwd_d256.hip
                if (a.dkv_split > 1) launch_dkv_reduce<T>(a, stream);
                done = true;
            }
        }
        if (grid > 0 && !done) {
            if (drop) { if (a.has_bias) FA_LAUNCH_DKV(1, true); else FA_LAUNCH_DKV(0, true); }
            else if (a.has_bias) { if (lin_alibi) FA_LAUNCH_DKV(2, false); else FA_LAUNCH_DKV(1, false); }
            else      FA_LAUNCH_DKV(0, false);
        }
#undef FA_LAUNCH_DKV
    }
    };
    // 3. dQ
    auto launch_dq = [&]() {
    if ((g_bwd_phase_mask & 4) && a.ds_ws) {
        if constexpr (D <= 128) {
            const int grid = work_grid(p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
            const size_t smem = DqdsSmem<D>::TOTAL;
            auto kern = fa_bwd_dq_from_ds_kernel<T, D>;
            FA_SET_LDS_ONCE(kern, smem);
            if (grid > 0) hipLaunchKernelGGL(kern, dim3(grid), dim3(BWD_THREADS), smem, stream, a);
        }
    } else if ((g_bwd_phase_mask & 4) && D == 128 && a.ds2_ws) {
        launch_bwd_dq_ds(a, stream);                      // one GEMM over the handed-off dS tiles (fa_bwd_dq_ds.hip)
    } else if ((g_bwd_phase_mask & 4) && D == 128 && bwd_dq_asm_applicable(a)) {
        launch_bwd_dq_asm(a, stream);                     // hand-scheduled body (fa_bwd_dq_asm.hip)
    } else if (g_bwd_phase_mask & 4) {
        const int grid = a.flat_blocks ? a.flat_blocks * p.nheads_q : work_grid(p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
        const size_t smem = DqSmem<D>::TOTAL;
        // D = 128: two waves per SIMD spill ~10 registers but measure 11 % faster than one wave
        constexpr int OCC = (D > 128) ? 1 : (D <= 64 ? FA_DQ_OCC64 : 2);
#define FA_LAUNCH_DQ(BIAS, DROP)                                                                                  \
        do {                                                                                                      \
            /* dropout needs the Philox registers: two waves per SIMD spill 84 of them (4.0 ms), one wave none */ \
            if ((D == 256 && valid_cols(p) <= 192) || (D == 128 && valid_cols(p) <= 96)) {                          \
                auto kern = fa_bwd_dq_kernel<T, D, BIAS, (DROP) ? 1 : OCC, DROP, (D == 256 ? 192 : (D == 128 ? 96 : D))>; \
                FA_SET_LDS_ONCE(kern, smem);                                                                      \
                hipLaunchKernelGGL(kern, dim3(grid), dim3(BWD_THREADS), smem, stream, a);                         \
            } else {                                                                                              \
                auto kern = fa_bwd_dq_kernel<T, D, BIAS, (DROP) ? 1 : OCC, DROP>;                                 \
                FA_SET_LDS_ONCE(kern, smem);                                                                      \
                hipLaunchKernelGGL(kern, dim3(grid), dim3(BWD_THREADS), smem, stream, a);                         \
            }                                                                                                     \
        } while (0)
        if (grid > 0) {
            if (drop) { if (a.has_bias) FA_LAUNCH_DQ(1, true); else FA_LAUNCH_DQ(0, true); }
            else if (a.has_bias) {
                if (lin_alibi) FA_LAUNCH_DQ(2, false);
                else if (!p.alibi_slopes) FA_LAUNCH_DQ(3, false);            // softcap only
                else FA_LAUNCH_DQ(1, false);
            }
            else      FA_LAUNCH_DQ(0, false);
        }
#undef FA_LAUNCH_DQ
    }
    };
    if (fused_pre) { launch_dq(); launch_dkdv(); } else { launch_dkdv(); launch_dq(); }
    return 0;
}

int launch_bwd(const KArgs& a_in, hipStream_t stream) {
    KArgs a = a_in;
    a.ds_ws = nullptr;
    a.stats_ws = nullptr;
    a.fuse_pre = 0;
    const size_t need = bwd_ds_workspace_bytes(a.p);
    if (need > 0 && a.p.dq && a.p.dk && a.p.workspace && a.p.workspace_bytes >= need) {
        a.ds_ws = a.p.workspace;
        a.ds_nqb = (a.p.seqlen_q + 31) / 32;
        a.ds_nkb = (a.p.seqlen_k + 31) / 32;
    } else {
        if (bwd_asm_applicable(a) && a.p.workspace && a.p.workspace_bytes >= bwd_asm_workspace_bytes(a.p))
            a.stats_ws = reinterpret_cast<float*>(a.p.workspace);     // (without a workspace the hipcc kernels run)
        // every path whose dQ kernel recomputes S / dP: D = rowsum(dO o O) comes out of that kernel's prologue, it runs
        // first, and there is no preprocess launch (one pass over dO and O less: 0.10 of 1.97 ms at config 3)
#ifndef FA_NO_FUSE_PRE
        a.fuse_pre = 1;
#else
        if (a.p.cu_seqlens_q) a.stats_ws = nullptr;      // (A/B build: the preprocess kernel only knows the dense statistics layout)
#endif
        if (!a.p.dq) {
            // dQ not requested: its kernel - and with it the fused row-dot - does not run, so the preprocess kernel
            // provides softmax_d and the statistics planes (dense layout only; packed sequences take the hipcc dK/dV kernel)
            a.fuse_pre = 0;
            if (a.p.cu_seqlens_q) a.stats_ws = nullptr;
        }
    }
    // dS hand-off: preprocess -> generated dK/dV kernel (+ tile stores) -> one-GEMM dQ kernel
    a.ds2_ws = nullptr;
    if (a.stats_ws && !a.ds_ws && bwd_takes_ds2(a, true, bwd_dkv_split_for(a, true))) {
        const size_t off = align256(bwd_asm_workspace_bytes(a.p));
        if (a.p.workspace_bytes >= off + bwd_ds2_bytes(a.p)) {
            a.ds2_ws = reinterpret_cast<char*>(a.p.workspace) + off;
            a.ds2_nqb = (a.p.seqlen_q + 31) / 32;
            a.ds2_nkb = 4 * ((a.p.seqlen_k + 127) / 128);
            a.fuse_pre = 0;                               // the dK/dV kernel runs first: statistics from the preprocess kernel
        }
    }
    // a dense dK/dV launch smaller than the chip: query tiles split over several workgroups + a reduction (dkv_split_factor)
    a.dkv_split = 0;
    a.dkv_part = nullptr;
    const bool packed = a.p.cu_seqlens_q || a.p.cu_seqlens_k;
    if (a.p.dk && !a.ds_ws && !a.ds2_ws && (!packed || (a.flat_blocks && a.p.cu_seqlens_q && a.p.cu_seqlens_k))) {
        const bool asm_kernel = a.p.head_dim == 128 && a.stats_ws != nullptr;
        const int split = bwd_dkv_split_for(a, asm_kernel);
        const size_t off = align256(bwd_asm_applicable(a) ? bwd_asm_workspace_bytes(a.p) : 0);      // as bwd_workspace_bytes lays it out
        if (split > 1 && a.p.workspace && a.p.workspace_bytes >= off + dkv_split_bytes(a.p, split)) {
            a.dkv_split = split;
            a.dkv_part = reinterpret_cast<char*>(a.p.workspace) + off;
        }
    }
    const bool bf = a.p.dtype == FA_BF16;
    switch (a.p.head_dim) {
        case 64:  return bf ? launch_bwd_td<bf16_tag, 64>(a, stream) : launch_bwd_td<fp16_tag, 64>(a, stream);
        case 128: return bf ? launch_bwd_td<bf16_tag, 128>(a, stream) : launch_bwd_td<fp16_tag, 128>(a, stream);
        case 256: return bf ? launch_bwd_td<bf16_tag, 256>(a, stream) : launch_bwd_td<fp16_tag, 256>(a, stream);
        default:  return -2;
    }
}

}  // namespace fa
