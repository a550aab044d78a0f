// fa_bwd_dq_ds.hip - dQ of the dense backward from HANDED-OFF dS tiles (D = 128, no bias / dropout): one GEMM instead of three.
//
// The reference computes dQ in a second kernel that recomputes S and dP (kernel/fused_mha_backward.cu:58-253), and so does this
// library's default path (fa_bwd_dq_asm.hip: 3 GEMMs executed for 1, 0.72 ms at BASELINE config 2 = 15 % of the matrix peak
// algorithmically).  Summing dQ out of the dK/dV kernel with atomics is not an option on this part (profiles/r06_dq_atomics.txt:
// fp32 atomics run at 1.4 TB/s of operands).  What is left is the hand-off: the generated dK/dV kernel (gen_bwd_dkdv_asm.py,
// DKV(ds=True)) stores the packed 16-bit dS tile of every (32-row tile, 32-key block) stage as it is in its registers - two
// 1-KiB stores per wave and stage - and this kernel reads the tiles back and runs the ONE product that is dQ's own:
//
//     dQ^T[d][q] = softmax_scale * sum_keys K^T[d][key] dS^T[key][q]
//
// HBM-bound by construction: 2 bytes per visible (q, key) pair written once and read once (config 2: 2.15 GB each way) against
// 0.27 TFLOP.  Workspace layout (per (batch, q-head)): [32-key block kb][32-row tile qt][2 KiB]; a tile is the dK/dV kernel's
// register image - piece t (1 KiB) x lane (key l31 + 32 g) x 16 bytes = rows 16 t + 4 g + (0..3) and 16 t + 8 + 4 g + (0..3) of
// that key - i.e. a [key][row] matrix whose 4-row groups sit at  t * 1024 + (key + 32 (rg & 1)) * 16 + ((rg >> 1) & 1) * 8,
// rg = row / 4, t = rg / 4: exactly what ds_read_b64_tr_b16 wants to hand every lane (= query row) eight consecutive keys.
//
// Structure: a workgroup owns 256 query rows of one (batch, head) (mirrored block pairs under a causal mask, units placed per
// XCD as everywhere), a wave 64 of them = two row tiles.  A stage is 32 keys: the K rows (8 KiB, shared: each wave fetches two of
// the eight 1-KiB pieces) and every wave's own two dS tiles (4 KiB), all by LDS-DMA into a SIX-stage ring behind a counted
// vmcnt - the kernel waits for memory, not for the matrix pipe (16 MFMAs per stage and wave), so what matters is bytes in
// flight: five stages = 80 KiB of dS per CU (a first version with wave-private three-stage rings, 32 KiB in flight, streamed
// the tiles at 3.5 TB/s - latency-bound).  One barrier per stage (K is shared); dS, the HBM stream, is read exactly once.
#include <type_traits>
#include "fa_common.h"

namespace fa {

constexpr int DQS_BM = 256;                      // query rows per workgroup (64 per wave)
#ifndef DQS_NST_V
#define DQS_NST_V 6
#endif
constexpr int DQS_NST = DQS_NST_V;                       // ring stages: five 32-key stages (80 KiB of dS per CU) in flight
constexpr int DQS_KT = 32 * 256;                 // K rows of one 32-key stage (shared by the four waves)
constexpr int DQS_STAGE = DQS_KT + 4 * 2 * 2048; // + two dS tiles per wave
constexpr int DQS_LDS = DQS_NST * DQS_STAGE;     // 144 KiB
constexpr int DQS_DMA = 6;                       // LDS-DMA instructions per wave and stage (2 of the 8 K pieces + 4 dS)

// tiles the dK/dV kernel wrote for key block kb (its workgroup = 128 keys): query tiles [mt0, mt1) (fa_bwd_asm.hip)
__device__ __forceinline__ void dqs_written_tiles(int kb, int seqlen_q, int seqlen_k, int off, int wl, int wr, int& mt0, int& mt1) {
    const int n0 = (kb >> 2) * 128;
    const int n_last = (n0 + 128 < seqlen_k ? n0 + 128 : seqlen_k) - 1;
    int m_lo = 0, m_hi = seqlen_q;
    if (wr >= 0) { const int t = n0 - off - wr; m_lo = t > 0 ? t : 0; }
    if (wl >= 0) { const int t = n_last - off + wl + 1; m_hi = t < m_hi ? t : m_hi; }
    mt0 = m_lo / 32;
    mt1 = m_hi > m_lo ? (m_hi + 31) / 32 : mt0;
}

template <typename T>
__global__ void __launch_bounds__(256, 1) fa_bwd_dq_ds_kernel(const KArgs a) {
    using E = Elem<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const fa_params& p = a.p;
    const WorkItem w = decode_work(blockIdx.x, p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
    if (!w.valid) return;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seqlen_q = p.seqlen_q, seqlen_k = p.seqlen_k;
    const int off = seqlen_k - seqlen_q;
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;
    const int nqb = a.ds2_nqb, nkb = a.ds2_nkb;

    const uint16_t* kp = reinterpret_cast<const uint16_t*>(p.k) + (int64_t)w.b * p.k_batch_stride + (int64_t)w.hk * p.k_head_stride;
    const __amdgpu_buffer_rsrc_t k_rsrc = make_rsrc(kp, p.k_row_stride, seqlen_k, 128);
    const uint32_t k_row_bytes = (uint32_t)p.k_row_stride * 2;
    const int64_t head_bytes = (int64_t)nkb * nqb * 2048;
    const char* ds_base = reinterpret_cast<const char*>(a.ds2_ws) + ((int64_t)w.b * p.nheads_q + w.h) * head_bytes;
    const __amdgpu_buffer_rsrc_t ds_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(ds_base), 0, (int)__builtin_amdgcn_readfirstlane((uint32_t)head_bytes), 0x00020000);

    // K: instruction i covers rows 4 i .. 4 i + 3; the lane's 16-byte chunk comes from the swizzled source column
    uint32_t k_voff[2];                                   // (this wave's pieces 2 wave, 2 wave + 1 of the stage's eight)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * (2 * wave + i) + (lane >> 4);
        const int cbs = swzt_row_off<128>(row, (lane & 15) * 16) - row * 256;
        k_voff[i] = (uint32_t)row * k_row_bytes + (uint32_t)cbs;
    }
    const uint32_t ds_voff0 = (uint32_t)lane * 16u, ds_voff1 = ds_voff0 + 1024u;
    char* const ring = smem;

    // lane constants of the transposing reads: the lane supplies key row (lane & 15) >> 2 of a 4-key group and 4 columns
    const int rr = (lane & 15) >> 2;
    const int cb = (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);                    // K: column byte inside a 64-byte d group
    // dS tile: query group 16 ((lane >> 4) & 1), 4-row group (lane & 3): piece t = (lane >> 4) & 1, g bit = lane & 1, half = (lane >> 1) & 1
    const int ds_lane = ((lane >> 4) & 1) * 1024 + (lane & 1) * 512 + ((lane >> 1) & 1) * 8 + rr * 16;
    const lds_char* k_rp[4][2];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) k_rp[d][h2] = lds_pin(ring + swzt_row_off<128>(8 * g + rr + 4 * h2, d * 64 + cb));
    const lds_char* ds_rp = lds_pin(ring + DQS_KT + wave * 4096 + ds_lane + 8 * g * 16);     // + keys 8 g (16 bytes per key)

    const int n_pass = (a.pair_qblocks && (a.n_qblocks_total - 1 - w.qb) != w.qb) ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
        const int qb_cur = pass == 0 ? w.qb : a.n_qblocks_total - 1 - w.qb;
        const int rows0 = qb_cur * DQS_BM + wave * 64;
        const int qt0 = rows0 >> 5;
        // key blocks the rows [r0, r0 + n) can see
        auto key_range = [&](int r0, int n, int& lo, int& hi) {
            lo = hi = 0;
            if (r0 >= seqlen_q) return;
            hi = (seqlen_k + 31) / 32;
            const int r_last = (r0 + n - 1 < seqlen_q ? r0 + n - 1 : seqlen_q - 1);
            if (wr >= 0) { const int kmax = r_last + off + wr; const int t = kmax < 0 ? 0 : kmax / 32 + 1; hi = t < hi ? t : hi; }
            if (wl >= 0) { const int kmin = r0 + off - wl; lo = kmin > 0 ? kmin / 32 : 0; }
            if (lo > hi) lo = hi;
        };
        int kb_lo, kb_hi, wv_lo, wv_hi;                      // the workgroup's loop range, this wave's own
        key_range(qb_cur * DQS_BM, DQS_BM, kb_lo, kb_hi);
        key_range(rows0, 64, wv_lo, wv_hi);

        auto issue_stage = [&](int kb, int slot) {           // 6 LDS-DMA instructions, whatever kb is (uniform vmcnt arithmetic)
            char* base = ring + slot * DQS_STAGE;
            const uint32_t ks = kb < kb_hi ? (uint32_t)kb * 32u * k_row_bytes : kOobVoff;
#pragma unroll
            for (int i = 0; i < 2; ++i) buf_load_lds_b128(k_rsrc, base + (2 * wave + i) * 1024, k_voff[i], ks);
            const bool real = kb >= wv_lo && kb < wv_hi;     // (outside the wave's own range: zeros - its MFMAs add nothing)
            int mt0 = 0, mt1 = 0;
            if (real) dqs_written_tiles(kb, seqlen_q, seqlen_k, off, wl, wr, mt0, mt1);
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
                const int qt = qt0 + tq;
                const uint32_t so = (real && qt >= mt0 && qt < mt1) ? (uint32_t)(((int64_t)kb * nqb + qt) * 2048) : kOobVoff;
                buf_load_lds_b128(ds_rsrc, base + DQS_KT + wave * 4096 + tq * 2048, ds_voff0, so);
                buf_load_lds_b128(ds_rsrc, base + DQS_KT + wave * 4096 + tq * 2048 + 1024, ds_voff1, so);
            }
        };

        f32x16 acc[2][4];
#pragma unroll
        for (int tq = 0; tq < 2; ++tq)
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tq][d][r] = 0.f;

        // ---- software pipeline (per wave): the 12 transposing reads of a 16-key step are issued one step AHEAD of its eight MFMAs,
        // so every read has a whole MFMA group (256 matrix-pipe cycles) to return; the wait + barrier for stage kb + 1 and the
        // DMA of stage kb + 5 sit between the two steps of stage kb.  Ring: stage kb being read, kb + 1 .. kb + 4 landed or in
        // flight, kb + 5 issued into the slot stage kb - 1 left (every wave is past it when it reaches the mid-stage barrier).
        struct Frag { u32x4 b[2], a[4]; };
        auto reads = [&](int so, auto t_c, Frag& f) {
            constexpr int t = decltype(t_c)::value;
            const lds_char* ds_s = ds_rp + so;
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
                const u32x2 b0 = lds_read_tr16_nw(ds_s, tq * 2048 + t * 256);                   // keys 16 t + 8 g + (0..3)
                const u32x2 b1 = lds_read_tr16_nw(ds_s, tq * 2048 + t * 256 + 64);              // ... + 4
                f.b[tq] = u32x4{b0[0], b0[1], b1[0], b1[1]};
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const u32x2 a0 = lds_read_tr16_nw(k_rp[d][0] + so, t * 16 * 256);
                const u32x2 a1 = lds_read_tr16_nw(k_rp[d][1] + so, t * 16 * 256);
                f.a[d] = u32x4{a0[0], a0[1], a1[0], a1[1]};
            }
        };
        auto mfmas = [&](Frag& f, int) {                     // twelve younger reads (the next group) may be in flight
            asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]));
#ifndef DQS_KO_MFMA                                      // (timing experiment: -DDQS_KO_MFMA streams the tiles without the products)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                acc[0][d] = E::mfma(f.a[d], f.b[0], acc[0][d]);
                acc[1][d] = E::mfma(f.a[d], f.b[1], acc[1][d]);
            }
#endif
        };
        __builtin_amdgcn_s_barrier();                                     // the previous pass is done with the ring
#pragma unroll
        for (int i = 0; i < DQS_NST - 2; ++i) issue_stage(kb_lo + i, i);  // stages 0 .. 3
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DQS_NST - 3) * DQS_DMA) : "memory");       // my pieces of the first stage
        __builtin_amdgcn_s_barrier();
        issue_stage(kb_lo + DQS_NST - 2, DQS_NST - 2);                    // stage 4
        Frag r0, r1;
        int slot = 0;
        reads(0, std::integral_constant<int, 0>{}, r0);
        for (int kb = kb_lo; kb < kb_hi; ++kb) {
            const int so = slot * DQS_STAGE;
            const int nslot = slot + 1 < DQS_NST ? slot + 1 : 0;
            reads(so, std::integral_constant<int, 1>{}, r1);
            mfmas(r0, 12);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DQS_NST - 3) * DQS_DMA) : "memory");   // my pieces of stage kb + 1 (loads return in order)
            __builtin_amdgcn_s_barrier();                                 // ... everybody's; and everybody is past stage kb - 1
            issue_stage(kb + DQS_NST - 1, slot >= 1 ? slot - 1 : DQS_NST - 1);
            reads(nslot * DQS_STAGE, std::integral_constant<int, 0>{}, r0);      // (past the last stage: a zero-filled look-ahead stage, unused)
            mfmas(r1, 12);
            slot = nslot;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0.b[0]), "+v"(r0.b[1]), "+v"(r0.a[0]), "+v"(r0.a[1]), "+v"(r0.a[2]), "+v"(r0.a[3]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the look-ahead stages (zeros past the range)

        // ---- epilogue: dQ = softmax_scale * acc, 16 bit; lane = query row, registers = d
        const float sc = p.softmax_scale;
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
            const int my_row = rows0 + 32 * tq + l31;
            if (my_row < seqlen_q) {
                uint16_t* dqp = reinterpret_cast<uint16_t*>(p.dq) + (int64_t)w.b * p.dq_batch_stride + (int64_t)my_row * p.dq_row_stride +
                                (int64_t)w.h * p.dq_head_stride;
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        u32x2 o2;
                        o2[0] = E::pack2(acc[tq][d][4 * rq + 0] * sc, acc[tq][d][4 * rq + 1] * sc);
                        o2[1] = E::pack2(acc[tq][d][4 * rq + 2] * sc, acc[tq][d][4 * rq + 3] * sc);
                        *reinterpret_cast<u32x2*>(dqp + d * 32 + 8 * rq + 4 * g) = o2;
                    }
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// OPT-IN (fa_params::flags & FA_FLAG_DS_HANDOFF; the Python layer sets it when FA_BWD_DS=1 is in the environment at import).
// Measured on BASELINE config 2 (profiles/r06_ds_handoff.txt): the tile stores cost the dK/dV kernel +0.14 ms (the 2.15 GB write
// stream itself - not instruction issue, not the counted waits, not the cache policy bits), this kernel takes 0.52 ms (tiles
// streamed at 4.9 TB/s) where the recomputing dQ kernel takes 0.74, and a preprocess launch returns (0.07 ms): break-even.
size_t bwd_ds2_bytes(const fa_params& p) {
    const int64_t nqb = (p.seqlen_q + 31) / 32, nkb = 4 * (int64_t)((p.seqlen_k + 127) / 128);
    return (size_t)(nqb * nkb * 2048) * (size_t)p.batch * (size_t)p.nheads_q;
}
bool bwd_ds2_applicable(const fa_params& p) {
    if (!(p.flags & FA_FLAG_DS_HANDOFF)) return false;
    if (p.head_dim != 128 || (p.head_dim_v != 0 && p.head_dim_v != 128)) return false;
    if (p.cu_seqlens_q || p.cu_seqlens_k || p.block_table || p.alibi_slopes || p.softcap > 0.f || p.p_dropout > 0.f) return false;
    if (!p.dq || !p.dk || !p.dv) return false;
    if (p.seqlen_q < 1 || p.seqlen_k < 1) return false;
    const int64_t nqb = (p.seqlen_q + 31) / 32, nkb = 4 * (int64_t)((p.seqlen_k + 127) / 128);
    const int64_t group = p.nheads_q / p.nheads_k;
    if (nqb * nkb * 2048 * group >= ((int64_t)1 << 31)) return false;        // the dK/dV kernel's descriptor spans the kv-head's group
    if ((int64_t)(p.seqlen_q + 256) * p.dq_row_stride * 2 >= ((int64_t)1 << 31)) return false;
    return true;
}

template <typename T>
static int launch_bwd_dq_ds_t(const KArgs& a0, hipStream_t stream) {
    KArgs a = a0;
    const fa_params& p = a.p;
    a.n_qblocks_total = (p.seqlen_q + DQS_BM - 1) / DQS_BM;
    a.pair_qblocks = ((p.is_causal || p.window_right >= 0) && p.window_left < 0 && a.n_qblocks_total >= 2) ? 1 : 0;
    a.n_qblocks = a.pair_qblocks ? (a.n_qblocks_total + 1) / 2 : a.n_qblocks_total;
    const int grid = work_grid(p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
    if (grid == 0) return 0;
    auto kern = fa_bwd_dq_ds_kernel<T>;
    FA_SET_LDS_ONCE(kern, DQS_LDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), DQS_LDS, stream, a);
    return 0;
}

int launch_bwd_dq_ds(const KArgs& a, hipStream_t stream) {
    return a.p.dtype == FA_BF16 ? launch_bwd_dq_ds_t<bf16_tag>(a, stream) : launch_bwd_dq_ds_t<fp16_tag>(a, stream);
}

}  // namespace fa
