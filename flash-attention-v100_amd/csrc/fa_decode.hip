// fa_decode.hip - split-KV decode kernel for flash_attn_with_kvcache (small T_q).
//
// The reference runs decode as one 512-thread CTA per (batch, q-head) with a 32/64-row Q tile
// holding ONE valid row, no GQA sharing and no split-KV (kernel/fused_mha_forward_kvcache.cu:344,462):
// every q-head re-reads its kv-head.  On MI355X decode is HBM-bound (K+V read once),
// so this kernel is organised around bytes, not FLOPs:
//   * one workgroup per (batch, KV-head, split): the T_q x G query rows that share a kv-head are
//     packed into ONE 32-row MFMA tile, so the KV stream is read once per kv-head (GQA packing);
//   * the waves of the workgroup - eight, two per SIMD, for head dims up to 128 (one LDS stage and one register set
//     each, Q fragments shared in LDS); four for D = 256 - take alternate 32-key tiles of the split's key range (no
//     barrier in the loop: every wave stages ITS tiles through a private LDS region with fully coalesced 16-byte
//     loads), then merge (m, l, O) through LDS once;
//   * optional num_splits > 1 writes normalised partial O + LSE; decode_combine_kernel merges;
//   * paged KV (block_table), cache_batch_idx, cache_leftpad, in-kernel RoPE on Q, causal /
//     window masks, and fp8-e4m3 K/V (dequantised while staging; k_descale folds into the softmax
//     scale, v_descale into the final normalisation).
// ALiBi / softcap: per-element score path of fa_decode_kernel (the token-major kernels step aside).
// Head dims: widths 64 / 128 / 256 (256: 16-key tiles), narrower rows through the NARROW instantiations.
#include <atomic>
#include "fa_common.h"
#include "fa_rope.h"

namespace fa {

#define FA_DEC_PIN(o) asm volatile("" : "+v"(o))
#define FA_DEC_UNIFORM(x) (x)

#ifndef FA_DEC_F8M
#define FA_DEC_F8M 1                           // fp8 caches at D = 128 on the fp8-operand MFMA (0: dequantise to 16 bit while staging)
#endif
constexpr int DEC_THREADS = 256;
constexpr int DEC_BN = 32;                     // keys per wave tile
constexpr float DEC_RESCALE_THR = 8.0f;        // log2 units

template <int D, int NW = 4> struct DecSmem {
    // D = 256: 16-key tiles (the same 8 KiB per tile and the same staging registers as 32 keys at D = 128); the S^T MFMA
    // still spans 32 key rows - the upper 16 are masked - and P V takes one 16-key k-step
    static constexpr int BN = D > 128 ? 16 : DEC_BN;
    static constexpr int TILE = BN * D * 2;                // one 16-bit K (or V) tile
    // four waves (one per SIMD): two LDS stages per wave; eight waves (two per SIMD, 256 registers each): one stage
    static constexpr int STAGES = NW == 8 ? 1 : 2;
    static constexpr int WAVE = STAGES * 2 * TILE;         // stages x (K + V), private per wave
    static constexpr int MERGE = 2 * NW * 32 * 4 + NW * 32 * D * 4;   // the waves' (m, l, O) at the end
    // D = 256 and the eight-wave form: the Q fragments (the same for all waves: D / 16 k-steps x 64 lanes x 16 B) live in
    // LDS behind the waves' tiles instead of D / 4 registers per lane
    static constexpr int QOFF = NW * WAVE;
    static constexpr int QBYTES = (D > 128 || NW == 8) ? (D / 16) * 64 * 16 : 0;
    static constexpr int TOTAL = QOFF + QBYTES > MERGE ? QOFF + QBYTES : MERGE;
};

// keys in the cache of batch entry b: cache_seqlens[b] (kv-cache op), or - decode issued through the varlen op - the
// cu_seqlens_k difference, clamped by seqused_k when both are given (include/template.h:65-68)
__device__ __forceinline__ int dec_cache_len(const fa_params& p, int b) {
    int L = p.cache_seqlens ? p.cache_seqlens[b] : 0;
    if (p.cu_seqlens_k) {
        const int d = p.cu_seqlens_k[b + 1] - p.cu_seqlens_k[b];
        L = p.cache_seqlens ? (L > 0 ? (L < d ? L : d) : 0) : d;
    }
    return L;
}

struct DecArgs {
    KArgs a;
    int n_splits;
    int rows;                  // T_q * G; a workgroup takes 32 of them (row block)
    int n_rb;                  // row blocks
    int bias;                  // ALiBi slopes or softcap: per-element score path in fa_decode_kernel
    const int32_t* cu_q;       // varlen-q mode (mixed batch through the varlen op): sequence b brings cu_q[b+1] - cu_q[b] query rows at
                               // packed row cu_q[b]; sequences with 0 or more than p.seqlen_q (= the class bound) rows are not ours
    int grid_splits;           // fa_decode_kernel: key splits of the grid (= n_splits there)
    int group;                 // G
    int local;                 // RoPE position advances with the query row (causal / window)
    int page_shift;            // log2(page_block_size) or -1
    int ksub;                  // token-major kernel: waves per head group = partial rows per grid split (1, 2, 4)
    float* o_partial;          // [n_splits, B, Hq, T_q, D] fp32 (n_splits > 1)
    float* lse_partial;        // [n_splits, B, Hq, T_q]
};

// F8M (fp8 cache, D = 128, eight waves): the cache bytes feed the matrix pipe AS STORED - v_mfma_f32_32x32x16_fp8_fp8 for both
// GEMMs - instead of being dequantised to 16 bit on their way into LDS (~480 VALU instructions per 8 KiB tile: the kernel
// was instruction-bound at 5.1 TB/s where the 16-bit cache streams at 5.9).  The 16-bit side of each product travels as an
// fp8 PAIR so that nothing is lost against the 16-bit path: Q rows are scaled to the e4m3 range per row (the factor joins
// k_descale in the softmax scale, a lane owns one query row) and split into head + remainder (two fp8 fragments, 3 + 3
// mantissa bits and the remainder's own exponent: ~2^-8 relative, bf16's resolution), P likewise (its values are <= 2^8
// under the deferred rescale).  Two MFMAs per k-step instead of one - the matrix pipe idles in decode - and V^T comes out of
// LDS through ds_read_b64_tr_b8 (tools/probes/probe_fp8_mfma.hip: operand and transpose layouts).
// HPW ("a head per wave"; eight waves, kv-heads adjacent in a cache row, one 32-row block): the eight waves of a workgroup take
// eight CONSECUTIVE kv-heads over the SAME key range instead of alternate tiles of one head.  A head's slice of a cache row is
// 128 / 256 bytes, the next token's slice sits a whole row further: a workgroup that streams one head touches the cache in
// such pieces (the pure-load ceiling of that pattern was 6.1 TB/s, tools/probes/probe_kv_stream.hip), eight waves side by side
// read whole rows (7.1).  Every wave owns its head's complete result for the split: no merge, no barrier, Q in registers.
template <typename T, int D, bool KV8, bool PAGED, bool NARROW = false, int NW = 4, bool F8M = false, bool HPW = false>
__global__ void __launch_bounds__(64 * NW, 1) fa_decode_kernel(const DecArgs da) {
    using E = Elem<T>;
    static_assert(!F8M || (KV8 && D == 128 && NW == 8 && !NARROW), "fp8 MFMA form: fp8 cache, D = 128, eight waves");
    static_assert(!HPW || (NW == 8 && D <= 128 && !NARROW), "a head per wave: eight waves");
    constexpr int KSTEPS = D / 16;
    constexpr int DBLKS = D / 32;
    static_assert(!HPW || F8M, "a head per wave: the fp8-operand form (a 16-bit build with 16-key LDS-DMA tiles measured level in rounds 4 "
                               "and 5 - 5.6-5.8 TB/s - and lives in tools/experiments/decode_round4_5_experiments.patch)");
    constexpr int TILE = DecSmem<D, NW>::TILE;
    constexpr int EB = KV8 ? 1 : 2;                         // bytes per cache element
    constexpr int CPR = D * EB / 16;                        // 16-byte chunks per cache row
    constexpr int BN = DecSmem<D, NW>::BN;                  // keys per wave tile
    constexpr int CH = BN * CPR / 64;                       // chunks per lane per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const KArgs& a = da.a;
    const fa_params& p = a.p;
    // grid (units, splits, 1) with one row block; with several - every row block streams the kv-head's cache again - a
    // 1-D grid places the row blocks of a (unit, split) on ONE XCD, back to back: workgroup ids go round-robin over the 8
    // XCDs, each with its own L2, so id = ((pair / 8) * n_rb + rb) * 8 + pair % 8 keeps them 8 ids apart.  The second
    // and later row blocks then read the stream from L2 instead of HBM (tools/spec_decode_sweep.py).
    int unit = blockIdx.x, split = blockIdx.y, rb = 0;      // unit = (b, hk)
    if (da.n_rb > 1) {
        const int slot = blockIdx.x & 7, rest = blockIdx.x >> 3;
        rb = rest % da.n_rb;
        const int pair = (rest / da.n_rb) * 8 + slot;
        const int n_units = p.batch * p.nheads_k;
        if (pair >= n_units * da.grid_splits) return;
        unit = pair % n_units;
        split = pair / n_units;
    }
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int units_per_b = HPW ? p.nheads_k / NW : p.nheads_k;          // HPW: unit = (b, group of NW kv-heads)
    const int b = unit / units_per_b;
    const int hk = HPW ? (unit - b * units_per_b) * NW + wave : unit - b * units_per_b;
    char* wsm = smem + wave * DecSmem<D, NW>::WAVE;

    const int L = dec_cache_len(p, b);
    const int lp = p.cache_leftpad ? p.cache_leftpad[b] : 0;
    const int cb = p.cache_batch_idx ? p.cache_batch_idx[b] : b;
    const int seqlen_k = L + p.seqlen_new;
    int Tq = p.seqlen_q, q_row0 = 0;
    const int G = da.group;
    if (da.cu_q) {
        q_row0 = da.cu_q[b];
        const int ql = da.cu_q[b + 1] - q_row0;
        if (ql < 1 || ql > p.seqlen_q) return;              // (uniform per workgroup: before any barrier)
        Tq = ql;
    }
    const int R = da.cu_q ? Tq * G : da.rows;
    if (32 * (da.n_rb > 1 ? rb : 0) >= R) return;           // a row block past this sequence's rows
    const int off = seqlen_k - Tq;
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;

    // ---- my packed query row: r = t * G + gq; blockIdx.z picks the 32-row block (multi-token queries over an fp8 cache:
    //      every row block streams the kv-head's cache again, dequantised into the same LDS tiles) ----
    const int rbase = 32 * rb;
    const int r = rbase + l31;
    const int t_row = r / G, gq = r - t_row * G;
    const int h = hk * G + gq;
    const bool row_ok = r < R;
    int lo = 0, hi = seqlen_k - 1;
    if (wr >= 0) { const int h2 = t_row + off + wr; hi = h2 < hi ? h2 : hi; }
    if (wl >= 0) { const int l2 = t_row + off - wl; lo = l2 > lo ? l2 : lo; }
    // wave-uniform bounds for mask elision: a tile inside [max lo, min hi] of the wave's valid rows needs no per-element
    // compare / select (64 of the ~540 instructions of a tile step); rows past R see unmasked garbage that is never stored
    int w_lo_max = row_ok ? lo : 0, w_hi_min = row_ok ? hi : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int a2 = __shfl_xor(w_lo_max, o), b2 = __shfl_xor(w_hi_min, o);
        w_lo_max = a2 > w_lo_max ? a2 : w_lo_max;
        w_hi_min = b2 < w_hi_min ? b2 : w_hi_min;
    }
    w_lo_max = __builtin_amdgcn_readfirstlane(w_lo_max);
    w_hi_min = __builtin_amdgcn_readfirstlane(w_hi_min);
    const bool any_row = __builtin_amdgcn_readfirstlane((int)(__ballot(row_ok) != 0ull));
    if (!row_ok) { lo = 0x7fffffff; hi = -1; }

    // NARROW: rows of head_dim_v (< D, a multiple of 8) valid columns in a D-wide kernel (D = 96 on the 128 width ...):
    // the missing columns are zeros in Q and in the LDS tiles, and are not written
    const int vcols = NARROW ? p.head_dim_v : D;
    // ---- Q fragments (B operand), RoPE applied in registers ----
    constexpr bool Q_LDS = !HPW && DecSmem<D, NW>::QBYTES > 0;          // (HPW: every wave has its own head's rows - registers)
    u32x4 qf[Q_LDS ? 1 : KSTEPS];
    float q_unscale = 1.0f;                                 // F8M: what this lane's query row was divided by on its way to fp8
#ifndef FA_DEC_F8M_TERMS
#define FA_DEC_F8M_TERMS 2
#endif
    constexpr int NTQ = FA_DEC_F8M_TERMS;                   // F8M: fp8 terms per Q element and per probability (2: ~2^-8 relative, 3: ~2^-12)
    constexpr int NTP = FA_DEC_F8M_TERMS;
    // e4m3 bottoms out at 2^-9 ABSOLUTE (subnormals), remainders included: the probabilities must not sit near it.  So this form
    // rescales on every new row maximum (P <= 1 instead of <= 2^8 under the deferred rescale) and converts 2^8 P: what is lost
    // lies below 2^-17 of the row's largest probability.  The 2^-8 returns in the final normalisation (exact).
    constexpr float RESCALE_THR = F8M ? 0.0f : DEC_RESCALE_THR;
    constexpr float P_UP = F8M ? 256.0f : 1.0f;
    {
        const uint16_t* qrow = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.q_batch_stride +
                               (int64_t)(q_row0 + t_row) * p.q_row_stride + (int64_t)h * p.q_head_stride;
        const int half = p.rotary_dim >> 1;
        const int pos = L + lp + (da.local ? t_row : 0);
        const uint16_t* cosp = reinterpret_cast<const uint16_t*>(p.rotary_cos) + (int64_t)pos * half;
        const uint16_t* sinp = reinterpret_cast<const uint16_t*>(p.rotary_sin) + (int64_t)pos * half;
        u32x4 xq[F8M ? KSTEPS : 1];
        float amax = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int d_base = 16 * ks + 8 * g;
            u32x4 x = {0, 0, 0, 0};
            if (row_ok && (!NARROW || d_base < vcols)) {
                x = *reinterpret_cast<const u32x4*>(qrow + d_base);
                if (p.rotary_dim > 0 && d_base < p.rotary_dim && pos >= 0 && pos < p.seqlen_ro) {
                    u32x4 xp = x;
                    if (!p.rotary_interleaved) {
                        const int pd = d_base < half ? d_base + half : d_base - half;
                        xp = *reinterpret_cast<const u32x4*>(qrow + pd);
                    }
                    rope_chunk<T>(x, xp, cosp, sinp, d_base, p.rotary_dim, p.rotary_interleaved != 0);
                }
            }
            if constexpr (F8M) {
                xq[ks] = x;
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) amax = fmaxf(amax, fmaxf(fabsf(E::lo(x[w2])), fabsf(E::hi(x[w2]))));
            } else if constexpr (Q_LDS) { if (wave == 0) lds_write_b128(smem + DecSmem<D, NW>::QOFF + (ks * 64 + lane) * 16, x); }
            else qf[ks] = x;
        }
        if constexpr (F8M) {
            // the row's largest magnitude -> 384 (e4m3 tops out at 448); head + remainder per element, 8 + 8 bytes per k-step
            amax = xhalf_max(amax);
            const float inv_q = amax > 0.f ? 384.0f / amax : 1.0f;
            q_unscale = amax > 0.f ? amax * (1.0f / 384.0f) : 1.0f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                float a8[8];
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) { a8[2 * w2] = E::lo(xq[ks][w2]) * inv_q; a8[2 * w2 + 1] = E::hi(xq[ks][w2]) * inv_q; }
                u32x2 qt[NTQ];
                fp8_terms8<NTQ>(a8, qt);
                const u32x4 hl = {qt[0][0], qt[0][1], qt[1][0], qt[1][1]};
                if constexpr (!Q_LDS) { static_assert(Q_LDS || NTQ == 2, "register Q fragments: two terms"); qf[ks] = hl; }
                else if (wave == 0) {  // head and second term where the 16-bit form keeps its fragment, a third behind the Q area
                    lds_write_b128(smem + DecSmem<D, NW>::QOFF + (ks * 64 + lane) * 16, hl);
                    if constexpr (NTQ > 2) *reinterpret_cast<u32x2*>(smem + DecSmem<D, NW>::QOFF + DecSmem<D, NW>::QBYTES + (ks * 64 + lane) * 8) = qt[NTQ - 1];
                }
            }
        }
        (void)xq; (void)amax;
    }
    if constexpr (Q_LDS) __syncthreads();
    (void)qf;

    // ---- key-tile range of this split / this wave ----
    int tile_lo = 0;
    const int t_first = rbase / G;                                       // query positions of this row block
    const int t_last = (rbase + 31) / G < Tq - 1 ? (rbase + 31) / G : Tq - 1;
    if (wl >= 0) { const int kmin = t_first + off - wl; if (kmin > 0) tile_lo = kmin / BN; }
    int tile_hi = (seqlen_k + BN - 1) / BN;
    if (wr >= 0) { const int kmax = t_last + off + wr; const int t2 = kmax < 0 ? 0 : kmax / BN + 1; tile_hi = t2 < tile_hi ? t2 : tile_hi; }
    const int n_all = tile_hi > tile_lo ? tile_hi - tile_lo : 0;
    const int per_split = HPW ? (n_all + da.n_splits - 1) / da.n_splits
                              : ((n_all + da.n_splits - 1) / da.n_splits + NW - 1) / NW * NW;     // multiple of the waves
    const int s_lo = tile_lo + split * per_split;
    int s_hi = s_lo + per_split; s_hi = s_hi < tile_hi ? s_hi : tile_hi;

    // ---- staging (wave private) ----
    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k) + (int64_t)hk * p.k_head_stride * EB;
    const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v) + (int64_t)hk * p.v_head_stride * EB;
    const int32_t* btab = PAGED ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;
    // the same table through the constant address space: a uniform load from it is a scalar load (s_load_dword, waited for
    // with lgkmcnt at its use); a plain load was a vector load whose `s_waitcnt vmcnt(0)` drained the tiles in flight
    typedef const int32_t __attribute__((address_space(4))) * const_i32_ptr;
    const const_i32_ptr btab_c = (const_i32_ptr)(uintptr_t)btab;
    // register sets = tiles in flight per wave.  Round 1 gave fp8 caches four (an fp8 tile is half the bytes; with two the
    // kernel of that time ran at 3.0 TB/s); since the one-row-per-head shapes moved to the token-major kernel what runs
    // here (GQA groups of 8+, multi-token blocks) is bound by its ~540 instructions per tile, not by the bytes in flight:
    // two sets are 8-15 % faster than four, six change nothing, eight spill (tools/decode_splits_sweep.py, round 3).
#ifndef FA_DEC_NS8
#define FA_DEC_NS8 2
#endif
#ifndef FA_DEC_NS16
#define FA_DEC_NS16 2
#endif
#ifndef FA_DEC_NS8W8
#define FA_DEC_NS8W8 2                 // fp8 cache, eight waves
#endif
#ifndef FA_DEC_NS256
#define FA_DEC_NS256 1                 // D = 256: two sets spill (172-280 bytes of scratch per lane)
#endif
    // (D = 256, eight waves: register budget - one set; an fp8 tile is half the registers AND half the bytes in flight: two
    //  at D = 64 - at D = 128 two sets spill 107 registers, and the fp8-operand form stages by LDS-DMA instead)
    constexpr int NS = (KV8 && NW == 8 && D <= 64) ? FA_DEC_NS8W8 : ((D > 128 || NW == 8) ? FA_DEC_NS256 : (KV8 ? FA_DEC_NS8 : FA_DEC_NS16));
    constexpr int STAGES = DecSmem<D, NW>::STAGES;
    u32x4 kS[NS][CH], vS[NS][CH];
    // loop-invariant per-lane byte offset inside a tile (row * row_stride + 16-byte column) of the lane's FIRST chunk; chunk i lies
    // i * RPS rows further down (64 lanes cover 64 / CPR whole rows per load step), which is a UNIFORM distance: it goes into the
    // scalar base of load i.  (As sixteen per-lane offsets - round 1's form - the 16-bit eight-wave kernel parked three of them in
    // scratch and reloaded them INSIDE the tile loop, each reload followed by an `s_waitcnt vmcnt(0)` that drained the K / V loads
    // issued before it: profiles/r06_decode.txt section 3.)
    static_assert(64 % CPR == 0, "a load step covers whole rows");
    bool cok[CH];                                           // NARROW: is this lane's chunk inside the row?
    const int vchunks = vcols * EB / 16;
    const int cc0 = lane % CPR;
#pragma unroll
    for (int i = 0; i < CH; ++i) cok[i] = !NARROW || cc0 < vchunks;
    const int ccl0 = cok[0] ? cc0 : 0;                      // (a chunk past the row reads chunk 0 and is zeroed)
    const uint32_t k_voff0 = (uint32_t)((lane / CPR) * p.k_row_stride * EB + ccl0 * 16);
    const uint32_t v_voff0 = (uint32_t)((lane / CPR) * p.v_row_stride * EB + ccl0 * 16);
    const u32x4 zero4 = {0, 0, 0, 0};
    // a 16-row half of a tile lies inside one page when the left pad and the page size are multiples of 16
    // (pages of 16 tokens - vLLM's default block - hold half a 32-key tile: the tile's two 16-row halves take their own
    //  page; a lane's chunk i lies in one half as a whole because 64 / CPR rows per load step divide 16)
    const bool tiles_aligned = !PAGED || (((lp & 15) == 0) && (p.page_block_size % 16) == 0);
    constexpr int RPS = 64 / CPR;                           // rows per load step
    static_assert(16 % RPS == 0 || RPS > 16, "a load step stays inside a 16-row half");
    // a full, page-aligned tile: ONE scalar base per tile, no predication, no branch (so that the
    // compiler can count the loads in flight: see the steady-state loop below)
    auto load_fast = [&](int tile, u32x4 (&kreg)[CH], u32x4 (&vreg)[CH]) {
        const int pos0 = lp + tile * BN;
        int64_t ko, vo;
        int64_t ko2 = 0, vo2 = 0;                              // second 16-row half of a 32-key tile (its own page when pages are 16 tokens)
        if (PAGED) {
            const int pg = da.page_shift >= 0 ? (pos0 >> da.page_shift) : pos0 / p.page_block_size;
            const int pr = pos0 - pg * p.page_block_size;
            // the page id is the same in every lane: in an SGPR the tile's base is scalar arithmetic and the eight loads
            // take the base from SGPRs (as a vector value it cost six 32-bit multiplies and a 64-bit add per load and tile)
            const int64_t phys = FA_DEC_UNIFORM(btab_c[pg]);
            ko = phys * p.k_batch_stride + (int64_t)pr * p.k_row_stride;
            vo = phys * p.v_batch_stride + (int64_t)pr * p.v_row_stride;
            if (BN > 16) {
                const int pos1 = pos0 + 16;
                const int pg1 = da.page_shift >= 0 ? (pos1 >> da.page_shift) : pos1 / p.page_block_size;
                const int64_t phys1 = FA_DEC_UNIFORM(btab_c[pg1]);
                // (base of the half minus its 16 rows: the lane offsets count rows from the tile's first row)
                ko2 = phys1 * p.k_batch_stride + (int64_t)(pos1 - pg1 * p.page_block_size - 16) * p.k_row_stride;
                vo2 = phys1 * p.v_batch_stride + (int64_t)(pos1 - pg1 * p.page_block_size - 16) * p.v_row_stride;
            }
        } else {
            ko = (int64_t)cb * p.k_batch_stride + (int64_t)pos0 * p.k_row_stride;
            vo = (int64_t)cb * p.v_batch_stride + (int64_t)pos0 * p.v_row_stride;
        }
        const uint8_t* kb = kbase + ko * EB;
        const uint8_t* vb = vbase + vo * EB;
        const uint8_t* kb2 = PAGED && BN > 16 ? kbase + ko2 * EB : kb;
        const uint8_t* vb2 = PAGED && BN > 16 ? vbase + vo2 * EB : vb;
        // (the empty asm keeps the zero-extension of the 32-bit lane offset next to the load: hoisted out of the loop as a
        //  64-bit pair it cost a v_lshl_add_u64 per load and 16 registers; here the load takes SGPR base + 32-bit VGPR offset)
        const int64_t kstep = (int64_t)(64 / CPR) * p.k_row_stride * EB, vstep = (int64_t)(64 / CPR) * p.v_row_stride * EB;
#pragma unroll
        for (int i = 0; i < CH; ++i) { uint32_t o = k_voff0; FA_DEC_PIN(o); kreg[i] = *reinterpret_cast<const u32x4*>((i * RPS >= 16 ? kb2 : kb) + i * kstep + o); }
#pragma unroll
        for (int i = 0; i < CH; ++i) { uint32_t o = v_voff0; FA_DEC_PIN(o); vreg[i] = *reinterpret_cast<const u32x4*>((i * RPS >= 16 ? vb2 : vb) + i * vstep + o); }
    };
    (void)cok; (void)zero4;
    auto load_tile = [&](int tile, u32x4 (&kreg)[CH], u32x4 (&vreg)[CH]) {
        const int j0 = tile * BN;
        if (tiles_aligned && j0 + BN <= seqlen_k) { load_fast(tile, kreg, vreg); return; }
#pragma unroll 1
        for (int i = 0; i < CH; ++i) {
            const int cidx = lane + 64 * i;
            const int row = cidx / CPR, cc = cidx % CPR;
            const int j = j0 + row;
            u32x4 z = {0, 0, 0, 0};
            u32x4 kx = z, vx = z;
            if (j < seqlen_k && (!NARROW || cc < vchunks)) {
                const int pos = lp + j;
                int64_t ko, vo;
                if (PAGED) {
                    const int pg = da.page_shift >= 0 ? (pos >> da.page_shift) : pos / p.page_block_size;
                    const int pr = pos - pg * p.page_block_size;
                    const int64_t phys = btab[pg];
                    ko = phys * p.k_batch_stride + (int64_t)pr * p.k_row_stride;
                    vo = phys * p.v_batch_stride + (int64_t)pr * p.v_row_stride;
                } else {
                    ko = (int64_t)cb * p.k_batch_stride + (int64_t)pos * p.k_row_stride;
                    vo = (int64_t)cb * p.v_batch_stride + (int64_t)pos * p.v_row_stride;
                }
                kx = *reinterpret_cast<const u32x4*>(kbase + ko * EB + cc * 16);
                vx = *reinterpret_cast<const u32x4*>(vbase + vo * EB + cc * 16);
            }
            // runtime-indexed store into the register arrays would spill: select with a static unroll
#pragma unroll
            for (int i2 = 0; i2 < CH; ++i2) if (i2 == i) { kreg[i2] = kx; vreg[i2] = vx; }
        }
    };
    auto store_tile = [&](int stage, const u32x4 (&kreg)[CH], const u32x4 (&vreg)[CH]) {
        char* ks = wsm + stage * (F8M ? TILE : 2 * TILE);
        char* vs = ks + (F8M ? TILE / 2 : TILE);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int cidx = lane + 64 * i;
            const int row = cidx / CPR, cc = cidx % CPR;
            if constexpr (F8M) {
                // the cache bytes as they are: 128-byte rows, 16-byte slots XOR-ed with row bits so that the 8-byte row reads
                // of S (32 lanes x 32 rows) are 2-way and the transposing reads of P V (8 rows x 16 B per 16 lanes) conflict-free
                lds_write_b128(ks + row * D + ((cc ^ ((row >> 1) & 7)) << 4), kreg[i]);
                lds_write_b128(vs + row * D + ((cc ^ ((((row >> 3) & 1) << 2) | (row & 3))) << 4), vreg[i]);
            } else if (KV8) {
                u32x4 l0, h0, l1, h1;
                fp8x16_to_16bit<T>(kreg[i], l0, h0);
                fp8x16_to_16bit<T>(vreg[i], l1, h1);
                lds_write_b128(ks + swz_row_off<D>(row, cc * 32), l0);
                lds_write_b128(ks + swz_row_off<D>(row, cc * 32 + 16), h0);
                lds_write_b128(vs + swzt_row_off<D>(row, cc * 32), l1);
                lds_write_b128(vs + swzt_row_off<D>(row, cc * 32 + 16), h1);
            } else {
                lds_write_b128(ks + swz_row_off<D>(row, cc * 16), (NARROW && !cok[i]) ? zero4 : kreg[i]);
                lds_write_b128(vs + swzt_row_off<D>(row, cc * 16), (NARROW && !cok[i]) ? zero4 : vreg[i]);
            }
        }
    };

    f32x16 oacc[DBLKS];
#pragma unroll
    for (int d = 0; d < DBLKS; ++d)
#pragma unroll
        for (int rr2 = 0; rr2 < 16; ++rr2) oacc[d][rr2] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // score modifiers (ALiBi slopes, softcap; reference order: include/mat_mul.h:113-116 - bias, then cap): the scores are
    // brought to log2 units element by element and the exponent's multiplier becomes 1.  A wave-uniform branch: this
    // kernel waits for HBM, not for the VALU.
    const bool bias = da.bias != 0;
    const float c = bias ? 1.0f : a.scale_log2e * (KV8 ? p.k_descale : 1.0f) * q_unscale;
    const float sc_lin = p.softmax_scale * (KV8 ? p.k_descale : 1.0f) * q_unscale;
    const float slope = (bias && p.alibi_slopes && row_ok) ? p.alibi_slopes[(int64_t)b * p.alibi_batch_stride + h] : 0.f;
    const float cap = p.softcap, rcap = p.softcap > 0.f ? 1.0f / p.softcap : 0.f;
    const int v_rr = (lane & 15) >> 2;
    const int v_cb = (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);

    auto compute_tile = [&](int tile, int stage) {
        // (F8M: an fp8 tile is half the bytes - the wave's region holds TWO stages of K | V images of TILE / 2 each)
        const char* ks = wsm + stage * (F8M ? TILE : 2 * TILE);
        const char* vs = ks + (F8M ? TILE / 2 : TILE);
        const int n0 = tile * BN;
        // S^T[key][row] = K Q^T
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
        if constexpr (F8M) {
            // A: key row l31, bytes 16 ksx + 8 g .. + 7 as stored; B: the query row's terms.  All fragment reads of the tile are
            // issued before the first MFMA (hipcc otherwise runs read -> wait -> MFMA through one temporary: a full LDS round
            // trip per k-step, and with two waves per SIMD nothing covers it)
            u32x2 kf[KSTEPS];
            u32x4 qhl[KSTEPS];
#pragma unroll
            for (int ksx = 0; ksx < KSTEPS; ++ksx) {
                kf[ksx] = *reinterpret_cast<const u32x2*>(ks + l31 * D + ((ksx ^ ((l31 >> 1) & 7)) << 4) + 8 * g);
                if constexpr (Q_LDS) qhl[ksx] = lds_read_b128(smem + DecSmem<D, NW>::QOFF + (ksx * 64 + lane) * 16);
                else qhl[ksx] = qf[ksx];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ksx = 0; ksx < KSTEPS; ++ksx) {
                const long ka = __builtin_bit_cast(long, kf[ksx]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(ka, __builtin_bit_cast(long, u32x2{qhl[ksx][0], qhl[ksx][1]}), s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(ka, __builtin_bit_cast(long, u32x2{qhl[ksx][2], qhl[ksx][3]}), s, 0, 0, 0);
                if constexpr (NTQ > 2) {
                    const u32x2 q3 = *reinterpret_cast<const u32x2*>(smem + DecSmem<D, NW>::QOFF + DecSmem<D, NW>::QBYTES + (ksx * 64 + lane) * 8);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(ka, __builtin_bit_cast(long, q3), s, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int ksx = 0; ksx < KSTEPS; ++ksx) {
            const u32x4 kf = lds_read_b128(ks + swz_row_off<D>(l31, 32 * ksx + 16 * g));
            if constexpr (Q_LDS) s = E::mfma(kf, lds_read_b128(smem + DecSmem<D, NW>::QOFF + (ksx * 64 + lane) * 16), s);
            else s = E::mfma(kf, qf[ksx], s);
        }
        }
        if (bias) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = n0 + (i & 3) + 8 * (i >> 2) + 4 * g;
                float x = fmaf(-slope, fabsf((float)(t_row + off - j)), s[i] * sc_lin);
                if (cap > 0.f) x = cap * fast_tanh(x * rcap);
                s[i] = x * kLog2e;
            }
        }
        if (n0 < w_lo_max || n0 + BN - 1 > w_hi_min || !any_row) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = n0 + (i & 3) + 8 * (i >> 2) + 4 * g;
                if (j < lo || j > hi) s[i] = -INFINITY;
            }
        }
        if (BN == 16) {                                     // (16-key tiles: key rows 16..31 are not there)
#pragma unroll
            for (int i = 8; i < 16; ++i) s[i] = -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
        mx = xhalf_max(mx) * c;
        // deferred rescale (as in fa_fwd_kernel): the running max is only raised - and the accumulators only
        // multiplied - when some row of the wave exceeds it by more than 2^DEC_RESCALE_THR
        // (NaN-safe: -inf - -inf compares false -> takes the rescale path)
        if (!__all(mx - m_run <= RESCALE_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float m_nu = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - m_nu);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < DBLKS; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
        }
        const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = fast_exp2(fmaf(s[i], c, -m_use)); psum += s[i]; }
        l_run += psum;
        if constexpr (F8M) {
            // transposing read: the 16 lanes of a group bring 8 key rows x 16 d-bytes (source lane s: key j = s >> 1 of the k-step's
            // eight keys in C-layout order, 8-byte half s & 1) and lane 8 h + c receives, for d = 16 (G & 1) + 8 h + c, the bytes
            // of keys j = 0 .. 7 - exactly the A operand of V^T (row d, contraction = key) in the key order P's registers have
            const int sl = lane & 15, gg = lane >> 4;
            const int jj = sl >> 1;
            const int key_l = (jj & 3) + 8 * (jj >> 2) + 4 * (gg >> 1);                 // + 16 t2
            const int dby = 16 * (gg & 1) + 8 * (sl & 1);                               // + 32 d
            // (asm form + counted waits, fa_common.h: the builtin gets an s_waitcnt vmcnt(0) in front of it - the DMA of the NEXT
            //  tile, issued a moment ago - which emptied the two-stage pipeline once per tile: 5.6 TB/s where whole rows stream at 7)
            u32x2 vt[BN / 16][DBLKS];
#pragma unroll
            for (int t2 = 0; t2 < BN / 16; ++t2) {
                const int key = 16 * t2 + key_l;
                const int fv = (((key >> 3) & 1) << 2) | (key & 3);
#pragma unroll
                for (int d = 0; d < DBLKS; ++d) {
                    const int slot = (2 * d + (gg & 1)) ^ fv;
                    vt[t2][d] = lds_read_tr8_nw((const lds_char*)(vs + key * D + (slot << 4) + (dby & 8)), 0);
                }
            }
            u32x2 pt[BN / 16][NTP];
#pragma unroll
            for (int t2 = 0; t2 < BN / 16; ++t2) {
                float p8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) p8[j] = s[8 * t2 + j] * P_UP;
                fp8_terms8<NTP>(p8, pt[t2]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t2 = 0; t2 < BN / 16; ++t2)
#pragma unroll
                for (int d = 0; d < DBLKS; ++d) {
                    lds_tr_wait(vt[t2][d], (BN / 16) * DBLKS - 1 - (t2 * DBLKS + d));
                    const long va = __builtin_bit_cast(long, vt[t2][d]);
#pragma unroll
                    for (int t = 0; t < NTP; ++t) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(va, __builtin_bit_cast(long, pt[t2][t]), oacc[d], 0, 0, 0);
                }
        } else
#pragma unroll
        for (int t2 = 0; t2 < BN / 16; ++t2) {
            u32x4 pf;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) pf[w2] = E::pack2(s[8 * t2 + 2 * w2], s[8 * t2 + 2 * w2 + 1]);
            const int row_a = 16 * t2 + 4 * g + v_rr;
#pragma unroll
            for (int d = 0; d < DBLKS; ++d) {
                const u32x2 v0 = lds_read_tr16(vs + swzt_row_off<D>(row_a, d * 64 + v_cb));
                const u32x2 v1 = lds_read_tr16(vs + swzt_row_off<D>(row_a + 8, d * 64 + v_cb));
                u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
                oacc[d] = E::mfma(vf, pf, oacc[d]);
            }
        }
    };

    constexpr int TS = HPW ? 1 : NW;                        // tile stride of a wave (HPW: every wave walks all tiles of the split)
    const int t0 = s_lo + (HPW ? 0 : wave);
    const int n_my = t0 < s_hi ? (s_hi - t0 + TS - 1) / TS : 0;
    if constexpr (F8M) {
        // The cache bytes go HBM -> LDS by LDS-DMA (buffer_load ... lds: no staging registers), two stages per wave: tile s + 1
        // lands while tile s is computed, behind a COUNTED vmcnt (the 8 pieces of the younger tile stay in flight).  The
        // destination is lane-linear (lane l of piece i -> row 8 i + l / 8, 16-byte slot l % 8), so the images' slot XORs
        // are applied to the source column.  Ragged / unaligned tiles (a sequence's last one) take the register path.
        uint32_t k_dma[CH], v_dma[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int row = (lane + 64 * i) / CPR, slot = lane % CPR;
            k_dma[i] = (uint32_t)(row * p.k_row_stride + ((slot ^ ((row >> 1) & 7)) << 4));
            v_dma[i] = (uint32_t)(row * p.v_row_stride + ((slot ^ ((((row >> 3) & 1) << 2) | (row & 3))) << 4));
        }
        const int n_full_tiles = seqlen_k / BN;                       // tiles that lie completely inside the sequence
        auto issue = [&](int tile, int stage) -> bool {               // true: by DMA (8 pieces in flight), false: done synchronously
            char* kdst = wsm + stage * (F8M ? TILE : 2 * TILE);
            char* vdst = kdst + (F8M ? TILE / 2 : TILE);
            if (!(tiles_aligned && tile < n_full_tiles)) {
                load_tile(tile, kS[0], vS[0]);
                store_tile(stage, kS[0], vS[0]);
                return false;
            }
            const int pos0 = lp + tile * BN;
            int64_t ko, vo, ko2 = 0, vo2 = 0;
            if (PAGED) {
                const int pg = da.page_shift >= 0 ? (pos0 >> da.page_shift) : pos0 / p.page_block_size;
                const int pr = pos0 - pg * p.page_block_size;
                const int64_t phys = btab_c[pg];
                ko = phys * p.k_batch_stride + (int64_t)pr * p.k_row_stride;
                vo = phys * p.v_batch_stride + (int64_t)pr * p.v_row_stride;
                if (BN > 16) {                                        // (the tile's second 16-row half may lie in the next page)
                    const int pos1 = pos0 + 16;
                    const int pg1 = da.page_shift >= 0 ? (pos1 >> da.page_shift) : pos1 / p.page_block_size;
                    const int64_t phys1 = btab_c[pg1];
                    ko2 = phys1 * p.k_batch_stride + (int64_t)(pos1 - pg1 * p.page_block_size - 16) * p.k_row_stride;
                    vo2 = phys1 * p.v_batch_stride + (int64_t)(pos1 - pg1 * p.page_block_size - 16) * p.v_row_stride;
                }
            } else {
                ko = (int64_t)cb * p.k_batch_stride + (int64_t)pos0 * p.k_row_stride;
                vo = (int64_t)cb * p.v_batch_stride + (int64_t)pos0 * p.v_row_stride;
            }
            auto rsrc_of = [](const uint8_t* ptr) {
                const uint64_t b64 = reinterpret_cast<uint64_t>(ptr);
                const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b64);
                const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b64 >> 32));
                return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi32 << 32) | lo32), 0, 0x7fffffff, 0x00020000);
            };
            const __amdgpu_buffer_rsrc_t kr = rsrc_of(kbase + ko * EB), vr = rsrc_of(vbase + vo * EB);
            const __amdgpu_buffer_rsrc_t kr2 = (PAGED && BN > 16) ? rsrc_of(kbase + ko2 * EB) : kr, vr2 = (PAGED && BN > 16) ? rsrc_of(vbase + vo2 * EB) : vr;
#pragma unroll
            for (int i = 0; i < CH; ++i) buf_load_lds_b128(i * RPS >= 16 ? kr2 : kr, kdst + i * 1024, k_dma[i], 0);
#pragma unroll
            for (int i = 0; i < CH; ++i) buf_load_lds_b128(i * RPS >= 16 ? vr2 : vr, vdst + i * 1024, v_dma[i], 0);
            return true;
        };
        bool cur_dma = n_my > 0 ? issue(t0, 0) : false;
        (void)cur_dma;
        for (int s1 = 0; s1 < n_my; ++s1) {
            bool nxt_dma = false;
            if (s1 + 1 < n_my) nxt_dma = issue(t0 + TS * (s1 + 1), (s1 + 1) & 1);
            if (nxt_dma) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // (2 x CH pieces of the younger tile may still fly)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            compute_tile(t0 + TS * s1, s1 & 1);
        }
    } else
    if constexpr (STAGES == 1) {
        // eight waves: one LDS stage per wave.  Step s: the set that holds tile s goes to LDS (waits for its loads), the
        // set is re-loaded with tile s + NS, tile s is computed - NS tiles in flight while it runs, and the SIMD's other
        // wave fills the waits
        const int n_full1 = t0 < s_hi ? ((seqlen_k / BN < s_hi ? seqlen_k / BN : s_hi) - t0 + TS - 1) / TS : 0;
        int s1 = 0;
        if (tiles_aligned && NS < n_full1) {
#pragma unroll
            for (int j = 0; j < NS; ++j) load_fast(t0 + TS * j, kS[j], vS[j]);
            for (; s1 + 2 * NS <= n_full1; s1 += NS) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    store_tile(0, kS[j], vS[j]);
                    load_fast(t0 + TS * (s1 + j + NS), kS[j], vS[j]);
                    compute_tile(t0 + TS * (s1 + j), 0);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NS; ++j)
                if (j < n_my) load_tile(t0 + TS * j, kS[j], vS[j]);
        }
        for (; s1 < n_my; s1 += NS) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int ss = s1 + j;
                if (ss < n_my) {
                    store_tile(0, kS[j], vS[j]);
                    if (ss + NS < n_my) load_tile(t0 + TS * (ss + NS), kS[j], vS[j]);
                    compute_tile(t0 + TS * ss, 0);
                }
            }
        }
    } else {
    // LDS stage of step s0 + j: a compile-time constant for even NS (s0 is a multiple of NS)
    auto stage_of = [](int s0_, int j_) { return (NS % 2 == 0) ? (j_ & 1) : ((s0_ + j_) & 1); };
    // pipeline over this wave's tiles t0 + 4 s: LDS stage s & 1 holds tile s while the register sets
    // hold tiles s+1 .. s+NS-1 (landed / landing) and the set just stored is re-loaded with s+1+NS.
    // my tiles that lie completely inside [0, seqlen_k): s < n_full
    const int n_full = t0 < s_hi ? ((seqlen_k / BN < s_hi ? seqlen_k / BN : s_hi) - t0 + TS - 1) / TS : 0;
    int s0 = 0;
    if (tiles_aligned && 2 * NS < n_full) {
        // steady state: every store / load is unconditional, so the vmcnt waits in front of the
        // stores are exact counts (NS - 1 tiles stay in flight) instead of conservative drains
#pragma unroll
        for (int j = 0; j < NS; ++j) load_fast(t0 + TS * j, kS[j], vS[j]);
        store_tile(0, kS[0], vS[0]);
        load_fast(t0 + TS * NS, kS[0], vS[0]);
        for (; s0 + 2 * NS < n_full; s0 += NS) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int nxt = (j + 1) % NS;
                store_tile(stage_of(s0, j + 1), kS[nxt], vS[nxt]);
                load_fast(t0 + TS * (s0 + j + 1 + NS), kS[nxt], vS[nxt]);
                compute_tile(t0 + TS * (s0 + j), stage_of(s0, j));
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (j < n_my) load_tile(t0 + TS * j, kS[j], vS[j]);
        if (n_my > 0) {
            store_tile(0, kS[0], vS[0]);
            if (NS < n_my) load_tile(t0 + TS * NS, kS[0], vS[0]);
        }
    }
    // remaining tiles (and short sequences): same schedule with every step guarded
    for (; s0 < n_my; s0 += NS) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int ss = s0 + j;
            if (ss < n_my) {
                const int nxt = (j + 1) % NS;              // compile-time after unrolling
                if (ss + 1 < n_my) store_tile(stage_of(s0, j + 1), kS[nxt], vS[nxt]);
                if (ss + 1 + NS < n_my) load_tile(t0 + TS * (ss + 1 + NS), kS[nxt], vS[nxt]);
                compute_tile(t0 + TS * ss, stage_of(s0, j));
            }
        }
    }

    }

    if constexpr (HPW) {
        // ---- every wave stores its own head: row l31, 8-byte pieces of the C layout (d = 32 dblk + 8 rq + 4 g .. + 3) ----
        const float l_w = xhalf_sum(l_run);
        const float inv = l_w > 0.f ? (KV8 ? p.v_descale : 1.0f) / (l_w * P_UP) : 0.f;
        const float lse = l_w > 0.f ? (m_run + fast_log2(l_w)) * kLn2 : -INFINITY;
        if (row_ok) {
            if (da.n_splits == 1) {
                uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + (int64_t)b * p.o_batch_stride + (int64_t)(q_row0 + t_row) * p.o_row_stride +
                               (int64_t)h * p.o_head_stride;
#pragma unroll
                for (int d = 0; d < DBLKS; ++d)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        u32x2 o2;
                        o2[0] = E::pack2(oacc[d][4 * rq + 0] * inv, oacc[d][4 * rq + 1] * inv);
                        o2[1] = E::pack2(oacc[d][4 * rq + 2] * inv, oacc[d][4 * rq + 3] * inv);
                        *reinterpret_cast<u32x2*>(op + d * 32 + 8 * rq + 4 * g) = o2;
                    }
                if (g == 0) p.lse[(int64_t)b * p.lse_batch_stride + (int64_t)h * p.lse_head_stride + q_row0 + t_row] = lse;
            } else {
                const int64_t prow = (((int64_t)split * p.batch + b) * p.nheads_q + h) * p.seqlen_q + t_row;
                float* op = da.o_partial + prow * D;
#pragma unroll
                for (int d = 0; d < DBLKS; ++d)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const f32x4 o4 = {oacc[d][4 * rq] * inv, oacc[d][4 * rq + 1] * inv, oacc[d][4 * rq + 2] * inv, oacc[d][4 * rq + 3] * inv};
                        *reinterpret_cast<f32x4*>(op + d * 32 + 8 * rq + 4 * g) = o4;
                    }
                if (g == 0) da.lse_partial[prow] = lse;
            }
        }
        return;
    }
    // ---- merge the 4 waves through LDS ----
    __syncthreads();                                        // everyone is done with its tiles
    const float l_w = xhalf_sum(l_run);
    float* red_m = reinterpret_cast<float*>(smem);                       // [4][32]
    float* red_l = red_m + NW * 32;                                      // [NW][32]
    float* red_o = red_l + NW * 32;                                      // [NW][32 rows][D]
    if (g == 0) { red_m[wave * 32 + l31] = m_run; red_l[wave * 32 + l31] = l_w; }
#pragma unroll
    for (int d = 0; d < DBLKS; ++d)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 o4 = {oacc[d][4 * rq], oacc[d][4 * rq + 1], oacc[d][4 * rq + 2], oacc[d][4 * rq + 3]};
            *reinterpret_cast<f32x4*>(red_o + ((wave * 32 + l31) * D + d * 32 + 8 * rq + 4 * g)) = o4;
        }
    __syncthreads();
    // thread -> (row = tid / 8, 16-column slice)
    {
        constexpr int TPR = 2 * NW;                         // threads per row (8 or 16)
        constexpr int CPT = D / TPR;                        // columns per thread
        const int row = tid / TPR;                          // 0..31
        const int cs = (tid % TPR) * CPT;
        const int t3 = (rbase + row) / G, gq3 = (rbase + row) - t3 * G;
        if (rbase + row < R) {
            float mw[NW], lw[NW];
            float m_all = -INFINITY;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) { mw[w2] = red_m[w2 * 32 + row]; lw[w2] = red_l[w2 * 32 + row]; m_all = fmaxf(m_all, mw[w2]); }
            const float m_s = (m_all == -INFINITY) ? 0.f : m_all;
            float l_all = 0.f, sc[NW];
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) { sc[w2] = fast_exp2(mw[w2] - m_s); l_all = fmaf(lw[w2], sc[w2], l_all); }
            const float inv = l_all > 0.f ? (KV8 ? p.v_descale : 1.0f) / (l_all * P_UP) : 0.f;
            const float lse = l_all > 0.f ? (m_all + fast_log2(l_all)) * kLn2 : -INFINITY;
            const int hq = hk * G + gq3;
            if (da.n_splits == 1) {
                uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + (int64_t)b * p.o_batch_stride + (int64_t)(q_row0 + t3) * p.o_row_stride +
                               (int64_t)hq * p.o_head_stride + cs;
#pragma unroll
                for (int x = 0; x < CPT; x += 2) {
                    if (NARROW && cs + x >= vcols) break;
                    float v0 = 0.f, v1 = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < NW; ++w2) {
                        v0 = fmaf(red_o[(w2 * 32 + row) * D + cs + x], sc[w2], v0);
                        v1 = fmaf(red_o[(w2 * 32 + row) * D + cs + x + 1], sc[w2], v1);
                    }
                    *reinterpret_cast<uint32_t*>(op + x) = E::pack2(v0 * inv, v1 * inv);
                }
                if (tid % TPR == 0)
                    p.lse[(int64_t)b * p.lse_batch_stride + (int64_t)hq * p.lse_head_stride + q_row0 + t3] = lse;
            } else {
                const int64_t prow = (((int64_t)split * p.batch + b) * p.nheads_q + hq) * p.seqlen_q + t3;
                float* op = da.o_partial + prow * D + cs;
#pragma unroll
                for (int x = 0; x < CPT; ++x) {
                    float v0 = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < NW; ++w2) v0 = fmaf(red_o[(w2 * 32 + row) * D + cs + x], sc[w2], v0);
                    op[x] = v0 * inv;
                }
                if (tid % TPR == 0) da.lse_partial[prow] = lse;
            }
        }
    }
}

// out[row] = sum_s w_s O_s,  w_s = exp(lse_s - LSE),  LSE = log sum_s exp(lse_s)
template <typename T>
__global__ void __launch_bounds__(256) decode_combine_kernel(const DecArgs da) {
    using E = Elem<T>;
    const fa_params& p = da.a.p;
    const int D = p.head_dim;
    const int cpr = D / 8;
    const int64_t rows = (int64_t)p.batch * p.nheads_q * p.seqlen_q;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = idx / cpr;
    const int cc = idx % cpr;
    if (row >= rows) return;
    float m = -INFINITY;
    for (int s = 0; s < da.n_splits; ++s) m = fmaxf(m, da.lse_partial[(int64_t)s * rows + row]);
    const float m_s = (m == -INFINITY) ? 0.f : m;
    float den = 0.f;
    for (int s = 0; s < da.n_splits; ++s) den += __expf(da.lse_partial[(int64_t)s * rows + row] - m_s);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < da.n_splits; ++s) {
        const float wgt = den > 0.f ? __expf(da.lse_partial[(int64_t)s * rows + row] - m_s) / den : 0.f;
        const float* op = da.o_partial + ((int64_t)s * rows + row) * D + cc * 8;
#pragma unroll
        for (int x = 0; x < 8; ++x) acc[x] = fmaf(op[x], wgt, acc[x]);
    }
    int t = row % p.seqlen_q;
    const int64_t bh = row / p.seqlen_q;
    const int hq = bh % p.nheads_q;
    const int64_t b = bh / p.nheads_q;
    if (da.cu_q) {                                          // varlen-q mode: rows of other classes / past the sequence
        const int q0 = da.cu_q[b], ql = da.cu_q[b + 1] - q0;
        if (ql < 1 || ql > p.seqlen_q || t >= ql) return;
        t += q0;
    }
    uint16_t* out = reinterpret_cast<uint16_t*>(p.o) + b * p.o_batch_stride + (int64_t)t * p.o_row_stride +
                    (int64_t)hq * p.o_head_stride + cc * 8;
    u32x4 o4;
#pragma unroll
    for (int x = 0; x < 4; ++x) o4[x] = E::pack2(acc[2 * x], acc[2 * x + 1]);
    if (p.head_dim_v == 0 || cc * 8 < p.head_dim_v) *reinterpret_cast<u32x4*>(out) = o4;     // (narrow rows: valid columns only)
    if (cc == 0) p.lse[b * p.lse_batch_stride + (int64_t)hq * p.lse_head_stride + t] = den > 0.f ? m + __logf(den) : -INFINITY;
}

// The same merge for MANY partials (small batches split the key range until the grid fills the chip: up to 256 partial
// rows per output row).  The kernel above walks the partials one after the other in every thread - 126 dependent trips
// to L2 = 80 us for a 15 us decode (rocprofv3, B 1, H 32/8, 4k context).  Here one workgroup owns an output row: the
// weights are computed once (one partial per thread, two block reductions), then D / 4 column groups x 256 / (D / 4)
// partial lanes stream the partial rows with independent 16-byte loads and meet in LDS.
template <typename T>
__global__ void __launch_bounds__(256) decode_combine_wide_kernel(const DecArgs da) {
    using E = Elem<T>;
    const fa_params& p = da.a.p;
    const int D = p.head_dim;
    const int ncg = D / 4;                                  // float4 column groups: 32 (D 128) or 16 (D 64)
    const int npl = 256 / ncg;                              // partial lanes: 8 or 16
    const int64_t rows = (int64_t)p.batch * p.nheads_q * p.seqlen_q;
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = da.n_splits;
    int t_out = row % p.seqlen_q;
    const int64_t bh_ = row / p.seqlen_q;
    const int hq_ = bh_ % p.nheads_q;
    const int64_t b_ = bh_ / p.nheads_q;
    if (da.cu_q) {                                          // varlen-q mode: rows of other classes / past the sequence
        const int q0 = da.cu_q[b_], ql = da.cu_q[b_ + 1] - q0;
        if (ql < 1 || ql > p.seqlen_q || t_out >= ql) return;          // (uniform per workgroup: before the barriers)
        t_out += q0;
    }
    __shared__ float s_w[256];
    __shared__ float s_red[8];
    __shared__ f32x4 s_acc[256];
    // ---- weights: w_s = exp(lse_s - max) / sum ----
    float m = -INFINITY;
    for (int s = tid; s < n; s += 256) m = fmaxf(m, da.lse_partial[(int64_t)s * rows + row]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    const float m_s = (m == -INFINITY) ? 0.f : m;
    float den = 0.f;
    for (int s = tid; s < n; s += 256) den += __expf(da.lse_partial[(int64_t)s * rows + row] - m_s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
    if (lane == 0) s_red[4 + wave] = den;
    __syncthreads();
    den = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    const float rden = den > 0.f ? 1.0f / den : 0.f;
    const int cg = tid % ncg, pl = tid / ncg;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < n; s0 += 256) {                   // (n <= 256 in practice: one trip)
        __syncthreads();
        if (s0 + tid < n) s_w[tid] = __expf(da.lse_partial[(int64_t)(s0 + tid) * rows + row] - m_s) * rden;
        __syncthreads();
        const int cnt = n - s0 < 256 ? n - s0 : 256;
        const float* base = da.o_partial + ((int64_t)s0 * rows + row) * D + cg * 4;
        const int64_t stride = rows * D;
        int s = pl;
        for (; s + 3 * npl < cnt; s += 4 * npl) {           // four independent loads in flight per thread
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(base + (int64_t)s * stride);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(base + (int64_t)(s + npl) * stride);
            const f32x4 x2 = *reinterpret_cast<const f32x4*>(base + (int64_t)(s + 2 * npl) * stride);
            const f32x4 x3 = *reinterpret_cast<const f32x4*>(base + (int64_t)(s + 3 * npl) * stride);
            const float w0 = s_w[s], w1 = s_w[s + npl], w2 = s_w[s + 2 * npl], w3 = s_w[s + 3 * npl];
#pragma unroll
            for (int x = 0; x < 4; ++x) acc[x] = fmaf(x0[x], w0, fmaf(x1[x], w1, fmaf(x2[x], w2, fmaf(x3[x], w3, acc[x]))));
        }
        for (; s < cnt; s += npl) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(base + (int64_t)s * stride);
            const float w0 = s_w[s];
#pragma unroll
            for (int x = 0; x < 4; ++x) acc[x] = fmaf(x0[x], w0, acc[x]);
        }
    }
    s_acc[tid] = acc;
    __syncthreads();
    if (tid < ncg) {
        for (int q = 1; q < npl; ++q) {
            const f32x4 y = s_acc[q * ncg + tid];
#pragma unroll
            for (int x = 0; x < 4; ++x) acc[x] += y[x];
        }
        const int t = t_out, hq = hq_;
        const int64_t b = b_;
        uint16_t* out = reinterpret_cast<uint16_t*>(p.o) + b * p.o_batch_stride + (int64_t)t * p.o_row_stride +
                        (int64_t)hq * p.o_head_stride + tid * 4;
        u32x2 o2 = {E::pack2(acc[0], acc[1]), E::pack2(acc[2], acc[3])};
        if (p.head_dim_v == 0 || tid * 4 < p.head_dim_v) *reinterpret_cast<u32x2*>(out) = o2;
        if (tid == 0) p.lse[b * p.lse_batch_stride + (int64_t)hq * p.lse_head_stride + t] = den > 0.f ? m + __logf(den) : -INFINITY;
    }
}

// more than this many partial rows per output row: one workgroup per row (decode_combine_wide_kernel)
constexpr int DEC_COMBINE_WIDE_MIN = 8;
template <typename T>
static void launch_decode_combine(const DecArgs& da, hipStream_t stream) {
    const fa_params& p = da.a.p;
    const int64_t rows = (int64_t)p.batch * p.nheads_q * p.seqlen_q;
    if (da.n_splits > DEC_COMBINE_WIDE_MIN) {
        hipLaunchKernelGGL(decode_combine_wide_kernel<T>, dim3((unsigned)rows), dim3(256), 0, stream, da);
    } else {
        const int64_t total = rows * (p.head_dim / 8);
        hipLaunchKernelGGL(decode_combine_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, da);
    }
}


// ---------------------------------------------------------------------------------------------
// fp8 KV, ONE query row per kv-head (T_q = 1, H_q == H_k: BASELINE config 4): streaming matrix-vector kernel.
//
// With a single query row the 32-row MFMA tile of fa_decode_kernel carries one useful column, and its fp8 path
// pays a dequantise + DOUBLED LDS round trip per byte (3.8 TB/s).  Here nothing touches
// LDS or the matrix pipe: 8 lanes own one key (16 fp8 bytes = 16 head-dim columns each), a wave instruction
// streams 8 keys (8 x 128 contiguous bytes), q . k is 16 FMAs per lane + a 3-step DPP sum over the 8 lanes,
// P V is 16 more FMAs into per-lane partial outputs; the 8 key groups of a wave keep their own (m, l, o)
// and merge once at the end (LDS), like split-KV.  VALU work: ~1.8 lane-instructions per byte, < 25 % of
// the vector rate at 5 TB/s - the kernel is bound by how 128-byte rows stream from HBM (tools/decode_rowsize_probe.py).
// k_descale folds into the softmax scale, v_descale into the final normalisation; RoPE on q as in fa_decode_kernel.
// ---------------------------------------------------------------------------------------------
constexpr int GEMV_THREADS = 256;
constexpr int GEMV_KEYS = 32;                  // keys per wave step: 4 loads x 8 keys (x K and V)

__device__ __forceinline__ float grp8_sum(float x) {
    // sum over the 8 consecutive lanes of a key group: xor 1, xor 2 (quad permutes), then the other quad (half-row mirror)
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));
    return x;
}

template <typename T, bool PAGED>
__global__ void __launch_bounds__(GEMV_THREADS) fa_decode_gemv_fp8_kernel(const DecArgs da) {
    using E = Elem<T>;
    constexpr int D = 128;
    __shared__ float red[32 * (D + 2)];                     // 4 waves x 8 key groups: o[128], m, l

    const KArgs& a = da.a;
    const fa_params& p = a.p;
    const int unit = blockIdx.x, split = blockIdx.y;
    const int b = unit / p.nheads_k, hk = unit - b * p.nheads_k;
    const int tid = threadIdx.x, lane = tid & 63, grp = lane >> 3, sub = lane & 7;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // varlen-q mode (decode through the varlen op): the sequence's ONE row sits at packed row cu_q[b]; sequences that bring
    // no row (a padded q: cu_q[-1] < total_q) are not ours - uniform per workgroup, before any barrier
    int64_t q_row0 = 0;
    if (da.cu_q) {
        q_row0 = da.cu_q[b];
        if (da.cu_q[b + 1] - (int)q_row0 != 1) return;
    }
    const int L = dec_cache_len(p, b);
    const int lp = p.cache_leftpad ? p.cache_leftpad[b] : 0;
    const int cb = p.cache_batch_idx ? p.cache_batch_idx[b] : b;
    const int seqlen_k = L + p.seqlen_new;
    const int off = seqlen_k - 1;                           // T_q == 1: the row sits at the bottom-right corner
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;
    int lo = 0, hi = seqlen_k - 1;
    if (wr >= 0) { const int h2 = off + wr; hi = h2 < hi ? h2 : hi; }
    if (wl >= 0) { const int l2 = off - wl; lo = l2 > lo ? l2 : lo; }

    // ---- q: this lane's 16 columns, RoPE as in fa_decode_kernel, rounded to the io type, then x scale ----
    float qs[16];
    {
        const int h = hk;                                   // G == 1
        const uint16_t* qrow = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.q_batch_stride + q_row0 * p.q_row_stride +
                               (int64_t)h * p.q_head_stride;
        const int half = p.rotary_dim >> 1;
        const int pos = L + lp;
        const uint16_t* cosp = reinterpret_cast<const uint16_t*>(p.rotary_cos) + (int64_t)pos * half;
        const uint16_t* sinp = reinterpret_cast<const uint16_t*>(p.rotary_sin) + (int64_t)pos * half;
        const float c = a.scale_log2e * p.k_descale;
#pragma unroll
        for (int cpart = 0; cpart < 2; ++cpart) {
            const int d_base = 16 * sub + 8 * cpart;
            u32x4 x = *reinterpret_cast<const u32x4*>(qrow + d_base);
            if (p.rotary_dim > 0 && d_base < p.rotary_dim && pos >= 0 && pos < p.seqlen_ro) {
                u32x4 xp = x;
                if (!p.rotary_interleaved) {
                    const int pd = d_base < half ? d_base + half : d_base - half;
                    xp = *reinterpret_cast<const u32x4*>(qrow + pd);
                }
                rope_chunk<T>(x, xp, cosp, sinp, d_base, p.rotary_dim, p.rotary_interleaved != 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { qs[8 * cpart + 2 * i] = E::lo(x[i]) * c; qs[8 * cpart + 2 * i + 1] = E::hi(x[i]) * c; }
        }
    }

    // ---- key range of this split; the 4 waves take alternate 32-key steps ----
    const int step_lo = lo / GEMV_KEYS, step_hi = hi >= lo ? hi / GEMV_KEYS + 1 : step_lo;
    const int n_all = step_hi - step_lo;
    const int per_split = ((n_all + da.n_splits - 1) / da.n_splits + 3) & ~3;
    const int s_lo = step_lo + split * per_split;
    int s_hi = s_lo + per_split; s_hi = s_hi < step_hi ? s_hi : step_hi;

    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k) + (int64_t)hk * p.k_head_stride + 16 * sub;
    const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v) + (int64_t)hk * p.v_head_stride + 16 * sub;
    const int32_t* btab = PAGED ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;

    float o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    auto addr = [&](int j, int64_t& ko, int64_t& vo) {
        const int pos = lp + j;
        if (PAGED) {
            const int pg = da.page_shift >= 0 ? (pos >> da.page_shift) : pos / p.page_block_size;
            const int pr = pos - pg * p.page_block_size;
            const int64_t phys = btab[pg];
            ko = phys * p.k_batch_stride + (int64_t)pr * p.k_row_stride;
            vo = phys * p.v_batch_stride + (int64_t)pr * p.v_row_stride;
        } else {
            ko = (int64_t)cb * p.k_batch_stride + (int64_t)pos * p.k_row_stride;
            vo = (int64_t)cb * p.v_batch_stride + (int64_t)pos * p.v_row_stride;
        }
    };
    // lane part of a load address: key group row + the lane's 16 columns (kbase / vbase carry the columns and the head)
    const uint32_t lane_k = (uint32_t)grp * (uint32_t)p.k_row_stride, lane_v = (uint32_t)grp * (uint32_t)p.v_row_stride;
    auto load_step = [&](int step, u32x4 (&kx)[4], u32x4 (&vx)[4]) {
        // Fast path (every full step that does not straddle a page): ONE block-table entry and ONE row offset per
        // step, both wave-uniform (scalar unit), lane-constant vector offsets - the general per-lane form below costs
        // ~40 quarter-rate integer multiplies per step (64-bit page offsets), more than the dot products themselves.
        const int j0 = step * GEMV_KEYS, pos0 = lp + j0;
        bool uni = j0 + GEMV_KEYS <= seqlen_k;
        int64_t kb = 0, vb = 0;
        if (PAGED) {
            const int pg0 = da.page_shift >= 0 ? (pos0 >> da.page_shift) : pos0 / p.page_block_size;
            const int pg1 = da.page_shift >= 0 ? ((pos0 + GEMV_KEYS - 1) >> da.page_shift) : (pos0 + GEMV_KEYS - 1) / p.page_block_size;
            uni = uni && pg0 == pg1;
            if (uni) {
                const int64_t phys = btab[pg0];
                const int pr0 = pos0 - pg0 * p.page_block_size;
                kb = phys * p.k_batch_stride + (int64_t)pr0 * p.k_row_stride;
                vb = phys * p.v_batch_stride + (int64_t)pr0 * p.v_row_stride;
            }
        } else {
            kb = (int64_t)cb * p.k_batch_stride + (int64_t)pos0 * p.k_row_stride;
            vb = (int64_t)cb * p.v_batch_stride + (int64_t)pos0 * p.v_row_stride;
        }
        if (uni) {
            const uint8_t* kr = kbase + kb;
            const uint8_t* vr = vbase + vb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kx[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kr + (int64_t)(8 * i) * p.k_row_stride + lane_k));
                vx[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vr + (int64_t)(8 * i) * p.v_row_stride + lane_v));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = step * GEMV_KEYS + 8 * i + grp;
            const int jc = j < seqlen_k ? j : (seqlen_k > 0 ? seqlen_k - 1 : 0);      // clamp: masked below, never faults
            int64_t ko, vo;
            addr(jc, ko, vo);
            kx[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kbase + ko));
            vx[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vbase + vo));
        }
    };
    auto dot16 = [&](const u32x4& w) {
        float acc = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) {
            const f32x2 a0 = __builtin_amdgcn_cvt_pk_f32_fp8(w[d4], false);
            const f32x2 a1 = __builtin_amdgcn_cvt_pk_f32_fp8(w[d4], true);
            acc = fmaf(a0[0], qs[4 * d4 + 0], acc); acc = fmaf(a0[1], qs[4 * d4 + 1], acc);
            acc = fmaf(a1[0], qs[4 * d4 + 2], acc); acc = fmaf(a1[1], qs[4 * d4 + 3], acc);
        }
        return acc;
    };
    auto compute_step = [&](int step, const u32x4 (&kx)[4], const u32x4 (&vx)[4]) {
        float sv[4];
        float mx = m_run;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = step * GEMV_KEYS + 8 * i + grp;
            const float sd = grp8_sum(dot16(kx[i]));
            sv[i] = (j >= lo && j <= hi) ? sd : -INFINITY;
            mx = fmaxf(mx, sv[i]);
        }
        const float m_use = (mx == -INFINITY) ? 0.f : mx;
        const float alpha = fast_exp2(m_run - m_use);        // m_run = -inf -> 0
        m_run = mx;
        float pw[4], ps = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { pw[i] = fast_exp2(sv[i] - m_use); ps += pw[i]; }
        l_run = fmaf(l_run, alpha, ps);
#pragma unroll
        for (int x = 0; x < 16; ++x) o[x] *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                const f32x2 a0 = __builtin_amdgcn_cvt_pk_f32_fp8(vx[i][d4], false);
                const f32x2 a1 = __builtin_amdgcn_cvt_pk_f32_fp8(vx[i][d4], true);
                o[4 * d4 + 0] = fmaf(pw[i], a0[0], o[4 * d4 + 0]); o[4 * d4 + 1] = fmaf(pw[i], a0[1], o[4 * d4 + 1]);
                o[4 * d4 + 2] = fmaf(pw[i], a1[0], o[4 * d4 + 2]); o[4 * d4 + 3] = fmaf(pw[i], a1[1], o[4 * d4 + 3]);
            }
        }
    };

    // two register sets: the loads of step s + 4 are in flight while step s is consumed
    u32x4 kA[4], vA[4], kB[4], vB[4];
    int st = s_lo + wave;
    if (st < s_hi) load_step(st, kA, vA);
    for (; st < s_hi; st += 8) {
        if (st + 4 < s_hi) load_step(st + 4, kB, vB);
        compute_step(st, kA, vA);
        if (st + 4 < s_hi) {
            if (st + 8 < s_hi) load_step(st + 8, kA, vA);
            compute_step(st + 4, kB, vB);
        }
    }

    // ---- merge the 32 key groups of the workgroup ----
    float* mine = red + (wave * 8 + grp) * (D + 2);
#pragma unroll
    for (int x = 0; x < 16; ++x) mine[16 * sub + x] = o[x];
    if (sub == 0) { mine[D] = m_run; mine[D + 1] = l_run; }
    __syncthreads();
    if (tid < D) {
        float m_all = -INFINITY;
        for (int g2 = 0; g2 < 32; ++g2) m_all = fmaxf(m_all, red[g2 * (D + 2) + D]);
        const float m_s = (m_all == -INFINITY) ? 0.f : m_all;
        float l_all = 0.f, acc = 0.f;
        for (int g2 = 0; g2 < 32; ++g2) {
            const float sc = fast_exp2(red[g2 * (D + 2) + D] - m_s);
            l_all = fmaf(red[g2 * (D + 2) + D + 1], sc, l_all);
            acc = fmaf(red[g2 * (D + 2) + tid], sc, acc);
        }
        const float inv = l_all > 0.f ? p.v_descale / l_all : 0.f;
        const float lse = l_all > 0.f ? (m_all + fast_log2(l_all)) * kLn2 : -INFINITY;
        const int hq = hk;
        if (da.n_splits == 1) {
            uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + (int64_t)b * p.o_batch_stride + q_row0 * p.o_row_stride +
                           (int64_t)hq * p.o_head_stride;
            const float val = acc * inv;
            const float other = __shfl_xor(val, 1);
            if ((tid & 1) == 0) *reinterpret_cast<uint32_t*>(op + tid) = E::pack2(val, other);
            if (tid == 0) p.lse[(int64_t)b * p.lse_batch_stride + (int64_t)hq * p.lse_head_stride + q_row0] = lse;
        } else {
            const int64_t prow = ((int64_t)split * p.batch + b) * p.nheads_q + hq;
            da.o_partial[prow * D + tid] = acc * inv;
            if (tid == 0) da.lse_partial[prow] = lse;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same matrix-vector decode, TOKEN-major: a workgroup owns a key range of ONE batch entry for ALL heads; a wave
// instruction covers one token x 8 heads = 1 KiB of contiguous cache (heads are adjacent in a cache row), wave w takes
// heads 8w .. 8w+7 (+32 per round).  tools/probes/probe_kv_stream.hip: with one workgroup per head (above) the 128-byte
// pieces at 4 KiB stride stream at 6.1 TB/s even without any arithmetic; token-major streams at 7.1-7.2 TB/s.  Every
// 8-lane group owns a head (no merge inside the workgroup); the key range is always split (grid = batch x splits).
// ---------------------------------------------------------------------------------------------
// GQA: the lane group of a kv-head keeps G query rows (their q chunks, running (m, l) and output columns): the cache
// row is converted once and used G times.  Few kv-heads (fewer head groups than waves): the waves that share a head
// group take contiguous sub-ranges of the split's keys and write their own partial rows (KSUB per split), so the
// combine kernel sees n_splits x KSUB partials.
__host__ __device__ inline int gemv_tm_hpw(const fa_params& p) { return p.kv_dtype == FA_FP8_E4M3 ? 8 : 4; }
__host__ __device__ inline int gemv_tm_ksub(const fa_params& p) {
    const int n_hg = p.nheads_k / gemv_tm_hpw(p);
    return n_hg >= 4 || n_hg == 3 ? 1 : (n_hg == 2 ? 2 : 4);
}
__host__ __device__ inline bool gemv_tm_applicable(const fa_params& p) {
    const bool kv8 = p.kv_dtype == FA_FP8_E4M3;
    if (!kv8 && p.kv_dtype != p.dtype) return false;
    if (p.head_dim != 128 || p.head_dim_v != 0 || p.seqlen_q != 1 || p.alibi_slopes || p.softcap > 0.f) return false;
    if (p.nheads_k < 1 || p.nheads_q % p.nheads_k) return false;
    const int G = p.nheads_q / p.nheads_k;
    // (GQA groups up to 4: VALU-bound - H 32/8, round 2: fp8 4.0 vs 3.7 TB/s on the MFMA kernel, fp16 5.6 vs 5.4)
    // Groups of 1 and 2.  Groups of 4 were VALU-bound here (fp8 4.0-4.4 TB/s, 16 bit 5.5): since the MFMA decode kernel runs
    // two waves per SIMD it is 3-43 % faster on fp8 caches (H 32/8: B 64 285 -> 223 us, B 8 55 -> 38, B 1 over 32 k 52 -> 30)
    // and 4-9 % on 16-bit ones (B 1 33.0 -> 30.0, B 8 61.4 -> 57.4, B 64 389 -> 372; B 32 196 vs 200), and pages of 16
    // tokens cost it nothing (this kernel restarts its pipeline per page: + 12 %)
    if (!(G == 1 || G == 2)) return false;
    return p.nheads_k % gemv_tm_hpw(p) == 0 && p.k_head_stride == 128 && p.v_head_stride == 128;
}

// acc + a.lo * b.lo + a.hi * b.hi on packed 16-bit pairs, fp32 accumulate: one VALU op per two elements (v_dot2_f32_f16 /
// v_dot2_f32_bf16).  Through the builtins, not inline asm: the result feeds a DPP sum, and the compiler only inserts
// the VALU -> DPP wait states behind instructions it knows.
template <typename T> __device__ __forceinline__ float dot2_acc(uint32_t a, uint32_t b, float acc);
template <> __device__ __forceinline__ float dot2_acc<fp16_tag>(uint32_t a, uint32_t b, float acc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), acc, false);
}
template <> __device__ __forceinline__ float dot2_acc<bf16_tag>(uint32_t a, uint32_t b, float acc) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a), __builtin_bit_cast(b2, b), acc, false);
}

__device__ __forceinline__ float grp16_sum(float x) {
    x = grp8_sum(x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));   // row mirror
    return x;
}

template <typename T, bool PAGED, bool KV8, int G>
__global__ void __launch_bounds__(GEMV_THREADS) fa_decode_gemv_tm_kernel(const DecArgs da) {
    using E = Elem<T>;
    constexpr int D = 128;
    constexpr int LPH = KV8 ? 8 : 16;          // lanes per head (16 bytes of the head's row each)
    constexpr int HPW = 64 / LPH;              // kv-heads per wave instruction
    constexpr int CPL = KV8 ? 16 : 8;          // head-dim columns per lane
    constexpr int ES = KV8 ? 1 : 2;            // bytes per cache element
    constexpr int KPS = (G * CPL >= 64) ? 2 : 4;   // keys per wave step (x K and V loads): register budget
    const KArgs& a = da.a;
    const fa_params& p = a.p;
    const int b = blockIdx.x, split = blockIdx.y, n_grid_splits = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, grp = lane / LPH, sub = lane % LPH;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_hg = p.nheads_k / HPW;
    const int ksub_n = da.ksub;                                   // waves per head group (1, 2 or 4)
    const int hg0 = ksub_n == 1 ? wave : wave % n_hg;
    const int ksub = ksub_n == 1 ? 0 : wave / n_hg;
    const int hg_step = ksub_n == 1 ? 4 : n_hg;                   // (with sub-ranges every head group has its waves: one round)

    int64_t q_row0 = 0;                                           // varlen-q mode: as in fa_decode_gemv_fp8_kernel
    if (da.cu_q) {
        q_row0 = da.cu_q[b];
        if (da.cu_q[b + 1] - (int)q_row0 != 1) return;
    }
    const int L = dec_cache_len(p, b);
    const int lp = p.cache_leftpad ? p.cache_leftpad[b] : 0;
    const int cb = p.cache_batch_idx ? p.cache_batch_idx[b] : b;
    const int seqlen_k = L + p.seqlen_new;
    const int off = seqlen_k - 1;
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;
    int lo = 0, hi = seqlen_k - 1;
    if (wr >= 0) { const int h2 = off + wr; hi = h2 < hi ? h2 : hi; }
    if (wl >= 0) { const int l2 = off - wl; lo = l2 > lo ? l2 : lo; }
    // key range of this (split, sub-range), in whole steps
    const int n_keys = hi >= lo ? hi - lo + 1 : 0;
    const int n_parts = n_grid_splits * ksub_n;
    const int per_part = (((n_keys + n_parts - 1) / n_parts) + KPS - 1) / KPS * KPS;
    const int part = split * ksub_n + ksub;
    const int k_lo = lo + part * per_part;
    int k_hi = k_lo + per_part; k_hi = k_hi < hi + 1 ? k_hi : hi + 1;              // exclusive

    const int32_t* btab = PAGED ? p.block_table + (int64_t)b * p.block_table_batch_stride : nullptr;
    const int half = p.rotary_dim >> 1;
    const int pos_q = L + lp;
    const uint16_t* cosp = reinterpret_cast<const uint16_t*>(p.rotary_cos) + (int64_t)pos_q * half;
    const uint16_t* sinp = reinterpret_cast<const uint16_t*>(p.rotary_sin) + (int64_t)pos_q * half;
    const float c = a.scale_log2e * p.k_descale;

    for (int hg = hg0; hg < n_hg; hg += hg_step) {
        const int h = HPW * hg + grp;                             // kv-head of this lane group
        // ---- q rows of the head's group: CPL columns each, RoPE, x scale ----
        float qs[G][CPL];
        uint32_t qh[G][4];                                        // 16-bit caches: the row's packed pairs (v_dot2), scaled after the sum
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const uint16_t* qrow = reinterpret_cast<const uint16_t*>(p.q) + (int64_t)b * p.q_batch_stride + q_row0 * p.q_row_stride +
                                   (int64_t)(h * G + gi) * p.q_head_stride;
#pragma unroll
            for (int cpart = 0; cpart < CPL / 8; ++cpart) {
                const int d_base = CPL * sub + 8 * cpart;
                u32x4 x = *reinterpret_cast<const u32x4*>(qrow + d_base);
                if (p.rotary_dim > 0 && d_base < p.rotary_dim && pos_q >= 0 && pos_q < p.seqlen_ro) {
                    u32x4 xp = x;
                    if (!p.rotary_interleaved) {
                        const int pd = d_base < half ? d_base + half : d_base - half;
                        xp = *reinterpret_cast<const u32x4*>(qrow + pd);
                    }
                    rope_chunk<T>(x, xp, cosp, sinp, d_base, p.rotary_dim, p.rotary_interleaved != 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { qs[gi][8 * cpart + 2 * i] = E::lo(x[i]) * c; qs[gi][8 * cpart + 2 * i + 1] = E::hi(x[i]) * c; }
                if constexpr (!KV8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) qh[gi][i] = x[i];
                }
            }
        }
        (void)qh;
        const uint32_t lane_off = (uint32_t)h * (128u * ES) + 16u * (uint32_t)sub;   // bytes inside a cache row
        const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k) + lane_off;
        const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v) + lane_off;

        float o[G][CPL];
        float m_run[G], l_run[G];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            m_run[gi] = -INFINITY; l_run[gi] = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) o[gi][i] = 0.f;
        }

        // A page (or the whole dense cache) is a SEGMENT of consecutive rows: its base is looked up once (the lookup
        // needs a full wait on the memory queue - done per key it serialised every load, 4.7 TB/s), the loads inside a
        // segment are address arithmetic on the scalar unit and stay a step deep in flight.
        const uint8_t* kseg = nullptr;
        const uint8_t* vseg = nullptr;
        int seg_n = 0;                                             // rows in the current segment
        auto load_step = [&](int t0, u32x4 (&kx)[KPS], u32x4 (&vx)[KPS]) {
#pragma unroll
            for (int i = 0; i < KPS; ++i) {
                int t = t0 + i;
                t = t < seg_n ? t : seg_n - 1;                     // tail of the segment: reload its last row (masked below)
                kx[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kseg + (int64_t)t * p.k_row_stride * ES));
                vx[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vseg + (int64_t)t * p.v_row_stride * ES));
            }
        };
        auto unpack = [&](const u32x4& w, float (&f)[CPL]) {       // one cache chunk -> fp32, once for all G rows
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                if constexpr (KV8) {
                    const f32x2 a0 = __builtin_amdgcn_cvt_pk_f32_fp8(w[d4], false);
                    const f32x2 a1 = __builtin_amdgcn_cvt_pk_f32_fp8(w[d4], true);
                    f[4 * d4 + 0] = a0[0]; f[4 * d4 + 1] = a0[1]; f[4 * d4 + 2] = a1[0]; f[4 * d4 + 3] = a1[1];
                } else {
                    f[2 * d4 + 0] = E::lo(w[d4]); f[2 * d4 + 1] = E::hi(w[d4]);
                }
            }
        };
        auto compute_step = [&](int j0, const u32x4 (&kx)[KPS], const u32x4 (&vx)[KPS]) {
            float sv[G][KPS];
#pragma unroll
            for (int i = 0; i < KPS; ++i) {
                if constexpr (KV8) {
                    float kf[CPL];
                    unpack(kx[i], kf);
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        float acc = 0.f;
#pragma unroll
                        for (int x = 0; x < CPL; ++x) acc = fmaf(kf[x], qs[gi][x], acc);
                        const float sd = grp8_sum(acc);
                        sv[gi][i] = (j0 + i < seg_n) ? sd : -INFINITY;
                    }
                } else {
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        float acc = 0.f;
#pragma unroll
                        for (int x = 0; x < 4; ++x) acc = dot2_acc<T>(kx[i][x], qh[gi][x], acc);
                        const float sd = grp16_sum(acc) * c;
                        sv[gi][i] = (j0 + i < seg_n) ? sd : -INFINITY;
                    }
                }
            }
            float pw[G][KPS];
#pragma unroll
            for (int gi = 0; gi < G; ++gi) {
                float mx = m_run[gi];
#pragma unroll
                for (int i = 0; i < KPS; ++i) mx = fmaxf(mx, sv[gi][i]);
                const float m_use = (mx == -INFINITY) ? 0.f : mx;
                const float alpha = fast_exp2(m_run[gi] - m_use);
                m_run[gi] = mx;
                float ps = 0.f;
#pragma unroll
                for (int i = 0; i < KPS; ++i) { pw[gi][i] = fast_exp2(sv[gi][i] - m_use); ps += pw[gi][i]; }
                l_run[gi] = fmaf(l_run[gi], alpha, ps);
#pragma unroll
                for (int x = 0; x < CPL; ++x) o[gi][x] *= alpha;
            }
#pragma unroll
            for (int i = 0; i < KPS; ++i) {
                float vf[CPL];
                unpack(vx[i], vf);
#pragma unroll
                for (int gi = 0; gi < G; ++gi)
#pragma unroll
                    for (int x = 0; x < CPL; ++x) o[gi][x] = fmaf(pw[gi][i], vf[x], o[gi][x]);
            }
        };

        // two register sets: the loads of the next step are in flight while a step is consumed
        u32x4 kA[KPS], vA[KPS], kB[KPS], vB[KPS];
        int j = k_lo;
        while (j < k_hi) {
            const int pos = lp + j;
            int64_t ko, vo;
            if (PAGED) {
                const int pg = da.page_shift >= 0 ? (pos >> da.page_shift) : pos / p.page_block_size;
                const int pr = pos - pg * p.page_block_size;
                const int64_t phys = __builtin_amdgcn_readfirstlane(btab[pg]);
                ko = (phys * p.k_batch_stride + (int64_t)pr * p.k_row_stride) * ES;
                vo = (phys * p.v_batch_stride + (int64_t)pr * p.v_row_stride) * ES;
                seg_n = p.page_block_size - pr;
            } else {
                ko = ((int64_t)cb * p.k_batch_stride + (int64_t)pos * p.k_row_stride) * ES;
                vo = ((int64_t)cb * p.v_batch_stride + (int64_t)pos * p.v_row_stride) * ES;
                seg_n = k_hi - j;
            }
            seg_n = seg_n < k_hi - j ? seg_n : k_hi - j;
            kseg = kbase + ko;
            vseg = vbase + vo;
            load_step(0, kA, vA);
            for (int t = 0; t < seg_n; t += 2 * KPS) {
                if (t + KPS < seg_n) load_step(t + KPS, kB, vB);
                compute_step(t, kA, vA);
                if (t + KPS < seg_n) {
                    if (t + 2 * KPS < seg_n) load_step(t + 2 * KPS, kA, vA);
                    compute_step(t + KPS, kB, vB);
                }
            }
            j += seg_n;
        }

        // ---- the lane group's q-heads: normalised partial (or final) output ----
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const int hq = h * G + gi;
            const float inv = l_run[gi] > 0.f ? p.v_descale / l_run[gi] : 0.f;
            const float lse = l_run[gi] > 0.f ? (m_run[gi] + fast_log2(l_run[gi])) * kLn2 : -INFINITY;
            if (da.n_splits == 1) {
                uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + (int64_t)b * p.o_batch_stride + q_row0 * p.o_row_stride +
                               (int64_t)hq * p.o_head_stride + CPL * sub;
#pragma unroll
                for (int c8 = 0; c8 < CPL / 8; ++c8) {
                    u32x4 w0;
#pragma unroll
                    for (int x = 0; x < 4; ++x) w0[x] = E::pack2(o[gi][8 * c8 + 2 * x] * inv, o[gi][8 * c8 + 2 * x + 1] * inv);
                    *reinterpret_cast<u32x4*>(op + 8 * c8) = w0;
                }
                if (sub == 0) p.lse[(int64_t)b * p.lse_batch_stride + (int64_t)hq * p.lse_head_stride + q_row0] = lse;
            } else {
                const int64_t prow = ((int64_t)part * p.batch + b) * p.nheads_q + hq;
                float* dst = da.o_partial + prow * D + CPL * sub;
#pragma unroll
                for (int x = 0; x < CPL; x += 4) {
                    f32x4 w = {o[gi][x] * inv, o[gi][x + 1] * inv, o[gi][x + 2] * inv, o[gi][x + 3] * inv};
                    *reinterpret_cast<f32x4*>(dst + x) = w;
                }
                if (sub == 0) da.lse_partial[prow] = lse;
            }
        }
    }
}

// ---- host ---------------------------------------------------------------------------------------
bool decode_applicable(const fa_params& p) {
    // widths 64 / 128 (16-bit and fp8 caches) and 256 (16-bit); narrower rows (head_dim_v valid columns, 16-bit caches)
    const bool kv8 = p.kv_dtype == FA_FP8_E4M3;
    if (!(p.head_dim == 64 || p.head_dim == 128 || (p.head_dim == 256 && !kv8))) return false;
    if (p.head_dim_v != 0 && (kv8 || p.head_dim_v % 8 != 0 || p.head_dim_v > p.head_dim)) return false;
    return true;                  // any number of packed rows, 32 per workgroup (blockIdx.z); decode_takes() decides who runs
}

// Which kernel serves a multi-token query block (speculative / tree decode, chunked prefill) over the cache?  Both stream
// the K / V of a kv-head once per "pass": the decode kernel per 32 PACKED rows (t x G + g: the heads of a group share the
// stream, split-KV fills the chip at small batch; row blocks after the first read from L2), fa_fwd_kernel per 128 query
// positions of ONE head (no packing: G passes per kv-head, no split).  At large batch a pass costs about the same in
// both (tools/spec_decode_sweep.py: ~150 us per row block vs 156 us per head pass at B 64, H 64/8, 4 k fp8), so the
// decode kernel runs while it needs no more passes - i.e. up to 32 query positions whatever the group size (16-bit
// caches stopped at 32 packed ROWS before: T_q 9 at G = 4 fell onto the general path at 6 x the time).  When the general
// path cannot fill the chip (batch x heads x 128-row blocks <= CUs) the decode kernel's split-KV is worth up to
// 8 x the passes (B 1, H 32/8, T_q 128 over 8 k: 72 us against 240).  fp8 caches with two row blocks stay here too.
bool decode_takes(const fa_params& p) {
    if (!decode_applicable(p)) return false;
    const int G = p.nheads_q / p.nheads_k;
    const int row_blocks = (p.seqlen_q * G + 31) / 32;
    const int fwd_blocks = (p.seqlen_q + 127) / 128;
    const int fwd_passes = G * fwd_blocks;
    const int64_t fwd_wgs = (int64_t)p.batch * p.nheads_q * fwd_blocks;
    // (chunked_prefill_probe.py with FA_DEC_FACTOR: at one general-path workgroup per CU or fewer the row blocks win even at
    //  4 x the passes - B 1, T_q 512, H 64/8 over 32 k: 0.92 ms against 1.34 -; at two per CU the general path wins)
    const int cus = fa_device_cu_count();
    int64_t factor = fwd_wgs <= cus ? 8 : (fwd_wgs < 2 * cus ? 2 : 1);
    if (row_blocks <= factor * fwd_passes) return true;
    return p.kv_dtype == FA_FP8_E4M3 && row_blocks <= 2;
}


// grid splits of the key range (x up to 4 key sub-ranges per workgroup in the token-major kernel: <= 1024 partial rows per
// output row, merged by decode_combine_wide_kernel)
constexpr int DEC_MAX_SPLITS = 256;
static bool decode_eight_waves(const fa_params& p);
// a head per wave (HPW in fa_decode_kernel): D = 128, kv-heads adjacent in the cache rows and a multiple of 8, the packed query
// rows of a kv-head in one 32-row block; the token-major kernel keeps its shapes
#ifndef FA_DEC_HPW
#define FA_DEC_HPW 1
#endif
static bool decode_hpw(const fa_params& p) {
    if (!FA_DEC_HPW || p.head_dim != 128 || p.head_dim_v != 0 || p.nheads_k < 8 || p.nheads_k % 8 != 0) return false;
    if (p.k_head_stride != p.head_dim || p.v_head_stride != p.head_dim || p.nheads_q % p.nheads_k != 0) return false;
    if (p.kv_dtype != FA_FP8_E4M3 || !FA_DEC_F8M) return false;      // (fp8 caches; 16-bit ones keep a workgroup per kv-head)
    if (p.seqlen_q * (p.nheads_q / p.nheads_k) > 32) return false;
    return !gemv_tm_applicable(p) && decode_eight_waves(p);
}
int decode_num_splits(const fa_params& p) {
    if (p.num_splits >= 1) return p.num_splits > DEC_MAX_SPLITS ? DEC_MAX_SPLITS : p.num_splits;
    const int units = decode_hpw(p) ? p.batch * (p.nheads_k / 8)      // (a workgroup streams eight kv-heads)
                                    : p.batch * p.nheads_k * ((p.seqlen_q * (p.nheads_q / p.nheads_k) + 31) / 32);   // x row blocks
    const int max_tiles = (p.seqlen_k + DEC_BN - 1) / DEC_BN;
    int s = 1;
    // one workgroup per CU: every further doubling costs 6-15 % in this kernel (each split re-reads the query rows and
    // writes a partial row; tools/decode_splits_sweep.py, H 64/8 and 32/2, B 2-64)
    while (units * s < fa_device_cu_count() && s < 64 && max_tiles / (s * 2) >= 8) s *= 2;       // >= 8 tiles (256 keys) per split
    // The streaming fp8 kernel keeps three workgroups per CU resident: with 4096 units on 256 CUs the grid runs 5.33
    // "rounds" and the last one is a third full (11 % of the time at a third of the rate).  Split the key range so that
    // the grid is a near-multiple of what is resident; the partials cost 4 MB and one tiny combine launch.
    if (gemv_tm_applicable(p)) {
        // token-major streaming kernel: grid = batch x splits.  Measured at config 4
        // (tools/decode_splits_probe.py): one workgroup per CU in ONE round is best, every doubling
        // beyond costs ~2 %, a partial last round costs its idle share - pick the split count with the best of both.
        // (one workgroup per CU streams best at 32 kv-heads: 2 splits 7.0 TB/s, 4 splits 6.8, 12 splits 6.4; with few kv-heads -
        // the waves of a workgroup share head groups and take key sub-ranges - two per CU are 11-16 % faster than one
        // and than four: tools/decode_splits_sweep.py, H 32/8 fp16 and fp8, B 16-256)
        const double resident = (gemv_tm_ksub(p) > 1 ? 2.0 : 1.0) * fa_device_cu_count();
        const int cap = p.seqlen_k / 32 > 0 ? (p.seqlen_k / 32 < DEC_MAX_SPLITS ? p.seqlen_k / 32 : DEC_MAX_SPLITS) : 1;     // >= 32 keys per split
        int best = 1;
        double best_score = -1.0;
        for (int k = 1; k <= cap; ++k) {
            const double r = p.batch * (double)k / resident;
            const double eff = r / (double)(int64_t)(r + 0.999999);
            double pen = 0.0;
            for (int t = k; t > 1; t >>= 1) pen += 0.02;
            const double score = eff - pen;
            if (score > best_score + 1e-9) { best_score = score; best = k; }
        }
        return best;
    }
    const bool gemv = p.kv_dtype == FA_FP8_E4M3 && p.head_dim == 128 && p.seqlen_q == 1 && p.nheads_q == p.nheads_k;
    if (gemv && s == 1) {
        const double resident = 3.0 * fa_device_cu_count();
        auto eff = [&](int k) { const double r = units * (double)k / resident; return r / (double)(int64_t)(r + 0.999999); };
        for (int k = 1; k <= 8; ++k) {
            if (k > 1 && max_tiles / k < 16) break;                             // >= 512 keys per split
            if (eff(k) >= 0.94) { s = k; break; }
        }
    }
    return s;
}

// partial (O, LSE) rows per (batch, head): grid splits x the token-major kernel's key sub-ranges
static int decode_num_partials(const fa_params& p) {
    return decode_num_splits(p) * (gemv_tm_applicable(p) ? gemv_tm_ksub(p) : 1);
}

size_t decode_split_workspace_bytes(const fa_params& p) {
    const int s = decode_num_partials(p);
    if (s <= 1) return 0;
    const size_t rows = (size_t)p.batch * p.nheads_q * p.seqlen_q;
    return (size_t)s * rows * (p.head_dim + 1) * sizeof(float);
}

// two waves per SIMD for the MFMA decode kernel?  (FA_DEC_NW = 4 / 8 forces it: A/B in tools/decode_splits_sweep.py)
static bool decode_eight_waves(const fa_params& p) {
    static const int forced = [] { const char* e = getenv("FA_DEC_NW"); return e ? (int)e[0] : 0; }();   // (read once, not per call)
    if (forced == '8') return true;
    if (forced == '4') return false;
    // head dims up to 128: the SIMD's second wave fills the first one's waits (LDS round trips, MFMA chains, loads) - 7-24 %
    // faster on every shape measured (profiles/r03_decode_features.txt (5)); D = 256 keeps its 128 accumulator registers
    // and four waves
    return p.head_dim <= 128;
}

template <typename T, int D>
static int launch_decode_td(DecArgs& da, hipStream_t stream) {
    const fa_params& p = da.a.p;
    const bool kv8 = p.kv_dtype == FA_FP8_E4M3;
    const bool paged = p.block_table != nullptr;
    da.n_rb = (da.rows + 31) / 32;
    da.grid_splits = da.n_splits;
    dim3 grid(p.batch * p.nheads_k, da.n_splits, 1);
    if (da.n_rb > 1) grid = dim3((unsigned)(((p.batch * p.nheads_k * da.n_splits + 7) / 8) * 8 * da.n_rb), 1, 1);
    if constexpr (D == 128) {
        // one query position, heads adjacent in the cache rows: the token-major streaming kernel (fp8 and 16-bit caches, GQA)
        if (gemv_tm_applicable(p)) {
            dim3 grid_tm(p.batch, da.n_splits / da.ksub);
#define FA_LAUNCH_TM(KV8_, G_)                                                                                      \
            do {                                                                                                    \
                if (paged) hipLaunchKernelGGL((fa_decode_gemv_tm_kernel<T, true, KV8_, G_>), grid_tm, dim3(GEMV_THREADS), 0, stream, da);  \
                else       hipLaunchKernelGGL((fa_decode_gemv_tm_kernel<T, false, KV8_, G_>), grid_tm, dim3(GEMV_THREADS), 0, stream, da); \
            } while (0)
            if (kv8) { if (da.group == 1) FA_LAUNCH_TM(true, 1); else FA_LAUNCH_TM(true, 2); }
            else { if (da.group == 1) FA_LAUNCH_TM(false, 1); else FA_LAUNCH_TM(false, 2); }
#undef FA_LAUNCH_TM
            if (da.n_splits > 1) launch_decode_combine<T>(da, stream);
            return 0;
        }
        // fp8 cache, one query row per kv-head, a head layout the token-major kernel does not take: one workgroup per head
        if (kv8 && da.rows == 1 && da.group == 1 && !da.bias) {
            if (paged) hipLaunchKernelGGL((fa_decode_gemv_fp8_kernel<T, true>), grid, dim3(GEMV_THREADS), 0, stream, da);
            else       hipLaunchKernelGGL((fa_decode_gemv_fp8_kernel<T, false>), grid, dim3(GEMV_THREADS), 0, stream, da);
            if (da.n_splits > 1) launch_decode_combine<T>(da, stream);
            return 0;
        }
    }
    if constexpr (D == 128) {
        if (decode_hpw(p)) {
            const dim3 grid_h(p.batch * (p.nheads_k / 8), da.n_splits, 1);
            const size_t smem_ = DecSmem<D, 8>::QOFF;                       // the waves' tile regions only: Q in registers, no merge
#define FA_LAUNCH_HPW(KV8_, PAGED_, F8M_)                                                                          \
            do {                                                                                                    \
                auto kern = fa_decode_kernel<T, D, KV8_, PAGED_, false, 8, F8M_, true>;                            \
                FA_SET_LDS_ONCE(kern, smem_);                                                                       \
                hipLaunchKernelGGL(kern, grid_h, dim3(512), smem_, stream, da);                                     \
            } while (0)
            if (kv8) {
#if FA_DEC_F8M
                if (paged) FA_LAUNCH_HPW(true, true, true); else FA_LAUNCH_HPW(true, false, true);
#endif
            } else {
            }
#undef FA_LAUNCH_HPW
            if (da.n_splits > 1) launch_decode_combine<T>(da, stream);
            return 0;
        }
    }
#define FA_LAUNCH_DEC(KV8, PAGED, NARROW, NW_)                                                                      \
    do {                                                                                                            \
        auto kern = fa_decode_kernel<T, D, KV8, PAGED, NARROW, NW_>;                                                \
        const size_t smem_ = DecSmem<D, NW_>::TOTAL;                                                                \
        FA_SET_LDS_ONCE(kern, smem_);                                                                               \
        hipLaunchKernelGGL(kern, grid, dim3(64 * NW_), smem_, stream, da);                                          \
    } while (0)
    const bool narrow = p.head_dim_v != 0;
    if constexpr (D <= 128) {
        // eight waves (two per SIMD, one LDS stage and 256 registers each) or four (one per SIMD, two stages)
        const bool w8 = decode_eight_waves(p);
        if (kv8) {
#if FA_DEC_F8M
            if constexpr (D == 128) {
                // the fp8 cache feeds v_mfma_f32_32x32x16_fp8_fp8 as stored (F8M above); FA_DEC_F8M=0 builds keep the dequantising form
                if (w8) {
                    auto kern = paged ? fa_decode_kernel<T, D, true, true, false, 8, true> : fa_decode_kernel<T, D, true, false, false, 8, true>;
                    const size_t smem_ = DecSmem<D, 8>::QOFF + DecSmem<D, 8>::QBYTES + (D / 16) * 64 * 8;    // (+ the third Q term)
                    static_assert(DecSmem<D, 8>::QOFF + DecSmem<D, 8>::QBYTES + (D / 16) * 64 * 8 >= DecSmem<D, 8>::MERGE, "merge area");
                    if (paged) { FA_SET_LDS_ONCE((fa_decode_kernel<T, D, true, true, false, 8, true>), smem_); }
                    else       { FA_SET_LDS_ONCE((fa_decode_kernel<T, D, true, false, false, 8, true>), smem_); }
                    hipLaunchKernelGGL(kern, grid, dim3(64 * 8), smem_, stream, da);
                    if (da.n_splits > 1) launch_decode_combine<T>(da, stream);
                    return 0;
                }
            }
#endif
            if (w8) { if (paged) FA_LAUNCH_DEC(true, true, false, 8); else FA_LAUNCH_DEC(true, false, false, 8); }
            else    { if (paged) FA_LAUNCH_DEC(true, true, false, 4); else FA_LAUNCH_DEC(true, false, false, 4); }
        } else if (!narrow) {
            if (w8) { if (paged) FA_LAUNCH_DEC(false, true, false, 8); else FA_LAUNCH_DEC(false, false, false, 8); }
            else    { if (paged) FA_LAUNCH_DEC(false, true, false, 4); else FA_LAUNCH_DEC(false, false, false, 4); }
        }
    }
    if (!kv8 && (narrow || D > 128)) {
        // one instantiation serves full-width D = 256 rows and every narrow width (vcols is a run-time count there)
        if (D > 128 && !narrow) da.a.p.head_dim_v = D;
        if constexpr (D <= 128) {
            if (decode_eight_waves(p)) { if (paged) FA_LAUNCH_DEC(false, true, true, 8); else FA_LAUNCH_DEC(false, false, true, 8); }
            else                       { if (paged) FA_LAUNCH_DEC(false, true, true, 4); else FA_LAUNCH_DEC(false, false, true, 4); }
        } else {
            if (paged) FA_LAUNCH_DEC(false, true, true, 4); else FA_LAUNCH_DEC(false, false, true, 4);
        }
    }
#undef FA_LAUNCH_DEC
    if (da.n_splits > 1) launch_decode_combine<T>(da, stream);
    return 0;
}

// workspace layout: [o_partial | lse_partial] at `ws`
int launch_decode_splitkv(const KArgs& a, void* ws, hipStream_t stream) {
    const fa_params& p = a.p;
    DecArgs da;
    da.a = a;
    da.group = p.nheads_q / p.nheads_k;
    da.rows = p.seqlen_q * da.group;
    da.local = (p.is_causal || p.window_left >= 0 || p.window_right >= 0) ? 1 : 0;
    da.bias = (p.alibi_slopes != nullptr || p.softcap > 0.f) ? 1 : 0;
    da.cu_q = p.cu_seqlens_q;                         // (NULL except for the routes of the varlen op)
    da.n_splits = decode_num_partials(p);             // (token-major kernel: grid splits x key sub-ranges; else the grid's y)
    da.ksub = gemv_tm_applicable(p) ? gemv_tm_ksub(p) : 1;
    da.page_shift = -1;
    if (p.block_table && (p.page_block_size & (p.page_block_size - 1)) == 0) {
        int sft = 0; while ((1 << sft) < p.page_block_size) ++sft;
        da.page_shift = sft;
    }
    da.o_partial = nullptr; da.lse_partial = nullptr;
    if (da.n_splits > 1) {
        if (!ws) return -1;
        const size_t rows = (size_t)p.batch * p.nheads_q * p.seqlen_q;
        da.o_partial = reinterpret_cast<float*>(ws);
        da.lse_partial = da.o_partial + (size_t)da.n_splits * rows * p.head_dim;
    }
    const bool bf = p.dtype == FA_BF16;
    switch (p.head_dim) {
        case 64:  return bf ? launch_decode_td<bf16_tag, 64>(da, stream) : launch_decode_td<fp16_tag, 64>(da, stream);
        case 128: return bf ? launch_decode_td<bf16_tag, 128>(da, stream) : launch_decode_td<fp16_tag, 128>(da, stream);
        case 256: return bf ? launch_decode_td<bf16_tag, 256>(da, stream) : launch_decode_td<fp16_tag, 256>(da, stream);
        default:  return -2;
    }
}

}  // namespace fa
