// fa_fwd.hip - fused attention forward for gfx950 (dense / varlen / paged / kv-cache prefill).
//
// Replaces the reference's three forward kernels
//   kernel/fused_mha_forward.cu:25-224, kernel/fused_mha_forward_varlen.cu:25-275,
//   kernel/fused_mha_forward_kvcache.cu:24-295 (attention part)
// with ONE CDNA4 design:
//   * workgroup = 4 waves, 128 query rows (32 per wave), KV tile = 64 keys;
//   * S^T = K Q^T on v_mfma_f32_32x32x16 ("swapped" so a lane owns ONE query row: the
//     online-softmax max/sum are lane-local + one v_permlane32_swap, no LDS for S/P);
//   * the S^T accumulator registers are packed to 16-bit and fed straight back as the B
//     operand of O^T = V^T P^T (the key index is only a contraction index, so the MFMA
//     C-layout key order is used as-is: no permute, no LDS round trip);
//   * V^T comes from LDS through ds_read_b64_tr_b16 (hardware 4x4 transpose), K through
//     XOR-swizzled ds_read_b128; both tiles are double buffered, buffer_load of tile t+1 is
//     issued before the MFMAs of tile t and written to LDS after them (one barrier per tile);
//     rows past the sequence end come back as zeros from the buffer range check;
//   * the running max is only raised when a tile exceeds it by more than 2^8 ("deferred
//     rescale"), so most tiles skip the O-accumulator rescale;
//   * running max / sum and the O accumulator never leave registers.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "fa_common.h"
#include "fa_rope.h"

namespace fa {

constexpr int FWD_BM = 128;
#ifndef FA_FWD_BN
#define FA_FWD_BN 64
#endif
#ifndef FA_FWD_PFK
#define FA_FWD_PFK 4                                   // K fragments in flight ahead of their MFMA (0: compiler's order)
#endif
#ifndef FA_FWD_PFV
#define FA_FWD_PFV 3                                   // V fragments (two transposing reads each) in flight
#endif
#ifndef FA_FWD_OCC
#define FA_FWD_OCC 2
#endif
constexpr int FWD_BN = FA_FWD_BN;
constexpr int FWD_NKB = FWD_BN / 32;               // 32-key blocks per tile
constexpr int FWD_THREADS = 256;
constexpr float FWD_RESCALE_THR = 8.0f;                // log2 units

template <int D> struct FwdSmem {
    static constexpr int TILE = FWD_BN * D * 2;        // bytes of one K (or V) tile
    static constexpr int STAGE = 2 * TILE;             // K + V
    static constexpr int TOTAL = 2 * STAGE;            // double buffered
};

// BIAS: 0 none, 1 general (ALiBi and/or softcap per element), 2 causal ALiBi through the matrix pipe,
//       3 softcap only (constants folded: exp2, add, rcp, fma per element)
// KV8: the K / V rows are fp8-e4m3 (KV cache): a tile is fetched to registers (16 codes per chunk), dequantised with the
//      packed converts of fa_common.h and written to the same LDS images - ONCE per 128 query rows (chunked prefill over an
//      fp8 cache); `k_descale` folds into the softmax scale, `v_descale` into the final normalisation.
// DV:  columns that can be non-zero (head dims 129 .. 192 run on the 256-wide LDS images, 65 .. 96 on the 128-wide ones - the missing
//      columns are read as zeros - but skip the k-steps of S and the accumulator blocks of O that would only see them)
template <typename T, int D, int BIAS, bool PAGED, bool DROPOUT, bool KV8 = false, int DV = D>
__global__ void __launch_bounds__(FWD_THREADS, (D > 128 ? 1 : FA_FWD_OCC)) fa_fwd_kernel(const KArgs a) {
    using E = Elem<T>;
    static_assert(DV == D || (D == 256 && DV == 192) || (D == 128 && DV == 96), "narrow forms: 192 of 256, 96 of 128 columns");
    constexpr int KSTEPS = DV / 16;
    constexpr int DBLKS = DV / 32;
    constexpr int CPR = D / 8;                          // 16-B chunks per row
    constexpr int CHUNKS = FWD_BN * CPR / FWD_THREADS;  // chunks per thread per tile
    constexpr int TILE = FwdSmem<D>::TILE;
    constexpr int STAGE = FwdSmem<D>::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const fa_params& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const WorkItem w = a.flat_blocks
        ? decode_work_flat(blockIdx.x, a.flat_blocks, FWD_BM, p.batch, p.nheads_q, p.nheads_k, p.cu_seqlens_q, lane)
        : decode_work(blockIdx.x, p.batch, p.nheads_q, p.nheads_k, a.n_qblocks);
    if (!w.valid) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int l31 = lane & 31;
    const int g = lane >> 5;

    // ---- per-sequence geometry --------------------------------------------------------
    int seqlen_q = p.seqlen_q, seqlen_k = p.seqlen_k;
    int64_t q_row0 = 0, k_row0 = 0;                    // row offsets (varlen packing / leftpad)
    int kv_b = w.b;
    if (p.cu_seqlens_q) {
        q_row0 = p.cu_seqlens_q[w.b];
        seqlen_q = p.cu_seqlens_q[w.b + 1] - (int)q_row0;
        if (a.skip_short_q > 0 && seqlen_q <= a.skip_short_q) return;      // (mixed batch: the decode kernels own this sequence)
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
    } else if (a.kv_mode) {
        seqlen_k = a.seqlen_k_add;                                                      // kv cache without cache_seqlens
    }
    if (a.kv_batch_idx) kv_b = a.kv_batch_idx[w.b];
    if (a.leftpad_k) k_row0 += a.leftpad_k[w.b];
    // workgroup-uniform by construction (everything above hangs off w.b) - said explicitly, so that the per-tile address and
    // descriptor arithmetic of the paged path stays on the scalar unit: without it ~100 VALU instructions and five
    // v_readfirstlane sat between the barrier and every tile's DMA issue (paged prefill 15 % behind a contiguous cache
    // even with ONE page per sequence, tools: page = whole cache)
    seqlen_k = __builtin_amdgcn_readfirstlane(seqlen_k);
    seqlen_q = __builtin_amdgcn_readfirstlane(seqlen_q);
    kv_b = __builtin_amdgcn_readfirstlane(kv_b);
    k_row0 = (int64_t)__builtin_amdgcn_readfirstlane((int)k_row0);
    q_row0 = (int64_t)__builtin_amdgcn_readfirstlane((int)q_row0);

    const int off = seqlen_k - seqlen_q;               // bottom-right alignment
    const int wl = p.window_left;
    const int wr = p.is_causal ? 0 : p.window_right;   // causal == window_right 0 (include/mat_mul.h:92,103)
    DropCtx dc = {0, 0, 0, 0};
    if (DROPOUT) {
        dc.k0 = (uint32_t)p.philox_seed; dc.k1 = (uint32_t)(p.philox_seed >> 32);
        dc.thr = a.drop_thr; dc.offset = p.philox_offset;
    }
    const uint64_t drop_n_glob = (uint64_t)p.seqlen_k;        // dense: S_k; varlen: max_seqlen_k
    // ---- the workgroup's passes: with a.pair_qblocks (causal load balance) q-block qb (heavy) and then its mirror
    //      n_qblocks-1-qb (light), so that every workgroup carries the same number of KV tiles; otherwise one block.
    //      (A workgroup walking a RUN of consecutive blocks with the tile stream continuing across the seams was built for
    //      BASELINE config 3 in round 4 and measured slower at every run length: tools/experiments/fwd_d64_round4_experiments.patch)
    const int n_pass = (a.pair_qblocks && (a.n_qblocks_total - 1 - w.qb) != w.qb) ? 2 : 1;
    auto qb_of = [&](int pass) { return pass == 0 ? w.qb : a.n_qblocks_total - 1 - w.qb; };
    // key-tile range of the 128-row block at row m0
    auto tile_range = [&](int m0, int& t_min, int& t_max) {
        t_min = 0; t_max = (seqlen_k + FWD_BN - 1) / FWD_BN;
        const int m_last = (m0 + FWD_BM < seqlen_q ? m0 + FWD_BM : seqlen_q) - 1;
        if (wr >= 0) {
            const int kmax = m_last + off + wr;
            const int t = kmax < 0 ? 0 : kmax / FWD_BN + 1;
            t_max = t < t_max ? t : t_max;
        }
        if (wl >= 0) {
            const int kmin = m0 + off - wl;
            if (kmin > 0) t_min = kmin / FWD_BN;
        }
        if (m0 >= seqlen_q) t_max = t_min;                                   // (no rows: no tiles)
    };
    // per-pass state (set by begin_pass)
    int m_block = 0, n_min = 0, n_max = 0;
    int wave_row0 = 0, my_row = 0;                     // this wave's first row, this lane's query row
    int lo = 0, hi = -1;                               // visible keys of my row: lo <= j <= hi
    int w_hi_min = 0, w_hi_max = 0, w_lo_max = 0, w_lo_min = 0;   // wave-uniform bounds for tile skipping / mask elision
    auto begin_pass = [&](int pass) {
        m_block = qb_of(pass) * FWD_BM;
        tile_range(m_block, n_min, n_max);
        wave_row0 = m_block + wave * 32;
        my_row = wave_row0 + l31;
        lo = 0; hi = seqlen_k - 1;
        if (wr >= 0) { const int h2 = my_row + off + wr; hi = h2 < hi ? h2 : hi; }
        if (wl >= 0) { const int l2 = my_row + off - wl; lo = l2 > lo ? l2 : lo; }
        const int wrow_last = wave_row0 + 31;
        w_hi_min = seqlen_k - 1; w_hi_max = seqlen_k - 1; w_lo_max = 0;
        if (wr >= 0) {
            const int h0 = wave_row0 + off + wr, h1 = wrow_last + off + wr;
            w_hi_min = h0 < w_hi_min ? h0 : w_hi_min;
            w_hi_max = h1 < w_hi_max ? h1 : w_hi_max;
        }
        if (wl >= 0) { const int l1 = wrow_last + off - wl; w_lo_max = l1 > 0 ? l1 : 0; }
        w_lo_min = (wl >= 0 && wave_row0 + off - wl > 0) ? wave_row0 + off - wl : 0;
        // a wave whose 32 rows all lie past the sequence (the tail of a sequence's last block) computes nothing: no tile is
        // "active" for it (it still takes part in the loads and barriers).  Its rows are not stored.
        if (wave_row0 >= seqlen_q) { w_hi_max = -1; w_hi_min = -1; }
    };

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
    // fp8 cache: byte pointers (strides are in elements = bytes)
    const uint8_t* kp8 = nullptr; const uint8_t* vp8 = nullptr;
    if (KV8) {
        kp8 = reinterpret_cast<const uint8_t*>(p.k) + (int64_t)w.hk * p.k_head_stride;
        vp8 = reinterpret_cast<const uint8_t*>(p.v) + (int64_t)w.hk * p.v_head_stride;
        if (!PAGED) {
            kp8 += (int64_t)kv_b * p.k_batch_stride + k_row0 * p.k_row_stride;
            vp8 += (int64_t)kv_b * p.v_batch_stride + k_row0 * p.v_row_stride;
        }
    }

    float slope = 0.f;                                 // ALiBi slope (natural-log score units)
    if (BIAS && p.alibi_slopes) slope = p.alibi_slopes[w.b * p.alibi_batch_stride + w.h];

    const int dv = valid_cols(p);
    // ---- Q fragments: B operand of S^T = K Q^T, lane holds Q[row][16ks + 8g .. +7] --------
    // Through a buffer descriptor over the sequence's rows of this head: rows past the sequence and columns past the valid
    // width come back as zeros from the range check, and the compiler counts the loads (vmcnt).
    const __amdgpu_buffer_rsrc_t q_rsrc = make_rsrc(qp, p.q_row_stride, seqlen_q, dv);
    auto load_q = [&](int row, u32x4 (&dst)[KSTEPS]) {
        const uint32_t base = (uint32_t)row * (uint32_t)(p.q_row_stride * 2) + 16 * g;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
            dst[ks] = buf_load_b128(q_rsrc, (row < seqlen_q && 16 * ks + 8 * g < dv) ? base + 32 * ks : kOobVoff, 0);
    };
    u32x4 qf[KSTEPS];
    auto rope_q = [&]() {
        if (a.rope_q && my_row < seqlen_q) {
            // kv-cache call with rotary tables: rotate the row in registers (no rotated-Q copy, no extra launch).  The
            // partner chunk of the non-interleaved form (d +- rotary_dim / 2) is one more 16-byte load per chunk.
            const int L = p.cache_seqlens ? p.cache_seqlens[w.b] : 0;
            const int lp = p.cache_leftpad ? p.cache_leftpad[w.b] : 0;
            const int local = (p.is_causal || p.window_left >= 0 || p.window_right >= 0) ? 1 : 0;
            const int pos = L + lp + (local ? my_row : 0);                       // include/rotary.h:177,201-202
            if (pos >= 0 && pos < p.seqlen_ro) {
                const int half = p.rotary_dim >> 1;
                const uint16_t* cosp = reinterpret_cast<const uint16_t*>(p.rotary_cos) + (int64_t)pos * half;
                const uint16_t* sinp = reinterpret_cast<const uint16_t*>(p.rotary_sin) + (int64_t)pos * half;
                const uint16_t* qrow0 = qp + (int64_t)my_row * p.q_row_stride;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    const int d_base = 16 * ks + 8 * g;
                    if (d_base < p.rotary_dim && d_base < dv) {
                        u32x4 xp = qf[ks];
                        if (!p.rotary_interleaved) {
                            const int pd = d_base < half ? d_base + half : d_base - half;
                            xp = *reinterpret_cast<const u32x4*>(qrow0 + pd);
                        }
                        rope_chunk<T>(qf[ks], xp, cosp, sinp, d_base, p.rotary_dim, p.rotary_interleaved != 0);
                    }
                }
            }
        }
    };

    // ---- staging ---------------------------------------------------------------------------
    // Dense / varlen: LDS-DMA (buffer_load ... lds).  Wave w issues instructions w*CHUNKS ..+CHUNKS-1
    // of each tile; instruction i covers ROWS_PI consecutive rows (64 lanes x 16 B = 1 KiB), lane l ->
    // row i*ROWS_PI + l / CPR, PHYSICAL 16-byte slot l % CPR, and fetches the logical chunk that the
    // XOR swizzle places there (the swizzle is an involution).
    // Paged K/V: one page lookup per row -> global_load to registers, ds_write after the MFMAs.
    constexpr int ROWS_PI = 64 / CPR;
    constexpr int CPR8 = D / 16;                                        // 16-byte chunks (16 codes) per fp8 row
    constexpr int CH8 = FWD_BN * CPR8 / FWD_THREADS;                    // fp8 chunks per thread and tile
    u32x4 k8reg[KV8 ? CH8 : 1], v8reg[KV8 ? CH8 : 1];
    uint32_t k_voff[CHUNKS], v_voff[CHUNKS];
    int k_lds[CHUNKS], v_lds[CHUNKS];
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        // LDS-DMA lane map (contiguous tiles, and paged tiles through a per-tile descriptor): lane-linear destination,
        // swizzle applied to the source column
        const int inst = wave * CHUNKS + i;
        const int row = inst * ROWS_PI + lane / CPR;
        const int slot = lane % CPR;
        const int k_cb = swz_row_off<D>(row, slot * 16) - row * D * 2;      // logical byte column
        const int v_cb = swzt_row_off<D>(row, slot * 16) - row * D * 2;
        k_voff[i] = k_cb < dv * 2 ? (uint32_t)(row * p.k_row_stride * 2 + k_cb) : kOobVoff;
        v_voff[i] = v_cb < dv * 2 ? (uint32_t)(row * p.v_row_stride * 2 + v_cb) : kOobVoff;
        k_lds[i] = inst * 1024;                                             // wave-uniform destination
        v_lds[i] = TILE + inst * 1024;
    }
    // Paged K/V, aligned case (a 64-key tile never straddles a page: page % 64 == 0 by contract and the left
    // pad is a multiple of 64): the same LDS-DMA as the contiguous path through a per-tile descriptor built
    // from ONE block-table lookup; otherwise the per-row path in store_tile.
    int page_shift = -1;
    if (PAGED && (p.page_block_size & (p.page_block_size - 1)) == 0) page_shift = __builtin_ctz(p.page_block_size);
    const bool paged_aligned = PAGED && (k_row0 % FWD_BN == 0) && (p.page_block_size % FWD_BN == 0);      // (fp8 path)
    // 16-bit path: a wave's LDS-DMA instructions cover rows 16 w .. 16 w + 15 of the tile (at every head dim), so pages of
    // 16 tokens and more work with ONE block-table entry per wave and tile: each wave builds its own descriptor
    // (power-of-two pages: the page index is a shift and the row in the page a mask; other page sizes - 48, 192, 320 ... - pay
    //  an inline integer division, ~30 scalar instructions per tile, on the same path)
    const bool paged_dma = PAGED && !KV8 && (k_row0 % 16 == 0) && (p.page_block_size % 16 == 0);
    static_assert(FWD_BN != 64 || (CHUNKS * (64 / (D / 8))) == 16, "a wave's DMA instructions span one 16-row quarter of a 64-key tile");
    // uniform (SGPR) copies of the K / V base pointers of this (batch, kv-head) for the per-tile descriptors
    auto uniform_ptr = [](const void* ptr) {
        const uint64_t b = reinterpret_cast<uint64_t>(ptr);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);          // (the builtin returns int: no sign extension)
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        return ((uint64_t)hi << 32) | lo;
    };
    const uint64_t kp_u = uniform_ptr(kp), vp_u = uniform_ptr(vp);
    // this wave's first row in a tile - or 0 when a whole tile lies in one page (one descriptor for the four waves, as before)
    const int pf_row = (PAGED && !KV8 && FWD_BN == 64 && (p.page_block_size % FWD_BN != 0 || k_row0 % FWD_BN != 0)) ? 16 * wave : 0;
    const __amdgpu_buffer_rsrc_t k_rsrc = make_rsrc(kp, p.k_row_stride, PAGED ? 0 : seqlen_k, dv);
    const __amdgpu_buffer_rsrc_t v_rsrc = make_rsrc(vp, p.v_row_stride, PAGED ? 0 : seqlen_k, dv);
    const uint32_t k_tile_bytes = (uint32_t)(FWD_BN * p.k_row_stride * 2);
    const uint32_t v_tile_bytes = (uint32_t)(FWD_BN * p.v_row_stride * 2);

    // paged_dma: block-table entry of the next tile to load, requested one tile ahead through the CONSTANT address space:
    // a uniform constant load is a scalar load (s_load_dword) whose wait the compiler places at the first use.  As a plain
    // C++ load it became a vector load + v_readfirstlane with an `s_waitcnt vmcnt(0)` right behind it - i.e. behind the
    // eight LDS-DMA loads just issued, so every tile's compute started only after the next tile had landed (chunked
    // prefill over a paged cache at 0.65 x a contiguous one, tools/chunked_prefill_probe.py).  The kernel never writes
    // the block table.
    typedef const int32_t __attribute__((address_space(4))) * const_i32_ptr;
    const const_i32_ptr btab_c = (const_i32_ptr)(uintptr_t)btab;
    int pf_phys = 0;
    auto pf_request = [&](int nb1) {
        int pos1 = nb1 * FWD_BN + pf_row;
        pos1 = (pos1 < seqlen_k ? pos1 : (seqlen_k > 0 ? seqlen_k - 1 : 0)) + (int)k_row0;      // (a quarter past the sequence: any valid entry - its rows are out of range)
        pf_phys = btab_c[page_shift >= 0 ? (pos1 >> page_shift) : pos1 / p.page_block_size];
    };
    // fp8 path with small pages: one entry per (wave, chunk step)
    const bool paged_q16 = PAGED && KV8 && !paged_aligned && (k_row0 % 16 == 0) && (p.page_block_size % 16 == 0);
    int pf8[KV8 ? CH8 : 1] = {};
    auto pf8_request = [&](int nb1) {
#pragma unroll
        for (int i = 0; i < (KV8 ? CH8 : 1); ++i) {
            int pos1 = nb1 * FWD_BN + (((64 * wave + i * FWD_THREADS) / CPR8) & ~15);
            pos1 = (pos1 < seqlen_k ? pos1 : (seqlen_k > 0 ? seqlen_k - 1 : 0)) + (int)k_row0;
            pf8[i] = btab_c[page_shift >= 0 ? (pos1 >> page_shift) : pos1 / p.page_block_size];
        }
    };
    // issue the loads of tile nb; for the DMA path they land directly in LDS stage `stage`
    auto load_tile = [&](int nb, auto stage_c) {
        constexpr int stage = decltype(stage_c)::value;
        if constexpr (KV8) {
            const int n0 = nb * FWD_BN;
            if (PAGED && paged_aligned) {
                // aligned tiles lie inside one page: ONE entry, requested one tile ago (scalar), no per-lane lookups in
                // front of the loads
                const int pos0 = n0 + (int)k_row0;
                const int pg0 = page_shift >= 0 ? (pos0 >> page_shift) : pos0 / p.page_block_size;
                const int64_t phys = pf_phys;
                const uint8_t* kpg = kp8 + phys * p.k_batch_stride + (int64_t)(pos0 - pg0 * p.page_block_size) * p.k_row_stride;
                const uint8_t* vpg = vp8 + phys * p.v_batch_stride + (int64_t)(pos0 - pg0 * p.page_block_size) * p.v_row_stride;
#pragma unroll
                for (int i = 0; i < CH8; ++i) {
                    const int c8 = tid + i * FWD_THREADS;
                    const int row = c8 / CPR8, cc8 = c8 % CPR8;
                    u32x4 z = {0, 0, 0, 0};
                    k8reg[i] = z; v8reg[i] = z;
                    if (n0 + row < seqlen_k && cc8 * 16 < dv) {
                        k8reg[i] = *reinterpret_cast<const u32x4*>(kpg + (int64_t)row * p.k_row_stride + cc8 * 16);
                        v8reg[i] = *reinterpret_cast<const u32x4*>(vpg + (int64_t)row * p.v_row_stride + cc8 * 16);
                    }
                }
                pf_request(nb + 1 < n_max ? nb + 1 : nb);
                return;
            }
            if (PAGED && paged_q16) {
                // pages of 16 tokens and more (not a multiple of the tile): the rows a wave fetches with its chunk i lie in one
                // 16-row quarter of the tile - ONE entry per (wave, i), requested one tile ago (scalars)
#pragma unroll
                for (int i = 0; i < CH8; ++i) {
                    const int c8 = tid + i * FWD_THREADS;
                    const int row = c8 / CPR8, cc8 = c8 % CPR8;
                    const int q16 = ((64 * wave + i * FWD_THREADS) / CPR8) & ~15;       // wave-uniform
                    const int pos = n0 + (int)k_row0 + q16;
                    const int pg = page_shift >= 0 ? (pos >> page_shift) : pos / p.page_block_size;
                    const int64_t phys = pf8[i];
                    const uint8_t* kpg = kp8 + phys * p.k_batch_stride + (int64_t)(pos - pg * p.page_block_size - q16) * p.k_row_stride;
                    const uint8_t* vpg = vp8 + phys * p.v_batch_stride + (int64_t)(pos - pg * p.page_block_size - q16) * p.v_row_stride;
                    u32x4 z = {0, 0, 0, 0};
                    k8reg[i] = z; v8reg[i] = z;
                    if (n0 + row < seqlen_k && cc8 * 16 < dv) {
                        k8reg[i] = *reinterpret_cast<const u32x4*>(kpg + (int64_t)row * p.k_row_stride + cc8 * 16);
                        v8reg[i] = *reinterpret_cast<const u32x4*>(vpg + (int64_t)row * p.v_row_stride + cc8 * 16);
                    }
                }
                pf8_request(nb + 1 < n_max ? nb + 1 : nb);
                return;
            }
#pragma unroll
            for (int i = 0; i < CH8; ++i) {
                const int c8 = tid + i * FWD_THREADS;
                const int row = c8 / CPR8, cc8 = c8 % CPR8;
                const int j = n0 + row;
                u32x4 z = {0, 0, 0, 0};
                k8reg[i] = z; v8reg[i] = z;
                if (j < seqlen_k && cc8 * 16 < dv) {
                    if (PAGED) {
                        const int pos = j + (int)k_row0;
                        const int pg = page_shift >= 0 ? (pos >> page_shift) : pos / p.page_block_size;
                        const int pr = pos - pg * p.page_block_size;
                        const int64_t phys = btab[pg];
                        k8reg[i] = *reinterpret_cast<const u32x4*>(kp8 + phys * p.k_batch_stride + (int64_t)pr * p.k_row_stride + cc8 * 16);
                        v8reg[i] = *reinterpret_cast<const u32x4*>(vp8 + phys * p.v_batch_stride + (int64_t)pr * p.v_row_stride + cc8 * 16);
                    } else {
                        k8reg[i] = *reinterpret_cast<const u32x4*>(kp8 + (int64_t)j * p.k_row_stride + cc8 * 16);
                        v8reg[i] = *reinterpret_cast<const u32x4*>(vp8 + (int64_t)j * p.v_row_stride + cc8 * 16);
                    }
                }
            }
            return;
        }
        if (PAGED && paged_dma) {
            const int n0 = nb * FWD_BN;
            const int pos0 = n0 + (int)k_row0 + pf_row;                       // this wave's quarter of the tile
            // row in the page (the lane offsets count rows from the tile's first row); pages that are not a power of two
            // (192, 320 ...) pay an integer division per tile here and in pf_request - still one descriptor per wave and tile
            const int pr = (page_shift >= 0 ? (pos0 & (p.page_block_size - 1)) : pos0 % p.page_block_size) - pf_row;
            // the block-table entry of THIS tile was looked up one tile ago (pf_phys): with the lookup at the top of the
            // step its round trip sat in front of every tile's DMA issue (chunked prefill over a paged cache ran 12-33 %
            // behind a contiguous one: tools/chunked_prefill_probe.py); the entry of the next tile is requested below
            const int64_t phys = pf_phys;
            int rows = seqlen_k - n0;
            rows = rows < 0 ? 0 : (rows > FWD_BN ? FWD_BN : rows);
            // descriptors on the scalar unit: uniform base pointers (hoisted), one 64-bit multiply-add per tensor, 32-bit extent
            const uint64_t ka = kp_u + (uint64_t)((phys * p.k_batch_stride + (int64_t)pr * p.k_row_stride) * 2);
            const uint64_t va = vp_u + (uint64_t)((phys * p.v_batch_stride + (int64_t)pr * p.v_row_stride) * 2);
            const int kext = ((rows - 1) * (int)p.k_row_stride + dv) * 2, vext = ((rows - 1) * (int)p.v_row_stride + dv) * 2;
            const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(ka), 0, rows > 0 ? kext : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(va), 0, rows > 0 ? vext : 0, 0x00020000);
            char* base = smem + stage * STAGE;
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(kr, base + k_lds[i], k_voff[i], 0);
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(vr, base + v_lds[i], v_voff[i], 0);
            pf_request(nb + 1 < n_max ? nb + 1 : nb);                         // (the tiles of a pass are walked upwards)
        } else if (PAGED) {
            // unaligned paged tiles (a left pad that is not a multiple of the tile): fetched row by row in store_tile
        } else {
            char* base = smem + stage * STAGE;
            const uint32_t ks_off = (uint32_t)nb * k_tile_bytes;
            const uint32_t vs_off = (uint32_t)nb * v_tile_bytes;
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(k_rsrc, base + k_lds[i], k_voff[i], ks_off);
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) buf_load_lds_b128(v_rsrc, base + v_lds[i], v_voff[i], vs_off);
        }
    };
    auto store_tile = [&](auto stage_c, int nb) {
        constexpr int stage = decltype(stage_c)::value;
        if constexpr (KV8) {
            char* base = smem + stage * STAGE;
#pragma unroll
            for (int i = 0; i < CH8; ++i) {
                const int c8 = tid + i * FWD_THREADS;
                const int row = c8 / CPR8, cc8 = c8 % CPR8;
                u32x4 l0, h0, l1, h1;
                fp8x16_to_16bit<T>(k8reg[i], l0, h0);
                fp8x16_to_16bit<T>(v8reg[i], l1, h1);
                lds_write_b128(base + swz_row_off<D>(row, cc8 * 32), l0);
                lds_write_b128(base + swz_row_off<D>(row, cc8 * 32 + 16), h0);
                lds_write_b128(base + TILE + swzt_row_off<D>(row, cc8 * 32), l1);
                lds_write_b128(base + TILE + swzt_row_off<D>(row, cc8 * 32 + 16), h1);
            }
            return;
        }
        if (PAGED && !paged_dma) {
            // rare (a tile may straddle pages): per-row lookup, load and LDS write back to back - nothing stays in
            // registers across the tile's compute (as loop-carried register arrays these 64 registers spilled the
            // ALIGNED path's loop: 60 scratch instructions, chunked prefill over a paged cache at 2/3 of a contiguous one)
            char* base = smem + stage * STAGE;
            const int n0 = nb * FWD_BN;
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) {
                const int c = tid + i * FWD_THREADS;
                const int row = c / CPR, cc = c % CPR;
                const int j = n0 + row;
                u32x4 kx = {0, 0, 0, 0}, vx = {0, 0, 0, 0};
                if (j < seqlen_k && cc * 8 < dv) {
                    const int pos = j + (int)k_row0;
                    const int pg = pos / p.page_block_size;
                    const int pr = pos - pg * p.page_block_size;
                    const int64_t phys = btab[pg];
                    kx = *reinterpret_cast<const u32x4*>(kp + phys * p.k_batch_stride + (int64_t)pr * p.k_row_stride + cc * 8);
                    vx = *reinterpret_cast<const u32x4*>(vp + phys * p.v_batch_stride + (int64_t)pr * p.v_row_stride + cc * 8);
                }
                lds_write_b128(base + swz_row_off<D>(row, cc * 16), kx);
                lds_write_b128(base + TILE + swzt_row_off<D>(row, cc * 16), vx);
            }
        }
    };

    // ---- accumulators -------------------------------------------------------------------------
    f32x16 oacc[DBLKS];
#pragma unroll
    for (int d = 0; d < DBLKS; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY;     // running (deferred) max, log2 domain (scaled)
    float l_run = 0.f;           // this lane's partial row sum (its 32 keys per tile)
    // lane-constant LDS read offsets
    int k_rd[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) k_rd[ks] = swz_row_off<D>(l31, 32 * ks + 16 * g);
    const int v_rr = (lane & 15) >> 2;                                  // row within a 4-row transpose group
    const int v_cb = (((lane >> 4) & 1) << 5) + ((lane & 3) << 3);       // byte column within a 64-byte d-block
#if FA_FWD_PFK > 0
    // lane-constant read addresses, pinned (stage, key block and 16-key step are immediate offsets: the swizzles
    // do not depend on row bits >= 4)
    const lds_char* k_ptr[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) k_ptr[ks] = lds_pin(smem + k_rd[ks]);
    const lds_char* v_ptr[2][DBLKS];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int d = 0; d < DBLKS; ++d) v_ptr[h][d] = lds_pin(smem + TILE + swzt_row_off<D>(4 * g + v_rr + 8 * h, d * 64 + v_cb));
#endif
    // ALiBi fast path ("rank-2 update"): when every visible key is at or left of the diagonal
    // (causal / window_right == 0) and there is no softcap, the bias -slope (i + off - j) is linear
    // in the key position j = n0 + pos.  Its row- and tile-constant part slope (n0 - i - off) only
    // shifts the row maximum (`shift`, one VALU per tile); the pos-dependent part slope * pos is
    // added by the matrix pipe: one extra MFMA per 32-key block whose A operand holds pos (exact
    // in 16 bits) in contraction slots 0 and 1 and whose B operand holds slope / softmax_scale
    // split into a 16-bit head and tail.  No per-element VALU work is left, versus 5 per element
    // on the general path below (measured at BASELINE config 5's shard: 652 -> see DESIGN.md).
    constexpr bool lin = BIAS == 2;        // host guarantees: slopes given, no softcap, wr == 0
    const float c = ((BIAS == 1) || (BIAS == 3)) ? 1.0f : a.scale_log2e * (KV8 ? p.k_descale : 1.0f);
    const float slope2 = slope * kLog2e;
    u32x4 pos_a[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    u32x4 slope_b = {0, 0, 0, 0};
    if (BIAS && lin && g == 0) {
        const float sv = slope / p.softmax_scale;
        const float head = E::lo(E::pack2(sv, 0.f));
        slope_b[0] = E::pack2(head, sv - head);
        pos_a[0][0] = E::pack2((float)l31, (float)l31);
        pos_a[1][0] = E::pack2((float)(32 + l31), (float)(32 + l31));
    }

    // one KV tile for this wave; STAGE is a compile-time constant (all LDS offsets immediates)
    auto compute_tile = [&](auto stage_c, int nb) {
        constexpr int stage = decltype(stage_c)::value;
        const int n0 = nb * FWD_BN;
        const char* sbase = smem + stage * STAGE;
        // ---- S^T = K Q^T : sacc[kb][r] = S[my_row][n0 + 32 kb + row(r, g)] ----
        f32x16 sacc[FWD_NKB];
#pragma unroll
        for (int kb = 0; kb < FWD_NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
        __builtin_amdgcn_s_setprio(1);
#if FA_FWD_PFK > 0
        {
            // hipcc's scheduler, left alone, serialises read -> wait -> MFMA through ONE temporary (every
            // MFMA eats a full LDS round trip).  Fences pin the K fragment of MFMA i + PFK ahead of MFMA i.
            constexpr int NQK = KSTEPS * FWD_NKB;
            u32x4 kk[NQK];
            auto kread = [&](int i) { return lds_read_b128(k_ptr[i / FWD_NKB] + (stage * STAGE + (i % FWD_NKB) * 32 * D * 2)); };
#pragma unroll
            for (int i = 0; i < FA_FWD_PFK && i < NQK; ++i) kk[i] = kread(i);
#pragma unroll
            for (int i = 0; i < NQK; ++i) {
                if (i + FA_FWD_PFK < NQK) kk[i + FA_FWD_PFK] = kread(i + FA_FWD_PFK);
                __builtin_amdgcn_sched_barrier(0);
                sacc[i % FWD_NKB] = E::mfma(kk[i], qf[i / FWD_NKB], sacc[i % FWD_NKB]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            u32x4 kk[FWD_NKB];
#pragma unroll
            for (int kb = 0; kb < FWD_NKB; ++kb) kk[kb] = lds_read_b128(sbase + k_rd[ks] + kb * 32 * D * 2);
#pragma unroll
            for (int kb = 0; kb < FWD_NKB; ++kb) sacc[kb] = E::mfma(kk[kb], qf[ks], sacc[kb]);
        }
#endif
        __builtin_amdgcn_s_setprio(0);
        // ---- bias / softcap (rare variants), then masking on edge tiles ----
        float shift = 0.f;
        if (BIAS && lin) {
#pragma unroll
            for (int kb = 0; kb < FWD_NKB; ++kb) sacc[kb] = E::mfma(pos_a[kb], slope_b, sacc[kb]);
            shift = slope2 * (float)(n0 - my_row - off);
        }
        if (BIAS == 3) {
            // cap * tanh(s scale / cap) in log2 units:  cap2 (1 - 2 / (1 + exp2(s k1))),  k1 = 2 scale log2e / cap
            const float k1 = 2.0f * a.scale_log2e / p.softcap, cap2 = p.softcap * kLog2e;
#pragma unroll
            for (int kb = 0; kb < FWD_NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float rr1 = fast_rcp(1.0f + fast_exp2(sacc[kb][r] * k1));
                    sacc[kb][r] = fmaf(rr1, -2.0f * cap2, cap2);
                }
        }
        if (BIAS == 1) {
            const float cap = p.softcap;
            const float rcap = cap > 0.f ? 1.0f / cap : 0.f;
#pragma unroll
            for (int kb = 0; kb < FWD_NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = n0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                    float s = sacc[kb][r] * (p.softmax_scale * (KV8 ? p.k_descale : 1.0f));
                    const int dist = my_row + off - j;
                    s = fmaf(-slope, fabsf((float)dist), s);
                    if (cap > 0.f) s = cap * fast_tanh(s * rcap);
                    sacc[kb][r] = s * kLog2e;
                }
        }
        const bool need_mask = (n0 + FWD_BN - 1 > w_hi_min) || (n0 < w_lo_max);
        if (need_mask) {
            // the key of register (kb, r) is n0 + 4 g + c with c a compile-time constant: the lane builds the visibility
            // bits of its row for this tile once (bit c set <=> lo <= n0 + 4 g + c <= hi; one 32-bit word per 32-key
            // block), and every element costs a sign-extending bit extract and a three-input bit op (v_bfe_i32,
            // v_bitop3_b32) - no compare, no VCC, none of the wait states hipcc puts between v_cmp and v_cndmask
            // (4 issue slots per element before, 2 now)
            const int lc = lo - n0 - 4 * g;                                   // visible positions [lc, hc)
            const int hc = hi < lo ? lc : hi - n0 - 4 * g + 1;
            uint32_t visw[FWD_NKB];
#pragma unroll
            for (int kb = 0; kb < FWD_NKB; ++kb) {
                int l = lc - 32 * kb, h = hc - 32 * kb;
                l = l < 0 ? 0 : (l > 32 ? 32 : l);
                h = h < 0 ? 0 : (h > 32 ? 32 : h);
                const uint32_t below_h = h >= 32 ? ~0u : ((1u << h) - 1u);
                const uint32_t below_l = l >= 32 ? ~0u : ((1u << l) - 1u);
                visw[kb] = below_h & ~below_l;
            }
#pragma unroll
            for (int kb = 0; kb < FWD_NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cpos = (r & 3) + 8 * (r >> 2);
                    const uint32_t m = (uint32_t)((int32_t)(visw[kb] << (31 - cpos)) >> 31);
                    sacc[kb][r] = select_bits(sacc[kb][r], m, 0xff800000u);
                }
        }
        // ---- online softmax (log2 domain) with deferred rescale ----
        // (v_max3_f32 through asm: one instruction per two elements, and none of the canonicalising v_max hipcc puts in
        //  front of fmaxf on MFMA outputs; one chain per 32-key block)
        float mxb[FWD_NKB];
#pragma unroll
        for (int kb = 0; kb < FWD_NKB; ++kb) {
            float m3 = max3_f32(sacc[kb][0], sacc[kb][1], sacc[kb][2]);
#pragma unroll
            for (int r = 3; r + 1 < 16; r += 2) m3 = max3_f32(m3, sacc[kb][r], sacc[kb][r + 1]);
            mxb[kb] = m3;                                               // (r = 15 joins below)
        }
        float mx = max3_f32(mxb[0], sacc[0][15], FWD_NKB > 1 ? sacc[FWD_NKB - 1][15] : sacc[0][15]);
#pragma unroll
        for (int kb = 1; kb < FWD_NKB; ++kb) mx = max3_f32(mx, mxb[kb], kb + 1 < FWD_NKB ? sacc[kb][15] : mxb[kb]);
        mx = xhalf_max(mx) * c;
        if (BIAS) mx += shift;
        // keep the old max unless some row of the wave would exceed it by > 2^THR
        // (NaN-safe: -inf - -inf compares false -> takes the rescale path)
        if (!__all(mx - m_run <= FWD_RESCALE_THR)) {
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
        const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
        const float ms = BIAS ? m_use - shift : m_use;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < FWD_NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = fast_exp2(fmaf(sacc[kb][r], c, -ms));
                sacc[kb][r] = e;
                psum += e;
            }
        l_run += psum;                                      // PRE-dropout sum (include/softmax.h:187)
        if (DROPOUT) {
            const uint64_t row_g = (uint64_t)(q_row0 + my_row);
            uint16_t* dm = nullptr;
            if (p.dmask && my_row < seqlen_q) {
                dm = reinterpret_cast<uint16_t*>(p.dmask) +
                     (p.cu_seqlens_q ? ((int64_t)(q_row0 + my_row) * p.nheads_q + w.h) * (int64_t)p.seqlen_k
                                     : (((int64_t)w.b * p.nheads_q + w.h) * seqlen_q + my_row) * (int64_t)seqlen_k);
            }
            const uint16_t one = std::is_same<T, bf16_tag>::value ? 0x3F80 : 0x3C00;
#pragma unroll
            for (int kb = 0; kb < FWD_NKB; ++kb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int j4 = n0 + kb * 32 + 8 * rg + 4 * g;
                    const uint32_t bits = dropout_keep4(dc, row_g * drop_n_glob + (uint64_t)j4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool keep = (bits >> e) & 1u;
                        if (!keep) sacc[kb][4 * rg + e] = 0.f;
                        if (dm && j4 + e < seqlen_k) dm[j4 + e] = keep ? one : (uint16_t)(one | 0x8000);
                    }
                }
        }

        // ---- O^T += V^T P^T : k-step t covers C-layout regs 8 (t&1) .. +7 of sacc[t>>1] ----
#if FA_FWD_PFV > 0
        {
            constexpr int NT = 2 * FWD_NKB, NPV = NT * DBLKS;
            u32x4 pf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int kb = t >> 1, ks2 = t & 1;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf[t][e] = E::pack2(sacc[kb][8 * ks2 + 2 * e], sacc[kb][8 * ks2 + 2 * e + 1]);
            }
            // rows kb*32 + 16 ks2 + 8 hf + 4 g + (0..3) = 16 t + ..., 64-byte column block d
            auto vread = [&](int i) {
                const int t = i / DBLKS, d = i % DBLKS;
                // (asm form + counted wait: the builtin gets an s_waitcnt vmcnt(0) - the DMA of the NEXT tile - in front of it, fa_common.h)
                const u32x2 v0 = lds_read_tr16_nw(v_ptr[0][d], stage * STAGE + 16 * t * D * 2);
                const u32x2 v1 = lds_read_tr16_nw(v_ptr[1][d], stage * STAGE + 16 * t * D * 2);
                return u32x4{v0[0], v0[1], v1[0], v1[1]};
            };
            u32x4 vf[NPV];
#pragma unroll
            for (int i = 0; i < FA_FWD_PFV && i < NPV; ++i) vf[i] = vread(i);
#pragma unroll
            for (int i = 0; i < NPV; ++i) {
                if (i + FA_FWD_PFV < NPV) vf[i + FA_FWD_PFV] = vread(i + FA_FWD_PFV);
                lds_tr_wait(vf[i], 2 * ((NPV - 1 - i) < FA_FWD_PFV ? (NPV - 1 - i) : FA_FWD_PFV));
                __builtin_amdgcn_sched_barrier(0);
                oacc[i % DBLKS] = E::mfma(vf[i], pf[i / DBLKS], oacc[i % DBLKS]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
#pragma unroll
        for (int t = 0; t < 2 * FWD_NKB; ++t) {
            const int kb = t >> 1, ks2 = t & 1;
            u32x4 pf;
            pf[0] = E::pack2(sacc[kb][8 * ks2 + 0], sacc[kb][8 * ks2 + 1]);
            pf[1] = E::pack2(sacc[kb][8 * ks2 + 2], sacc[kb][8 * ks2 + 3]);
            pf[2] = E::pack2(sacc[kb][8 * ks2 + 4], sacc[kb][8 * ks2 + 5]);
            pf[3] = E::pack2(sacc[kb][8 * ks2 + 6], sacc[kb][8 * ks2 + 7]);
#pragma unroll
            for (int d = 0; d < DBLKS; ++d) {
                // rows kb*32 + 16 ks2 + 8 hf + 4 g + (0..3), 64-byte column block d
                const int row_a = kb * 32 + 16 * ks2 + 4 * g + v_rr;
                const u32x2 v0 = lds_read_tr16(sbase + TILE + swzt_row_off<D>(row_a, d * 64 + v_cb));
                const u32x2 v1 = lds_read_tr16(sbase + TILE + swzt_row_off<D>(row_a + 8, d * 64 + v_cb));
                u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
                oacc[d] = E::mfma(vf, pf, oacc[d]);
            }
        }
#endif
    };

    auto tile_step = [&](auto stage_c, int nb) {
        constexpr int stage = decltype(stage_c)::value;
        const bool has_next = nb + 1 < n_max;
        const int nb_next = nb + 1;
        if (has_next) load_tile(nb_next, std::integral_constant<int, stage ^ 1>{});
        const int n0 = nb * FWD_BN;
        // wave-uniform: does this wave see anything in this tile?
        const bool wave_active = (n0 <= w_hi_max) && (n0 + FWD_BN - 1 >= w_lo_min);
        if (wave_active) compute_tile(stage_c, nb);
        if (has_next) store_tile(std::integral_constant<int, stage ^ 1>{}, nb_next);
        __syncthreads();
    };

    for (int pass = 0; pass < n_pass; ++pass) {
    begin_pass(pass);
    if (m_block >= seqlen_q) continue;
    load_q(my_row, qf);
    rope_q();

    if (n_min < n_max) {
        if (PAGED && (paged_aligned || paged_dma)) pf_request(n_min);
        if (PAGED && paged_q16) pf8_request(n_min);
        load_tile(n_min, std::integral_constant<int, 0>{});
        store_tile(std::integral_constant<int, 0>{}, n_min);
    }
    __syncthreads();
    for (int nb = n_min; nb < n_max; nb += 2) {
        tile_step(std::integral_constant<int, 0>{}, nb);
        if (nb + 1 < n_max) tile_step(std::integral_constant<int, 1>{}, nb + 1);
    }

    // ---- epilogue: O / l, LSE ---------------------------------------------------------------
    const float l_tot = xhalf_sum(l_run);
    const float inv = l_tot > 0.f ? (DROPOUT ? a.rp_dropout : (KV8 ? p.v_descale : 1.0f)) / l_tot : 0.f;
    if (my_row < seqlen_q) {
        // (the store addresses are formed HERE from opaque copies of the lane's row and half: left visible, hipcc computes the 64-bit lane
        //  pointers in front of the tile loop and parks them in scratch / AGPRs across it - csrc/spill_budget.json)
        int row_e = my_row, g_e = g;
#ifndef FA_FWD_TU_D256                                  // (head dim 256: one wave per SIMD, nothing is parked in scratch there; the copies cost it 4 AGPR moves)
        asm volatile("" : "+v"(row_e), "+v"(g_e));
#endif
        uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + (p.cu_seqlens_q ? 0 : (int64_t)w.b * p.o_batch_stride)
                       + (q_row0 + row_e) * p.o_row_stride + (int64_t)w.h * p.o_head_stride;
#pragma unroll
        for (int d = 0; d < DBLKS; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u32x2 o2;
                o2[0] = E::pack2(oacc[d][4 * rq + 0] * inv, oacc[d][4 * rq + 1] * inv);
                o2[1] = E::pack2(oacc[d][4 * rq + 2] * inv, oacc[d][4 * rq + 3] * inv);
                if (d * 32 + 8 * rq + 4 * g_e < dv) *reinterpret_cast<u32x2*>(op + d * 32 + 8 * rq + 4 * g_e) = o2;
            }
        if (g_e == 0) {
            const float lse = l_tot > 0.f ? (m_run + fast_log2(l_tot)) * kLn2 : -INFINITY;
            p.lse[(int64_t)w.b * p.lse_batch_stride + (int64_t)w.h * p.lse_head_stride + q_row0 + row_e] = lse;
        }
    }
    // ---- next pass: fresh accumulators ----
    if (pass + 1 < n_pass) {
#pragma unroll
        for (int d = 0; d < DBLKS; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
        m_run = -INFINITY;
        l_run = 0.f;
    }
    }   // pass
}

// ---- host launcher ---------------------------------------------------------------------------
template <typename T, int D>
static int launch_fwd_td(const KArgs& a, bool paged, hipStream_t stream) {
    const int grid = a.flat_blocks ? a.flat_blocks * a.p.nheads_q
                                   : work_grid(a.p.batch, a.p.nheads_q, a.p.nheads_k, a.n_qblocks);   // n_qblocks = grid-level count
    const size_t smem = FwdSmem<D>::TOTAL;
    if (grid == 0) return 0;
#ifndef FA_FWD_NARROW96
#define FA_FWD_NARROW96 1
#endif
    const bool narrow192 = (D == 256 && valid_cols(a.p) <= 192) || (FA_FWD_NARROW96 && D == 128 && valid_cols(a.p) <= 96);
#define FA_LAUNCH(BIAS, PAGED, DROP)                                                            \
    do {                                                                                        \
        if (narrow192) {                                                                        \
            auto kern = fa_fwd_kernel<T, D, BIAS, PAGED, DROP, false, (D == 256 ? 192 : (D == 128 ? 96 : D))>; \
            FA_SET_LDS_ONCE(kern, smem);                                                        \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(FWD_THREADS), smem, stream, a);           \
        } else {                                                                                \
            auto kern = fa_fwd_kernel<T, D, BIAS, PAGED, DROP>;                                 \
            FA_SET_LDS_ONCE(kern, smem);           /* once per instantiation and device */    \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(FWD_THREADS), smem, stream, a);           \
        }                                                                                       \
    } while (0)
    if (a.p.kv_dtype == FA_FP8_E4M3) {             // fp8 KV cache (general path: long query blocks, ALiBi, softcap)
        if constexpr (D <= 128) {
            if (a.p.p_dropout > 0.f) return -2;
#define FA_LAUNCH8(BIAS, PAGED)                                                                 \
            do {                                                                                \
                auto kern = fa_fwd_kernel<T, D, BIAS, PAGED, false, true>;                      \
                FA_SET_LDS_ONCE(kern, smem);                                                    \
                hipLaunchKernelGGL(kern, dim3(grid), dim3(FWD_THREADS), smem, stream, a);       \
            } while (0)
            if (a.has_bias) { if (paged) FA_LAUNCH8(1, true); else FA_LAUNCH8(1, false); }       // per-element ALiBi / softcap
            else            { if (paged) FA_LAUNCH8(0, true); else FA_LAUNCH8(0, false); }
#undef FA_LAUNCH8
            return 0;
        } else {
            return -2;
        }
    }
    const bool drop = a.p.p_dropout > 0.f;
    const bool lin_alibi = a.p.alibi_slopes && a.p.softcap <= 0.f && (a.p.is_causal || a.p.window_right == 0);
    if (drop) { if (a.has_bias) FA_LAUNCH(1, false, true); else FA_LAUNCH(0, false, true); }
    else if (a.has_bias) {
        if (paged) FA_LAUNCH(1, true, false);
        else if (lin_alibi) FA_LAUNCH(2, false, false);
        else if (!a.p.alibi_slopes) FA_LAUNCH(3, false, false);          // softcap only
        else FA_LAUNCH(1, false, false);
    }
    else if (paged) FA_LAUNCH(0, true, false);
    else FA_LAUNCH(0, false, false);
#undef FA_LAUNCH
    return 0;
}

// The head-dim-256 instantiations live in their own translation unit (fa_fwd_d256.hip = this file under FA_FWD_TU_D256, built with
// -mllvm -amdgpu-mfma-vgpr-form): at one wave per SIMD hipcc otherwise selects every MFMA in accumulator form, S lands in AGPRs, the
// softmax's operands bounce through v_accvgpr_read and the kernel spills 38 - 156 registers of 512 (round-5 review: "one wave per
// SIMD with 512 registers should not spill"); with VGPR-form MFMAs none does (tests/test_build_resources.py, csrc/spill_budget.json).
int launch_fwd_d256(const KArgs& a, bool paged, hipStream_t stream);
#ifdef FA_FWD_TU_D256
int launch_fwd_d256(const KArgs& a, bool paged, hipStream_t stream) {
    return a.p.dtype == FA_BF16 ? launch_fwd_td<bf16_tag, 256>(a, paged, stream) : launch_fwd_td<fp16_tag, 256>(a, paged, stream);
}
#else
bool fwd_asm_applicable(const KArgs& a);
int launch_fwd_asm(const KArgs& a, hipStream_t stream);

// FA_FWD_ASM=0 (read once, at the first call) keeps every shape on fa_fwd_kernel: the A/B switch for the
// hand-scheduled D = 128 path of fa_fwd_asm.hip.
static bool fwd_asm_enabled() {
    static const bool on = [] { const char* e = getenv("FA_FWD_ASM"); return !(e && e[0] == '0'); }();
    return on;
}

int launch_fwd(const KArgs& a0, hipStream_t stream) {
    if (fwd_asm_enabled() && fwd_asm_applicable(a0)) return launch_fwd_asm(a0, stream);
    const KArgs& a = a0;
    const bool paged = a.p.block_table != nullptr;
    const bool bf = a.p.dtype == FA_BF16;
    switch (a.p.head_dim) {
        case 64:  return bf ? launch_fwd_td<bf16_tag, 64>(a, paged, stream) : launch_fwd_td<fp16_tag, 64>(a, paged, stream);
        case 128: return bf ? launch_fwd_td<bf16_tag, 128>(a, paged, stream) : launch_fwd_td<fp16_tag, 128>(a, paged, stream);
        case 256: return launch_fwd_d256(a, paged, stream);
        default:  return -2;
    }
}
#endif   // FA_FWD_TU_D256

}  // namespace fa
