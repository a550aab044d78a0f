// fa_common.h - device-side building blocks shared by the gfx950 attention kernels.
//
// Fragment conventions (v_mfma_f32_32x32x16_{bf16,f16}, wave64):
//   lane l: l31 = l & 31, g = l >> 5
//   A operand: lane holds A[i = l31][k = 8g + j]  (j = 0..7, 16 bytes)
//   B operand: lane holds B[k = 8g + j][n = l31]
//   C/D:       lane holds D[row(r, g)][col = l31],  row(r, g) = (r & 3) + 8 (r >> 2) + 4 g
// The contraction index k only has to be named consistently by A and B, which lets a
// C/D fragment be fed straight back as a B operand ("slot order", see fa_fwd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdlib.h>
#include "fa_mi355.h"

namespace fa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct bf16_tag {};
struct fp16_tag {};

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

template <typename T> struct Elem;

template <> struct Elem<bf16_tag> {
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        f32x2 v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    }
    static __device__ __forceinline__ float lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
    static __device__ __forceinline__ float hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
};

template <> struct Elem<fp16_tag> {
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        f32x2 v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
    }
    static __device__ __forceinline__ float lo(uint32_t w) {
        return (float)__builtin_bit_cast(f16x2, w)[0];
    }
    static __device__ __forceinline__ float hi(uint32_t w) {
        return (float)__builtin_bit_cast(f16x2, w)[1];
    }
};

// ---- LDS helpers -----------------------------------------------------------------
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short lds_i16x4_t;

// ds_read_b64_tr_b16: within each 16-lane group the 16 x 8-byte loads form a
// [4 rows][16 cols] matrix (lane i' supplies row i'>>2, cols 4(i'&3)..+3); lane i
// receives column i: element j = M[j][i].
__device__ __forceinline__ u32x2 lds_read_tr16(const char* smem_ptr) {
    auto p = (__attribute__((address_space(3))) lds_i16x4_t*)(smem_ptr);
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(p));
}

// LDS-space pointers whose VALUE is pinned in a VGPR: hipcc otherwise re-derives "LDS base (0) + lane offset" with a
// v_add_u32 in front of every read instead of keeping the lane-constant address and using the immediate offset field.
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ const lds_char* lds_pin(const char* p) {
    const lds_char* q = (const lds_char*)p;
    asm volatile("" : "+v"(q));
    return q;
}
__device__ __forceinline__ u32x4 lds_read_b128(const lds_char* p) {
    return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(p);
}
__device__ __forceinline__ u32x2 lds_read_tr16(const lds_char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lds_i16x4_t*)(p)));
}
// ds_read_b64_tr_b16 as inline asm + an explicit counted wait.  Why: for the BUILTIN form hipcc's waitcnt pass puts an
// s_waitcnt vmcnt(0) in front of the first transposed read that follows an LDS-DMA (buffer_load ... lds) - it cannot
// tell that the DMA in flight fills the OTHER stage buffer - so every tile step stalled for the full latency of the loads
// it had just issued for the next tile (plain ds_read_b128 loads do not get that wait).  The asm form carries no memory
// operand; its completion is waited for with lds_tr_wait(frag, n): n = LDS instructions issued after the fragment's
// reads (LDS returns in order, so "at most n outstanding" means the fragment has landed; instructions the compiler
// adds in between only make the wait stricter).  The "+v" tie keeps the consuming MFMA behind the wait.
__device__ __forceinline__ u32x2 lds_read_tr16_nw(const lds_char* p, int off) {
    u32x2 r;
    const lds_char* q = p + (off & ~0xffff);             // (the offset field has 16 bits; `off` is a constant after inlining)
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(q), "i"(off & 0xffff) : "memory");
    return r;
}
__device__ __forceinline__ void lds_tr_wait(u32x4& frag, int n) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "i"(n));
}
// the byte form (fp8 operands: ds_read_b64_tr_b8) and a wait on one 8-byte fragment, same reasoning
__device__ __forceinline__ u32x2 lds_read_tr8_nw(const lds_char* p, int off) {
    u32x2 r;
    const lds_char* q = p + (off & ~0xffff);
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r) : "v"(q), "i"(off & 0xffff) : "memory");
    return r;
}
__device__ __forceinline__ void lds_tr_wait(u32x2& frag, int n) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "i"(n));
}
__device__ __forceinline__ u32x4 lds_read_b128(const char* smem_ptr) {
    return *reinterpret_cast<const u32x4*>(smem_ptr);
}
__device__ __forceinline__ void lds_write_b128(char* smem_ptr, u32x4 v) {
    *reinterpret_cast<u32x4*>(smem_ptr) = v;
}

// Row-major [rows][D] 16-bit tile, XOR-swizzled in 16-byte slots so that ds_read_b128 of
// one logical slot from 16 rows that differ mod 16 is bank-conflict free
// (bank = (addr/4) % 64 over a 256-byte line).
template <int D>
__device__ __forceinline__ int swz_row_off(int row, int col_byte) {
    constexpr int ROWB = D * 2;
    constexpr int SPR = ROWB / 16;                       // 16-B slots per row
    constexpr int MASK = (SPR < 16 ? SPR : 16) - 1;
    constexpr int SHIFT = (SPR >= 16) ? 0 : (SPR == 8 ? 1 : (SPR == 4 ? 2 : 3));
    const int f = (row >> SHIFT) & MASK;
    return row * ROWB + (col_byte ^ (f << 4));
}

// Row-major [rows][D] tile that is read BOTH by rows (ds_read_b128, A operand) and
// transposed (ds_read_b64_tr_b16 on 4 consecutive rows x 64 bytes).  The slot XOR uses
// row bits (1:0 -> slot bits 3:2) so that the 4 rows of a transpose read land in four
// different 64-byte bank groups, and stays a bijection over 16 rows for the row reads.
template <int D>
__device__ __forceinline__ int swzt_row_off(int row, int col_byte) {
    constexpr int ROWB = D * 2;
    static_assert(D == 64 || D == 128 || D == 256, "swzt layout defined for D = 64, 128, 256");
    const int f = (D >= 128) ? (((row & 3) << 2) | ((row >> 2) & 3))
                             : ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
    return row * ROWB + (col_byte ^ (f << 4));
}

// [rows][D] 16-bit tile stored as [rows/4][D/32] blocks of [4 rows][32 cols] (256 B each):
// a 32-lane ds_read_b64_tr_b16 then covers exactly one 256-byte bank line.
template <int D>
__device__ __forceinline__ int vtile_off(int row, int col) {
    return (((row >> 2) * (D / 32) + (col >> 5)) << 8) + ((row & 3) << 6) + ((col & 31) << 1);
}

// (x & m) | (other & ~m) in one v_bitop3_b32 (truth table 0xE2 with a = 0xF0, b = 0xCC, c = 0xAA): the masks' select
// without a compare, VCC or the wait states between v_cmp and v_cndmask.  m is all ones (keep x) or zero per element.
__device__ __forceinline__ float select_bits(float x, uint32_t m, uint32_t other) {
    return __builtin_bit_cast(float, __builtin_amdgcn_bitop3_b32(__builtin_bit_cast(uint32_t, x), m, other, 0xE2));
}

// max(a, b, c) in one VALU instruction (IEEE maxNum semantics; not volatile: the optimiser may still move / drop it)
__device__ __forceinline__ float max3_f32(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// tanh via exp2: tanh(x) = 1 - 2 / (1 + e^{2x}); saturates correctly at +-inf.
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = fast_exp2(x * (2.0f * kLog2e));
    return 1.0f - 2.0f * fast_rcp(1.0f + e);
}

// Cross-half (lane l <-> l ^ 32) reductions with one v_permlane32_swap (no LDS round trip).
// v_permlane32_swap_b32 vdst, src: lanes 32-63 of vdst swap with lanes 0-31 of src.  With
// a == b == x: a' = {lo: x_lo, hi: x_lo}, b' = {lo: x_hi, hi: x_hi}.
// Inline asm on purpose: clang (ROCm 7.2) lowers r[1] of __builtin_amdgcn_permlane32_swap to
// extractvalue 0, i.e. returns the first result twice.  "s_nop 1" = the 2 wait states a
// VALU-written operand needs before v_permlane*_swap reads it.
__device__ __forceinline__ void permlane32_swap(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xhalf_max(float v) {
    uint32_t a = __builtin_bit_cast(uint32_t, v), b = a;
    permlane32_swap(a, b);
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    uint32_t a = __builtin_bit_cast(uint32_t, v), b = a;
    permlane32_swap(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

// ---- global -> register tile loads through a buffer descriptor ------------------------------
// One descriptor per (batch, head) slice: rows past `nrows` read as zero (hardware range
// check), so tail tiles need no predication and no zero-fill code.  voffset is per-lane and
// loop-invariant; the tile / row offset travels in the scalar soffset.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t row_stride_elems,
                                                            int nrows, int d) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const int64_t bytes64 = nrows > 0 ? ((int64_t)(nrows - 1) * row_stride_elems + d) * 2 : 0;
    const uint32_t bytes = __builtin_amdgcn_readfirstlane((uint32_t)(bytes64 > 0xffffffffll ? 0xffffffffll : bytes64));
    void* ptr = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(ptr, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load_b128(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
// LDS-DMA: buffer_load_dwordx4 ... lds.  The 64 lanes write 1 KiB at `lds_dst` + lane * 16
// (destination is lane-linear; a swizzled LDS image is obtained by permuting the per-lane SOURCE
// offset `voff`), out-of-range lanes write zeros (probed: tools/probes/probe_lds_dma_oob.hip).
// No VGPR staging, no ds_write; completion is tracked by vmcnt (hipcc waits before the barrier).
__device__ __forceinline__ void buf_load_lds_b128(__amdgpu_buffer_rsrc_t r, char* lds_dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

// ---- dropout: Philox-4x32-10, the reference's stream (include/philox.h:13-73) -----------------
// element (i_glob, j): flat = i_glob * N_glob + j; counter = offset + (flat >> 2); word = flat & 3;
// kept iff word <= thr  (include/softmax.h:97-114).
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                             uint32_t k0, uint32_t k1) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: the integer
    // multiplier is the quarter-rate unit and Philox is all multiplies (40 -> 20 per call)
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
}
__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    philox_round(c0, c1, c2, c3, k0, k1);
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
struct DropCtx { uint32_t k0, k1, thr; uint64_t offset; };
// keep-mask (bit e = element flat0 + e kept) of 4 consecutive flat indices
__device__ __forceinline__ uint32_t dropout_keep4(const DropCtx& dc, uint64_t flat0) {
    uint32_t r[4];
    philox4x32_10(dc.offset + (flat0 >> 2), dc.k0, dc.k1, r);
    const int m = (int)(flat0 & 3);
    uint32_t bits = 0;
    if (m == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bits |= (r[e] <= dc.thr ? 1u : 0u) << e;
    } else {
        uint32_t r2[4];
        philox4x32_10(dc.offset + (flat0 >> 2) + 1, dc.k0, dc.k1, r2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int w = m + e;
            const uint32_t v = w < 4 ? (w == 1 ? r[1] : (w == 2 ? r[2] : r[3])) : (w == 4 ? r2[0] : (w == 5 ? r2[1] : r2[2]));
            bits |= (v <= dc.thr ? 1u : 0u) << e;
        }
    }
    return bits;
}
__device__ __forceinline__ bool dropout_keep1(const DropCtx& dc, uint64_t flat) {
    uint32_t r[4];
    philox4x32_10(dc.offset + (flat >> 2), dc.k0, dc.k1, r);
    const int m = (int)(flat & 3);
    const uint32_t v = m == 0 ? r[0] : (m == 1 ? r[1] : (m == 2 ? r[2] : r[3]));
    return v <= dc.thr;
}

// Work decomposition shared by all dense kernels: 1-D grid, block id -> XCD-aware.  A "unit" is one (batch, kv-head): its
// K / V (forward, dQ) or Q / dO stream (dK/dV) is shared by its `per_unit` items (q-blocks x q-heads of the group; key blocks x
// splits), and blocks are observed to land on XCD id % 8 (speed only, never correctness) - so unit u of a round of eight goes to
// XCD u % 8 and its items stay in that XCD's L2.  Units past the last full round of eight (batch x kv-heads = 1 for a
// tensor-parallel shard of a GQA model, 2-4 for MQA / Qwen2-style models at small batch) would leave XCDs empty that way:
// their items are laid end to end instead and cut into eight runs whose lengths differ by at most one, one per XCD (an XCD then
// serves one or two units).
struct UnitItem { int unit, item; bool valid; };
__host__ __device__ __forceinline__ int unit_grid(int units, int per_unit) {
    const int full8 = units & ~7, tail = units - full8;
    return full8 * per_unit + (tail ? 8 * (int)(((int64_t)tail * per_unit + 7) / 8) : 0);
}
__device__ __forceinline__ UnitItem decode_unit_item(int id, int units, int per_unit) {
    UnitItem w;
    const int full8 = units & ~7;
    const int head_ids = full8 * per_unit;
    if (id < head_ids) {
        const int xcd = id & 7, j = id >> 3;
        const int ul = j / per_unit;
        w.item = j - ul * per_unit;
        w.unit = ul * 8 + xcd;
        w.valid = true;
    } else {
        const int tail_items = (units - full8) * per_unit;
        const int id2 = id - head_ids, x = id2 & 7;
        const int lo = (int)(((int64_t)tail_items * x) >> 3), hi = (int)(((int64_t)tail_items * (x + 1)) >> 3);   // XCD x: items [lo, hi)
        const int lin = lo + (id2 >> 3);
        w.valid = lin < hi;
        const int u = lin / per_unit;
        w.item = lin - u * per_unit;
        w.unit = full8 + u;
    }
    return w;
}
struct WorkItem { int b, h, hk, qb; bool valid; };
__device__ __forceinline__ WorkItem decode_work(int id, int batch, int nheads_q, int nheads_k,
                                                int n_qblocks) {
    WorkItem w;
    const int group = nheads_q / nheads_k;
    const UnitItem ui = decode_unit_item(id, batch * nheads_k, group * n_qblocks);
    const int gq = ui.item / n_qblocks;
    w.valid = ui.valid;
    w.qb = n_qblocks - 1 - (ui.item - gq * n_qblocks);
    w.b = ui.unit / nheads_k;
    w.hk = ui.unit - w.b * nheads_k;
    w.h = w.hk * group + gq;
    return w;
}
// Varlen "flat" work list: instead of batch x ceil(max_seqlen / block) q-block slots (half of them empty for a
// typical length mix, and every empty workgroup still costs a dispatch slot), the grid has
//     F = floor(total_q / block) + batch
// slots per head.  Sequence b owns slots [start_b, start_{b+1}) with start_b = floor(cu[b] / block) + b, which is
// strictly increasing and leaves at least ceil(len_b / block) slots per sequence (at most one spare).  The owner of
// slot f is found with one vector load of cu_seqlens per 64 sequences + a ballot; slots run from the last one down
// so that the late (for causal masks: heavy) q-blocks of a sequence start first; id -> (slot, head) keeps a head on
// one XCD when nheads_q % 8 == 0, so its K/V stay in that XCD's L2 as before.
__device__ __forceinline__ void flat_owner(int f, int block, int batch, const int32_t* cu, int lane, int& b, int& blk) {
    int cnt = 0;
    for (int b0 = 0; b0 < batch; b0 += 64) {
        const int i = b0 + lane;
        const bool le = i < batch && (cu[i] / block + i) <= f;
        cnt += __popcll(__ballot(le));
    }
    b = __builtin_amdgcn_readfirstlane(cnt) - 1;
    const int bb = b >= 0 ? b : 0;
    blk = f - (cu[bb] / block + bb);
}
__device__ __forceinline__ WorkItem decode_work_flat(int id, int flat_blocks, int block_m, int batch, int nheads_q,
                                                     int nheads_k, const int32_t* cu_seqlens_q, int lane) {
    WorkItem w;
    const int f = flat_blocks - 1 - id / nheads_q;
    w.h = id - (id / nheads_q) * nheads_q;
    w.hk = w.h / (nheads_q / nheads_k);
    flat_owner(f, block_m, batch, cu_seqlens_q, lane, w.b, w.qb);
    w.valid = w.b >= 0;
    return w;
}
static inline int work_grid(int batch, int nheads_q, int nheads_k, int n_qblocks) {
    return unit_grid(batch * nheads_k, (nheads_q / nheads_k) * n_qblocks);
}

// 16 x fp8-e4m3 -> 16 x 16-bit with gfx950's packed converts: one VALU op per TWO elements
// (v_cvt_scalef32_pk_{f16,bf16}_fp8; probed in tools/probes/probe_cvt_scalef32.hip: exact for all
// 256 byte patterns at scale 1.0, and the f32 scale operand contributes only its exponent - so the
// cache descales stay folded into the softmax scale / the final normalisation).
template <typename T>
__device__ __forceinline__ uint32_t fp8x2_to_16bit(uint32_t w, bool hi_word);
template <>
__device__ __forceinline__ uint32_t fp8x2_to_16bit<fp16_tag>(uint32_t w, bool hi_word) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 r = hi_word ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true)
                         : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false);
    return __builtin_bit_cast(uint32_t, r);
}
template <>
__device__ __forceinline__ uint32_t fp8x2_to_16bit<bf16_tag>(uint32_t w, bool hi_word) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 r = hi_word ? __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true)
                         : __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false);
    return __builtin_bit_cast(uint32_t, r);
}
// Eight values as a SUM of NT e4m3 fragments (byte j of every fragment belongs to value j): each fragment is the round-
// to-nearest code of what the previous ones left over, so two carry ~2^-8 of the value (bf16's resolution), three ~2^-12
// (fp16's) - the remainders have their own exponents.  |a| must stay below the format's 448.
typedef int i32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int NT>
__device__ __forceinline__ void fp8_terms8(const float (&a)[8], u32x2 (&out)[NT]) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(r[0], r[1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(r[2], r[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(r[4], r[5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(r[6], r[7], w1, true);
        out[t] = u32x2{(uint32_t)w0, (uint32_t)w1};
        if (t + 1 < NT) {
            const f32x2_t b0 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, false), b1 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, true);
            const f32x2_t b2 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, false), b3 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, true);
            r[0] -= b0[0]; r[1] -= b0[1]; r[2] -= b1[0]; r[3] -= b1[1];
            r[4] -= b2[0]; r[5] -= b2[1]; r[6] -= b3[0]; r[7] -= b3[1];
        }
    }
}
template <typename T>
__device__ __forceinline__ void fp8x16_to_16bit(const u32x4& in, u32x4& lo, u32x4& hi) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t p0 = fp8x2_to_16bit<T>(in[w], false), p1 = fp8x2_to_16bit<T>(in[w], true);
        if (w < 2) { lo[2 * w] = p0; lo[2 * w + 1] = p1; }
        else       { hi[2 * (w - 2)] = p0; hi[2 * (w - 2) + 1] = p1; }
    }
}

// Raise a kernel's dynamic-LDS limit once per instantiation AND DEVICE: the attribute belongs to the device's copy of
// the function, and one process may drive several GPUs (the Python layer wraps every call in torch.cuda.device(q.device)).
// `done` is a per-instantiation bit mask over device ordinals (devices >= 64 set the attribute on every launch); a failed
// hipFuncSetAttribute leaves the bit clear and is reported by the hipGetLastError() check behind the launch (fa_api.hip).
static inline void fa_set_max_lds(std::atomic<uint64_t>& done, const void* kern, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    const uint64_t bit = dev < 64 ? (1ull << dev) : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess && bit)
        done.fetch_or(bit, std::memory_order_release);
}
#define FA_SET_LDS_ONCE(kern, bytes)                                                     \
    do {                                                                                 \
        static std::atomic<uint64_t> fa_attr_done_{0};   /* one per expansion site */     \
        fa_set_max_lds(fa_attr_done_, reinterpret_cast<const void*>(kern), (int)(bytes)); \
    } while (0)

// Compute units of the current device (cached per device ordinal: one process may drive several GPUs).
static inline int fa_device_cu_count() {
    // per device ordinal: one process may drive several GPUs (cf. FA_SET_LDS_ONCE in fa_common.h)
    constexpr int MAXDEV = 64;
    static std::atomic<int> cache[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 256;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cache[dev].store(v, std::memory_order_relaxed);
    return v;
}

// Mirrored block pairs (causal masks): the low block first in every workgroup.  -DFA_PAIR_FLIP=1 (experiment build)
// alternates the order with the workgroup index so that the pro / epilogues of the CUs do not line up at launch:
// measured 1-2 % SLOWER on forward and dQ at config 2 (tools/experiments/README.md)
#ifndef FA_PAIR_FLIP
#define FA_PAIR_FLIP 0
#endif

// Dense causal-like problems on the 256-row hand-scheduled kernels (forward, dQ): with an ODD number of 256-row blocks the
// middle block has no mirror and runs alone, and a last block that is half empty idles two of the four waves - measured
// against the 128-row compiler kernels (tools/seqlen_sweep.py, 8 k / 32 k / 131 k tokens): 3 blocks (S 640, 768) +17 ... +47 %,
// 5 and 7 blocks -5 ... +22 %, S 384 / 896 (128 valid rows in the last block) +4 ... +8 %, every other length 5 - 13 % faster.
// `paired`: the launch pairs mirrored blocks (causal-like mask without a left window).  FA_ASM_FORCE=1 (read per call:
// tests switch it) takes the hand-scheduled kernels wherever they are correct.
static inline bool asm_256row_blocks_pay(int seqlen_q, bool paired) {
    const char* e = getenv("FA_ASM_FORCE");
    if (e && e[0] == '1') return true;
    const int n = (seqlen_q + 255) / 256, waste = n * 256 - seqlen_q;
    if (paired && (n < 2 || ((n & 1) && n < 9))) return false;
    if (waste >= 128 && n < 6) return false;
    return true;
}

// Host-side launch args: the ABI struct plus derived values.
constexpr int FA_FS_MAX_ITEMS = 64, FA_FS_MAX_BLOCKS = 32;
struct KArgs {
    fa_params p;
    int n_qblocks;         // q-blocks per (batch, head) in the GRID (halved when pairing)
    int n_qblocks_total;   // ceil(seqlen_q / block_m)
    int pair_qblocks;      // causal load balance: a workgroup owns q-blocks (i, total-1-i)
    int flat_blocks;       // varlen flat work list: slots per head (0: batch x n_qblocks grid), see decode_work_flat
    int flat_kblocks;      // the same over 128-key blocks (dK/dV kernel)
    int has_bias;          // alibi or softcap
    float scale_log2e;
    uint32_t drop_thr;     // uint32((1 - p) * 4294967295.0f), fp32 arithmetic (include/softmax.h:51)
    float rp_dropout;      // 1 / (1 - p)
    const int32_t* seqlens_k;      // per-batch key count (seqused_k or cache_seqlens), or NULL
    int seqlen_k_add;              // added to seqlens_k[b] (kvcache: T_new)
    const int32_t* kv_batch_idx;   // cache_batch_idx or NULL
    const int32_t* leftpad_k;      // or NULL
    int kv_mode;                   // kv-cache call: seqlens_k == NULL means "the cache is empty" (keys = the new rows only),
                                   // as in the reference (fused_mha_forward_kvcache.cu:85) and oracle/kvcache.py
    // backward: dS hand-off from the dK/dV kernel to the dQ kernel (NULL: dQ recomputes S and dP)
    void* ds_ws;                   // [B, Hq, ds_nqb, ds_nkb][2 KiB]: one 32-query x 32-key dS tile each
    int ds_nqb, ds_nkb;            // ceil(seqlen_q / 32), ceil(seqlen_k / 32)
    // forward key split of one-wave causal launches (fa_fwd_asm.hip: launch_asm_t): fs_items > 0: the work list per (batch, head) is
    // fs_items entries (heaviest first) of (256-row block, key-tile range, part); blocks from fs_qb0 on are split into fs_parts[qb]
    // parts whose fp32 partial O / LSE land in fs_o / fs_lse ([part][B][Hq][rows from fs_qb0 * 256][128] / [..][rows])
    int fs_items, fs_qb0, fs_max_parts;
    uint8_t fs_qb[FA_FS_MAX_ITEMS], fs_part[FA_FS_MAX_ITEMS], fs_parts[FA_FS_MAX_BLOCKS];
    uint16_t fs_t0[FA_FS_MAX_ITEMS], fs_t1[FA_FS_MAX_ITEMS];
    float* fs_o;
    float* fs_lse;
    // dS hand-off between the GENERATED dK/dV kernel and fa_bwd_dq_ds_kernel (fa_bwd_dq_ds.hip): [B, Hq, ds2_nkb, ds2_nqb][2 KiB]
    void* ds2_ws;
    int ds2_nqb, ds2_nkb;          // ceil(seqlen_q / 32), 4 * ceil(seqlen_k / 128)
    // backward, asm dK/dV kernel (fa_bwd_asm.hip): row statistics written by the preprocess kernel, or NULL
    float* stats_ws;               // [2][B][Hq][Sq]: plane 0 = LSE log2(e) (+inf where LSE = -inf), plane 1 = -D
    int skip_short_q;              // varlen forward of a mixed batch: sequences with 1 .. skip_short_q query rows are served by the
                                   // decode kernels (fa_api.hip: varlen_mixed_route) - fa_fwd_kernel leaves them out
    int rope_q;                    // kv-cache general path: fa_fwd_kernel rotates its Q fragments in registers (rotary_cos / sin at
                                   // position cache_seqlens[b] + leftpad (+ row under a causal / local mask), include/rotary.h:176-202)
    int fuse_pre;                  // the dQ kernel computes D = rowsum(dO o O) itself, runs first and writes softmax_d + stats_ws
    // backward, dense dK/dV launches with fewer workgroups than the chip has slots (GQA at micro-batch 1, short-key
    // cross-attention): the query tiles of every pass are divided over dkv_split workgroups, each leaves an fp32 partial
    // dK / dV and dkv_reduce_kernel sums them (fa_bwd.hip: dkv_split_factor)
    int dkv_split;                 // 0 / 1: one workgroup per key block
    void* dkv_part;                // fp32 [dK | dV][dkv_split][B][Sk][Hk][head_dim]
};

// the query tiles [mt0, mt1) of a dK/dV pass that split `s` of `n` walks
__device__ __forceinline__ void dkv_split_range(int s, int n, int& mt0, int& mt1) {
    const int nt = mt1 - mt0;
    const int lo = mt0 + (int)((int64_t)nt * s / n), hi = mt0 + (int)((int64_t)nt * (s + 1) / n);
    mt0 = lo; mt1 = hi;
}

// ---- backward: geometry of one sequence, key-block size of the dK/dV kernels ----
constexpr int DKV_BN = 128;     // keys per workgroup (32 per wave)
struct SeqGeom {
    int seqlen_q, seqlen_k, off;
    int64_t q_row0, k_row0;
};
__device__ __forceinline__ SeqGeom seq_geom(const fa_params& p, int b) {
    SeqGeom s;
    s.seqlen_q = p.seqlen_q; s.seqlen_k = p.seqlen_k; s.q_row0 = 0; s.k_row0 = 0;
    if (p.cu_seqlens_q) { s.q_row0 = p.cu_seqlens_q[b]; s.seqlen_q = p.cu_seqlens_q[b + 1] - (int)s.q_row0; }
    if (p.cu_seqlens_k) { s.k_row0 = p.cu_seqlens_k[b]; s.seqlen_k = p.cu_seqlens_k[b + 1] - (int)s.k_row0; }
    s.off = s.seqlen_k - s.seqlen_q;
    return s;
}


// ---- ALiBi through the matrix pipe (causal-like masks: every visible key is at or left of the diagonal) ----
// bias(q, key) = slope (key - off - q) is split into three small terms relative to the 32 x 32 sub-tile,
//   slope * [ +-pos(register)  +  lane term  +  tile term ],
// and added to the score accumulator by ONE extra MFMA whose contraction slots carry
//   k0,k1: A = pos (0..31, exact in 16 bit)   B = head / tail of +-slope/scale
//   k2,k3: A = 1                               B = head / tail of the lane's term
//   k4-k6: A = 1                               B = three-way split of the tile term (24-bit mantissa)
// so no per-element VALU work is left.  All terms are small near the diagonal (where P matters) and the
// 16-bit splits bound the error by ~3e-4 in log2 units elsewhere.
template <typename T>
__device__ __forceinline__ u32x4 alibi_pos_operand(int lane) {       // the operand indexed by register position
    using E = Elem<T>;
    u32x4 r = {0, 0, 0, 0};
    if ((lane >> 5) == 0) {
        const float pos = (float)(lane & 31);
        r[0] = E::pack2(pos, pos);
        r[1] = E::pack2(1.f, 1.f);
        r[2] = E::pack2(1.f, 1.f);
        r[3] = E::pack2(1.f, 0.f);
    }
    return r;
}
template <typename T>
__device__ __forceinline__ uint32_t split2_16(float x) {
    using E = Elem<T>;
    const float hi = E::lo(E::pack2(x, 0.f));
    return E::pack2(hi, x - hi);
}
// sv = slope / softmax_scale; pos_sign: sign of the register-position term; lane_term, tile_term in key units
template <typename T>
__device__ __forceinline__ u32x4 alibi_lane_operand(int lane, float sv, float pos_sign, float lane_term, float tile_term) {
    using E = Elem<T>;
    u32x4 r = {0, 0, 0, 0};
    if ((lane >> 5) == 0) {
        r[0] = split2_16<T>(pos_sign * sv);
        r[1] = split2_16<T>(sv * lane_term);
        const float x = sv * tile_term;
        const float hi = E::lo(E::pack2(x, 0.f));
        const float mid = E::lo(E::pack2(x - hi, 0.f));
        r[2] = E::pack2(hi, mid);
        r[3] = E::pack2(x - hi - mid, 0.f);
    }
    return r;
}

// Valid columns of a row (fa_params::head_dim_v).  A 16-byte chunk at or past this column is fetched with
// kOobVoff, an offset the buffer descriptor's range check turns into zeros (slices are < 2 GiB then, so
// voffset + soffset cannot wrap), and is never stored.
constexpr uint32_t kOobVoff = 0x80000000u;
__host__ __device__ __forceinline__ int valid_cols(const fa_params& p) { return p.head_dim_v > 0 ? p.head_dim_v : p.head_dim; }

// One 32 x 32 (query, key) sub-tile holds at least one visible pair.  The dK/dV kernel writes a dS
// tile exactly when this is true and the dQ kernel reads exactly those tiles.
__device__ __forceinline__ bool subtile_active(int q0, int k0, int seqlen_q, int seqlen_k, int off, int wl, int wr) {
    if (k0 >= seqlen_k) return false;
    const int k_last = k0 + 31 < seqlen_k ? k0 + 31 : seqlen_k - 1;
    int qlo_min = 0, qhi_max = seqlen_q - 1;
    if (wr >= 0) { const int t = k0 - off - wr; qlo_min = t > 0 ? t : 0; }
    if (wl >= 0) { const int t = k_last - off + wl; qhi_max = t < qhi_max ? t : qhi_max; }
    return q0 <= qhi_max && q0 + 31 >= qlo_min;
}

}  // namespace fa
