// fa_rope.h - rotary embedding of one 16-byte chunk (8 elements), shared by the cache-append,
// q-rotate and decode kernels.  include/rotary.h:91-141 of the reference: math in fp32, result
// rounded to the 16-bit io type.
//   interleaved: (x[2t], x[2t+1]) -> (x0 c - x1 s, x0 s + x1 c)
//   NeoX:        (x[t], x[t+rd/2]) -> (x0 c - x1 s, x0 s + x1 c)
#pragma once
#include "fa_common.h"

namespace fa {

template <typename T>
__device__ __forceinline__ void rope_chunk(u32x4& x, const u32x4& xp, const uint16_t* cosp, const uint16_t* sinp,
                                           int d_base, int rd, bool interleaved) {
    // x: 8 elements at d_base..d_base+7; xp: partner chunk (NeoX only).
    using E = Elem<T>;
    if (d_base >= rd) return;
    const int half = rd >> 1;
    if (interleaved) {
        const u32x2 cw = *reinterpret_cast<const u32x2*>(cosp + d_base / 2);   // 4 cos values
        const u32x2 sw = *reinterpret_cast<const u32x2*>(sinp + d_base / 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x0 = E::lo(x[i]), x1 = E::hi(x[i]);
            const uint32_t cword = cw[i >> 1], sword = sw[i >> 1];
            const float c = (i & 1) ? E::hi(cword) : E::lo(cword);
            const float s = (i & 1) ? E::hi(sword) : E::lo(sword);
            x[i] = E::pack2(fmaf(x0, c, -x1 * s), fmaf(x0, s, x1 * c));
        }
    } else {
        const bool first = d_base < half;
        const int t0 = first ? d_base : d_base - half;
        const u32x4 cw = *reinterpret_cast<const u32x4*>(cosp + t0);           // 8 cos values
        const u32x4 sw = *reinterpret_cast<const u32x4*>(sinp + t0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a0 = E::lo(x[i]), a1 = E::hi(x[i]);
            const float b0 = E::lo(xp[i]), b1 = E::hi(xp[i]);
            const float c0 = E::lo(cw[i]), c1 = E::hi(cw[i]);
            const float s0 = E::lo(sw[i]), s1 = E::hi(sw[i]);
            // first half: y = x0 c - x1 s (x0 = own, x1 = partner); second: y = x0 s + x1 c (x0 = partner, x1 = own)
            const float y0 = first ? fmaf(a0, c0, -b0 * s0) : fmaf(b0, s0, a0 * c0);
            const float y1 = first ? fmaf(a1, c1, -b1 * s1) : fmaf(b1, s1, a1 * c1);
            x[i] = E::pack2(y0, y1);
        }
    }
}

}  // namespace fa
