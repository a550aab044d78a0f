// fa_rows.hip - row gather / scatter for the padding helpers either side of the varlen path
// (reference: flash_attn/bert_padding.py:9-60 index_first_axis / index_put_first_axis, :79-146 unpad_input / pad_input,
//  which go through torch.gather / index assignment with an index tensor expanded to every element).
// Pure byte movement, HBM-bound: a row is moved as 16-byte pieces, one wave-instruction per 1 KiB, rows are
// addressed by 64-bit row indices read once per row (scalar, wave-uniform).
//   gather :  dst[i]          = src[indices[i]]                      i < n_idx
//   scatter:  dst[indices[i]] = src[i], every other dst row = 0      (one pass over dst when the indices are
//             sorted and unique - what unpad_input produces: each dst row looks its source up by binary search;
//             otherwise memset + scatter)
#include <cstdint>
#include "fa_common.h"

namespace fa {

constexpr int ROWS_THREADS = 256;

// every byte is touched once: stream it (nontemporal) instead of leaving it in L2
__device__ __forceinline__ u32x4 ld_stream(const u32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(u32x4* p, u32x4 v) { __builtin_nontemporal_store(v, p); }

constexpr int ROWS_UNROLL = 4;                              // rows in flight per workgroup step (loads first, then stores)
constexpr int ROWS_CHUNK = 32;                              // destination rows per workgroup step of the sorted scatter

// a workgroup moves ROWS_UNROLL rows per step (grid-stride over row groups); within a row the 256 threads walk
// 16-byte pieces
__global__ void __launch_bounds__(ROWS_THREADS) gather_rows_kernel(const char* __restrict__ src, const int64_t* __restrict__ idx,
                                                                    char* __restrict__ dst, int64_t n_idx, int64_t row_bytes,
                                                                    int64_t src_stride, int64_t n_src_rows) {
    const int64_t pieces = row_bytes >> 4;
    for (int64_t i0 = (int64_t)blockIdx.x * ROWS_UNROLL; i0 < n_idx; i0 += (int64_t)gridDim.x * ROWS_UNROLL) {
        const u32x4* s[ROWS_UNROLL];
#pragma unroll
        for (int k = 0; k < ROWS_UNROLL; ++k) {
            const int64_t i = i0 + k < n_idx ? i0 + k : n_idx - 1;
            int64_t r = idx[i];
            if (r < 0) r += n_src_rows;                      // torch-style negative indices
            s[k] = reinterpret_cast<const u32x4*>(src + r * src_stride);
        }
        for (int64_t c = threadIdx.x; c < pieces; c += ROWS_THREADS) {
            u32x4 v[ROWS_UNROLL];
#pragma unroll
            for (int k = 0; k < ROWS_UNROLL; ++k) v[k] = ld_stream(s[k] + c);
#pragma unroll
            for (int k = 0; k < ROWS_UNROLL; ++k)
                if (i0 + k < n_idx) st_stream(reinterpret_cast<u32x4*>(dst + (i0 + k) * row_bytes) + c, v[k]);
        }
    }
}

__global__ void __launch_bounds__(ROWS_THREADS) scatter_rows_kernel(const char* __restrict__ src, const int64_t* __restrict__ idx,
                                                                     char* __restrict__ dst, int64_t n_idx, int64_t row_bytes,
                                                                     int64_t n_dst_rows) {
    const int64_t pieces = row_bytes >> 4;
    for (int64_t i0 = (int64_t)blockIdx.x * ROWS_UNROLL; i0 < n_idx; i0 += (int64_t)gridDim.x * ROWS_UNROLL) {
        u32x4* d[ROWS_UNROLL];
#pragma unroll
        for (int k = 0; k < ROWS_UNROLL; ++k) {
            const int64_t i = i0 + k < n_idx ? i0 + k : n_idx - 1;
            int64_t r = idx[i];
            if (r < 0) r += n_dst_rows;
            d[k] = reinterpret_cast<u32x4*>(dst + r * row_bytes);
        }
        for (int64_t c = threadIdx.x; c < pieces; c += ROWS_THREADS) {
            u32x4 v[ROWS_UNROLL];
#pragma unroll
            for (int k = 0; k < ROWS_UNROLL; ++k)
                v[k] = ld_stream(reinterpret_cast<const u32x4*>(src + (i0 + k < n_idx ? i0 + k : n_idx - 1) * row_bytes) + c);
#pragma unroll
            for (int k = 0; k < ROWS_UNROLL; ++k)
                if (i0 + k < n_idx) st_stream(d[k] + c, v[k]);
        }
    }
}

// sorted unique indices, one pass over dst: a workgroup step owns ROWS_CHUNK consecutive destination rows; ONE binary
// search finds the first index >= the chunk's first row, the (at most ROWS_CHUNK) indices that fall into the chunk
// are held one per lane, and two ballots per row say whether the row has a source and which one.
__global__ void __launch_bounds__(ROWS_THREADS) scatter_rows_sorted_kernel(const char* __restrict__ src, const int64_t* __restrict__ idx,
                                                                            char* __restrict__ dst, int64_t n_idx, int64_t row_bytes,
                                                                            int64_t n_dst_rows) {
    const int64_t pieces = row_bytes >> 4;
    const int lane = threadIdx.x & 63;
    for (int64_t r0 = (int64_t)blockIdx.x * ROWS_CHUNK; r0 < n_dst_rows; r0 += (int64_t)gridDim.x * ROWS_CHUNK) {
        int64_t lo = 0, hi = n_idx;                          // wave-uniform lower bound of r0
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (idx[mid] < r0) lo = mid + 1; else hi = mid;
        }
        const int64_t mine = lo + lane < n_idx ? idx[lo + lane] : INT64_MAX;       // ascending across lanes
        const int64_t r1 = r0 + ROWS_CHUNK < n_dst_rows ? r0 + ROWS_CHUNK : n_dst_rows;
        for (int64_t r = r0; r < r1; r += ROWS_UNROLL) {     // ROWS_UNROLL rows per step: loads first, then stores
            const u32x4* s[ROWS_UNROLL];
#pragma unroll
            for (int k = 0; k < ROWS_UNROLL; ++k) {
                const int64_t rr = r + k;
                const uint64_t eq = __ballot(mine == rr), lt = __ballot(mine < rr);
                s[k] = (eq != 0 && rr < r1) ? reinterpret_cast<const u32x4*>(src + (lo + __popcll(lt)) * row_bytes) : nullptr;
            }
            for (int64_t c = threadIdx.x; c < pieces; c += ROWS_THREADS) {
                u32x4 v[ROWS_UNROLL];
#pragma unroll
                for (int k = 0; k < ROWS_UNROLL; ++k) v[k] = s[k] ? ld_stream(s[k] + c) : u32x4{0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < ROWS_UNROLL; ++k)
                    if (r + k < r1) st_stream(reinterpret_cast<u32x4*>(dst + (r + k) * row_bytes) + c, v[k]);
            }
        }
    }
}

static int rows_grid(int64_t n_rows, int rows_per_step) {
    const int64_t cap = 256 * 16;                            // 16 workgroups per CU in flight, then grid-stride
    const int64_t n = (n_rows + rows_per_step - 1) / rows_per_step;
    return (int)(n < cap ? (n > 0 ? n : 1) : cap);
}

int launch_gather_rows(const void* src, const int64_t* idx, void* dst, int64_t n_idx, int64_t row_bytes,
                       int64_t src_stride, int64_t n_src_rows, hipStream_t stream) {
    if (n_idx == 0 || row_bytes == 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows_grid(n_idx, ROWS_UNROLL)), dim3(ROWS_THREADS), 0, stream,
                       static_cast<const char*>(src), idx, static_cast<char*>(dst), n_idx, row_bytes, src_stride, n_src_rows);
    return 0;
}

int launch_scatter_rows(const void* src, const int64_t* idx, void* dst, int64_t n_idx, int64_t n_dst_rows,
                        int64_t row_bytes, int sorted_unique, hipStream_t stream) {
    if (n_dst_rows == 0 || row_bytes == 0) return 0;
    if (sorted_unique) {
        hipLaunchKernelGGL(scatter_rows_sorted_kernel, dim3(rows_grid(n_dst_rows, ROWS_CHUNK)), dim3(ROWS_THREADS), 0, stream,
                           static_cast<const char*>(src), idx, static_cast<char*>(dst), n_idx, row_bytes, n_dst_rows);
        return 0;
    }
    if (hipMemsetAsync(dst, 0, (size_t)(n_dst_rows * row_bytes), stream) != hipSuccess) return -1;
    if (n_idx == 0) return 0;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(rows_grid(n_idx, ROWS_UNROLL)), dim3(ROWS_THREADS), 0, stream,
                       static_cast<const char*>(src), idx, static_cast<char*>(dst), n_idx, row_bytes, n_dst_rows);
    return 0;
}

}  // namespace fa
