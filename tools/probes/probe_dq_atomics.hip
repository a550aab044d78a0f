// Measurement aid (NOT part of libfa_mi355.so): what would it cost to sum dQ out of the dK/dV kernel with fp32 atomics?
// (round-5 review, task 1a)  The generated dK/dV kernel's launch shape at BASELINE config 2 is re-created
// without its arithmetic: 128 (batch, head) units x 16 mirrored key-block pairs = 2048 workgroups of four waves; a workgroup
// walks the 32-row query stages its two 128-key blocks see under the causal mask (132 stages per pair) and at every stage adds a
// 32 x 128 fp32 partial to dQ[unit][rows] - wave w owns the 32 columns 32w.. of the partial (the form a fifth GEMM
// dQ_partial = dS K with the contraction over the workgroup's 128 keys would leave: 16 accumulator registers per wave), i.e.
// 16 no-return global_atomic_add_f32 per wave and stage, each covering two rows x 128 contiguous bytes.
//   total: 128 units x 16 pairs x 132 stages x 16 KiB = 4.43 GB of atomic operands onto a 268 MB fp32 dQ buffer.
// Modes (bit field):  1 = atomics, 2 = MFMA stream (MF MFMAs per wave and stage, random operands), 4 = `sc1` (system scope)
//                     8 = plain stores instead of atomics (the floor of the same write pattern), 16 = scattered placement
//                     (workgroup id -> unit without the unit-per-XCD rule: the eight XCDs all add to every unit)
// C ABI:  int fa_probe_dq_atomics(int mode, int mfmas_per_stage, int launches, int verify, float* ms_mean, double* bad_frac)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

static constexpr int kS = 4096, kD = 128, kUnits = 128, kPairs = 16, kKB = 32;

__device__ __forceinline__ void atom_add(float* p, float v) { asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void atom_add_sc1(float* p, float v) { asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void plain_store(float* p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }

template <int MODE, int MF8>
__global__ void __launch_bounds__(256, 1) dq_atomics_kernel(float* __restrict__ dq, int verify, float* sink) {
    const int id = blockIdx.x;
    int unit, pair;
    if (MODE & 16) { unit = id % kUnits; pair = id / kUnits; }                       // scattered: a unit's workgroups on all XCDs
    else { const int xcd = id & 7, j = id >> 3; unit = (j / kPairs) * 8 + xcd; pair = j % kPairs; }   // unit u on XCD u % 8
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 5, n = lane & 31;
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    s16x8 a, b;
    unsigned h = (threadIdx.x + 977u * blockIdx.x) * 2654435761u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u; a[i] = (short)(((h >> 16) & 0x807f) | 0x3f80);
        h = h * 1664525u + 1013904223u; b[i] = (short)(((h >> 16) & 0x807f) | 0x3c00);
    }
    const bf16x8 av = __builtin_bit_cast(bf16x8, a), bv = __builtin_bit_cast(bf16x8, b);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.5f;
    float* base = dq + (size_t)unit * kS * kD + 32 * w + n;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int kb = half ? (kKB - 1 - pair) : pair;
#pragma unroll 1
        for (int st = kb * 4; st < kS / 32; ++st) {
            // eight groups per stage: mf / 8 MFMAs, then two of the 16 adds - of the PREVIOUS stage's values (`t`), as a
            // software-pipelined kernel body would place them: the adds never wait for the MFMAs issued beside them
            float* p = base + (size_t)(st * 32 + 4 * g) * kD;
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {
                if (MODE & 2) {
#pragma unroll
                    for (int m = 0; m < MF8; ++m) { const int ch = (grp * MF8 + m) & 3; c[ch] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c[ch], 0, 0, 0); }
                }
                if (MODE & (1 | 8)) {
#pragma unroll
                    for (int r = 2 * grp; r < 2 * grp + 2; ++r) {
                        const float v = verify ? 1.0f : t[r];
                        float* q = p + ((r & 3) + 8 * (r >> 2)) * kD;
                        if (MODE & 8) plain_store(q, v);
                        else if (MODE & 4) atom_add_sc1(q, v);
                        else atom_add(q, v);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);          // keep the groups as written: MFMAs, two adds, MFMAs, ...
            }
            if (MODE & 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = c[r & 3][r];
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c[i][0];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int MODE>
static void launch(float* dq, int mf, int verify, float* sink) {
    if (mf == 40) hipLaunchKernelGGL((dq_atomics_kernel<MODE, 5>), dim3(kUnits * kPairs), dim3(256), 0, 0, dq, verify, sink);
    else if (mf == 32) hipLaunchKernelGGL((dq_atomics_kernel<MODE, 4>), dim3(kUnits * kPairs), dim3(256), 0, 0, dq, verify, sink);
    else { fprintf(stderr, "probe_dq_atomics: 32 or 40 MFMAs per stage\n"); abort(); }
}

static void launch_mode(int mode, float* dq, int mf, int verify, float* sink) {
    switch (mode) {
#define C(M) case M: launch<M>(dq, mf, verify, sink); break;
        C(1) C(2) C(3) C(5) C(7) C(8) C(10) C(17) C(19) C(21) C(23)
#undef C
        default: fprintf(stderr, "probe_dq_atomics: mode %d not instantiated\n", mode); abort();
    }
}

extern "C" int fa_probe_dq_atomics(int mode, int mfmas_per_stage, int launches, int verify, float* ms_mean, double* bad_frac) {
    float *dq = nullptr, *sink = nullptr;
    const size_t n = (size_t)kUnits * kS * kD;
    if (hipMalloc(&dq, n * 4) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) return -1;
    (void)hipMemset(dq, 0, n * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    *bad_frac = -1.0;
    if (verify) {                                        // every add is 1.0: row block i of a unit must end at (i / 4 + 1) x launches(=1)
        launch_mode(mode, dq, mfmas_per_stage, 1, sink);
        if (hipDeviceSynchronize() != hipSuccess) return -2;
        float* hbuf = (float*)malloc(n * 4);
        (void)hipMemcpy(hbuf, dq, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t u = 0; u < (size_t)kUnits; ++u)
            for (int row = 0; row < kS; ++row) {
                const float want = (mode & 8) ? 1.0f : (float)(row / 128 + 1);
                const float* r = hbuf + (u * kS + row) * kD;
                for (int cidx = 0; cidx < kD; ++cidx) bad += (r[cidx] != want);
            }
        *bad_frac = (double)bad / (double)n;
        free(hbuf);
        (void)hipMemset(dq, 0, n * 4);
    }
    launch_mode(mode, dq, mfmas_per_stage, 0, sink);     // warm-up
    (void)hipEventRecord(e0, 0);
    for (int l = 0; l < launches; ++l) launch_mode(mode, dq, mfmas_per_stage, 0, sink);
    (void)hipEventRecord(e1, 0);
    const int rc = (int)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_mean = ms / launches;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(dq); (void)hipFree(sink);
    return rc;
}
