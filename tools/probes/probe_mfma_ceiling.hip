// Measurement aid (NOT part of libfa_mi355.so): the chip-wide bf16 MFMA rate this socket sustains on RANDOM operands.
// The 2.5 PFLOP/s dense peak the roofline is priced against assumes 2.4 GHz; under matrix load with real data the part
// sits at its power limit and clocks lower (profiles/r03_clock_power.txt), so bench.py runs this loop for ~1 s right after
// the timed steps and reports the result as roofline.practical_ceiling next to the nominal peak.
//   one wave per SIMD x 8 independent accumulator chains of v_mfma_f32_32x32x16_bf16, nothing else in the loop;
//   operands: pseudo-random bf16 in (-2, 2) per lane (data toggling sets the MFMA power draw; constant operands: `rnd` = 0).
// C ABI:  int fa_probe_mfma(int launches, int iters, int rnd, float* ms_total, double* flops_total)
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256, 1) mfma_loop(float* sink, int iters, int rnd) {
    f32x16 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    s16x8 a = {0x3f80, 0x4000, 0x3f00, 0x3fc0, 0x3f80, 0x4000, 0x3f00, 0x3fc0}, b = a;
    if (rnd) {
        unsigned h = (threadIdx.x + 977u * blockIdx.x) * 2654435761u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u; a[i] = (short)(((h >> 16) & 0x807f) | 0x3f80);
            h = h * 1664525u + 1013904223u; b[i] = (short)(((h >> 16) & 0x807f) | 0x3f00);
        }
    }
    const bf16x8 av = __builtin_bit_cast(bf16x8, a), bv = __builtin_bit_cast(bf16x8, b);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i][0];
    if (s == 12345.678f) sink[threadIdx.x] = s;      // (keeps the chains alive)
}

extern "C" int fa_probe_mfma(int launches, int iters, int rnd, float* ms_total, double* flops_total) {
    float* sink = nullptr;
    if (hipMalloc(&sink, 4096) != hipSuccess) return -1;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = cus * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, sink, iters, rnd);      // warm-up
    (void)hipEventRecord(e0, 0);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, sink, iters, rnd);
    (void)hipEventRecord(e1, 0);
    const int rc = (int)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(ms_total, e0, e1);
    *flops_total = (double)launches * blocks * 4.0 * (double)iters * 8.0 * 32768.0;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    return rc;
}
