// Probe: does the ORDER in which MFMAs visit their accumulators change the power draw (= the clock under the socket limit)?
// Chip-wide loop of v_mfma_f32_32x32x16_bf16 with random operands; the same 64 MFMAs per loop iteration, arranged as
//   rr8    : 8 accumulators round robin (every MFMA writes another accumulator than the one before)
//   rr4/rr2: 4 / 2 accumulators round robin
//   chain4 : 4 back-to-back MFMAs on one accumulator, then the next accumulator (srcC = the result just produced)
//   chain8 : 8 back-to-back
//   one    : a single accumulator throughout
// A back-to-back MFMA on the same accumulator has no wait states (the result is forwarded inside the matrix pipe); if the
// register file traffic of C / D (8 KiB per MFMA against 2 KiB for A and B) is skipped as well, chains draw less power.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8v __attribute__((ext_vector_type(8)));
#define MF(d) "v_mfma_f32_32x32x16_bf16 %" #d ", %8, %9, %" #d "\n"
#define REP8(x) x x x x x x x x
#define BODY(P, TXT)                                                                                          \
    if (PAT == P) asm volatile(REP8(TXT)                                                                      \
        : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6), "+a"(c7) : "v"(a), "v"(b));
template <int PAT>
__global__ void __launch_bounds__(256) k(float* sink, int iters) {
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
  s8v a, b;
  unsigned h = (threadIdx.x + 977u * blockIdx.x) * 2654435761u;
  for (int i = 0; i < 8; ++i) {      // pseudo-random bf16 operands: a in (-2, 2), b in (-1, 1)
    h = h * 1664525u + 1013904223u; a[i] = (short)(((h >> 16) & 0x807f) | 0x3f80);
    h = h * 1664525u + 1013904223u; b[i] = (short)(((h >> 16) & 0x807f) | 0x3f00);
  }
  for (int i = 0; i < iters; ++i) {
    BODY(0, MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7))
    BODY(1, MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3))
    BODY(2, MF(0) MF(1) MF(0) MF(1) MF(0) MF(1) MF(0) MF(1))
    BODY(3, MF(0) MF(0) MF(0) MF(0) MF(1) MF(1) MF(1) MF(1))
    BODY(4, MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0))
    if (PAT == 5) asm volatile(REP8(MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0)) : "+a"(c0) : "v"(c1), "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(c6), "v"(c7), "v"(a), "v"(b));
  }
  sink[threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0];
}
template <int P> void run(const char* name) {
  float* s; (void)hipMalloc(&s, 4096);
  const int iters = 20000;
  hipLaunchKernelGGL(k<P>, dim3(1024), dim3(256), 0, 0, s, iters);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f, sum = 0;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<P>, dim3(1024), dim3(256), 0, 0, s, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best; sum += ms;
  }
  const double flops = 1024.0 * 4 * iters * 64.0 * 32768.0;
  printf("%-8s random bf16 operands, chip-wide: %.3f ms (mean of 3: %.3f)  %.0f TFLOP/s\n", name, best, sum / 3, flops / (best * 1e-3) / 1e12);
}
int main() {
  run<0>("rr8"); run<1>("rr4"); run<2>("rr2"); run<3>("chain4"); run<4>("chain8"); run<0>("rr8"); run<3>("chain4");
  return 0;
}
