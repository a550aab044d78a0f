// Probe: VALU issue cost of v_exp_f32 next to plain VALU work (one wave per SIMD).
//   cycles per instruction for streams of independent instructions: all exp, all fma, exp : fma mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int P>
__global__ void k(unsigned long long* out, float* sink, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (P == 0) asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (P == 1) asm volatile(REP8("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %4\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %0\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (P == 2) asm volatile(REP8("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %1\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %5\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (P == 3) asm volatile(REP8("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %2\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %3, %1\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %5, %6\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %7, %5\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (P == 4) asm volatile(REP8("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %4\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %1\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (P == 5) asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %4\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %2\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int P> void run(const char* name, int threads) {
  unsigned long long* d; float* s;
  (void)hipMalloc(&d, 8 * 1024); (void)hipMalloc(&s, 4096);
  const int iters = 2000;
  hipLaunchKernelGGL(k<P>, dim3(1), dim3(threads), 0, 0, d, s, iters);
  hipLaunchKernelGGL(k<P>, dim3(1), dim3(threads), 0, 0, d, s, iters);
  unsigned long long h;
  (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-34s threads %4d: %.2f clock ticks / instruction\n", name, threads, (double)h / (iters * 64.0));
}
int main() {
  for (int th : {64, 256, 512}) {
    run<0>("exp only", th); run<1>("fma only", th); run<2>("exp : fma = 1 : 3", th); run<3>("exp : fma = 1 : 1", th);
    run<4>("exp : fma = 1 : 7", th); run<5>("exp exp fma x6", th);
  }
  return 0;
}
