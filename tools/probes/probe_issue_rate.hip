// Probe: instruction issue cost by type, alone and in the shadow of MFMAs, for one wave per SIMD (256 threads) and two
// (512).  Clock ticks per GROUP (one MFMA + its companions) or per instruction for the plain streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8v __attribute__((ext_vector_type(8)));
#define MF(d) "v_mfma_f32_32x32x16_bf16 %" #d ", %8, %9, %" #d "\n"
#define F3 "v_fma_f32 %10, %10, %10, %11\n v_fma_f32 %11, %11, %11, %12\n v_fma_f32 %12, %12, %12, %10\n"
#define P3 "v_pk_fma_f32 %13, %13, %13, %14\n v_pk_fma_f32 %14, %14, %14, %13\n v_pk_fma_f32 %13, %13, %14, %14\n"
#define S4 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
#define N4 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
#define W4 "s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n"
#define E2 "v_exp_f32 %10, %10\n v_exp_f32 %11, %11\n"
#define C2 "v_cvt_pk_bf16_f32 %10, %11, %12\n v_cvt_pk_bf16_f32 %11, %12, %10\n"
#define L2 "ds_read_b128 %15, %16\n ds_read_b128 %15, %16 offset:4096\n"
#define BODY(P, TXT)                                                                                                   \
    if (PAT == P) asm volatile(REP8(TXT)                                                                               \
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)                             \
        : "v"(a), "v"(b), "v"(x0), "v"(x1), "v"(x2), "v"(p0), "v"(p1), "v"(ld), "v"(addr) : "s20", "s21", "s22", "s23", "memory");
template <int PAT>
__global__ void __launch_bounds__(512) k(unsigned long long* out, float* sink, int iters, int rnd = 0) {
  __shared__ float lds[4096];
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
  s8v a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  if (rnd) {   // pseudo-random bf16 operands in (-2, 2) (data toggling sets the MFMA power draw)
    unsigned h = (threadIdx.x + 977u * blockIdx.x) * 2654435761u;
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; a[i] = (short)(((h >> 16) & 0x807f) | 0x3f80);
      h = h * 1664525u + 1013904223u; b[i] = (short)(((h >> 16) & 0x807f) | 0x3f00);
    }
  }
  float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f;
  double p0 = threadIdx.x, p1 = 3.0;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 ld = {};
  unsigned addr = (threadIdx.x & 63) * 16;
  lds[threadIdx.x] = 0.f;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    BODY(0, MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7))
    BODY(1, F3 F3 F3 F3 F3 F3 F3 F3)
    BODY(2, P3 P3 P3 P3 P3 P3 P3 P3)
    BODY(3, S4 S4 S4 S4 S4 S4)
    BODY(4, N4 N4 N4 N4 N4 N4)
    BODY(5, W4 W4 W4 W4 W4 W4)
    BODY(6, MF(0) F3 MF(1) F3 MF(2) F3 MF(3) F3)
    BODY(7, MF(0) F3 F3 MF(1) F3 F3 MF(2) F3 F3 MF(3) F3 F3)
    BODY(8, MF(0) S4 MF(1) S4 MF(2) S4 MF(3) S4)
    BODY(9, MF(0) S4 S4 MF(1) S4 S4 MF(2) S4 S4 MF(3) S4 S4)
    BODY(10, MF(0) F3 S4 MF(1) F3 S4 MF(2) F3 S4 MF(3) F3 S4)
    BODY(11, MF(0) P3 MF(1) P3 MF(2) P3 MF(3) P3)
    BODY(12, MF(0) P3 P3 MF(1) P3 P3 MF(2) P3 P3 MF(3) P3 P3)
    BODY(13, MF(0) E2 F3 MF(1) E2 F3 MF(2) E2 F3 MF(3) E2 F3)
    BODY(14, MF(0) L2 MF(1) L2 MF(2) L2 MF(3) L2)
    BODY(15, MF(0) L2 F3 MF(1) L2 F3 MF(2) L2 F3 MF(3) L2 F3)
    BODY(16, MF(0) L2 F3 S4 MF(1) L2 F3 S4 MF(2) L2 F3 S4 MF(3) L2 F3 S4)
    BODY(17, MF(0) N4 MF(1) N4 MF(2) N4 MF(3) N4)
    BODY(18, MF(0) W4 MF(1) W4 MF(2) W4 MF(3) W4)
    BODY(19, MF(0) C2 C2 MF(1) C2 C2 MF(2) C2 C2 MF(3) C2 C2)
    BODY(20, L2 L2 L2 L2)
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0] + x0 + x1 + x2 + (float)p0 + (float)p1 + ld[0];
}
// MFMA operand files: C/D in VGPRs or AGPRs, A / B from VGPRs or AGPRs (8 independent accumulators, one wave per SIMD)
template <int MODE>
__global__ void __launch_bounds__(256) kop(unsigned long long* out, float* sink, int iters) {
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
  s8v a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#define MFS(A_, B_, C_) asm volatile(REP8(MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7)) \
        : "+" C_(c0), "+" C_(c1), "+" C_(c2), "+" C_(c3), "+" C_(c4), "+" C_(c5), "+" C_(c6), "+" C_(c7) : A_(a), B_(b));
    if (MODE == 0) { MFS("v", "v", "v") }
    if (MODE == 1) { MFS("a", "v", "v") }
    if (MODE == 2) { MFS("a", "a", "v") }
    if (MODE == 3) { MFS("v", "v", "a") }
    if (MODE == 4) { MFS("a", "v", "a") }
    if (MODE == 5) { MFS("a", "a", "a") }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0];
}
template <int MODE> void run_op(const char* name) {
  unsigned long long* d; float* s;
  (void)hipMalloc(&d, 8 * 1024); (void)hipMalloc(&s, 4096);
  const int iters = 2000;
  hipLaunchKernelGGL(kop<MODE>, dim3(1), dim3(256), 0, 0, d, s, iters);
  hipLaunchKernelGGL(kop<MODE>, dim3(1), dim3(256), 0, 0, d, s, iters);
  unsigned long long h;
  (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("mfma operands %-28s: %6.2f ticks per mfma\n", name, (double)h / (iters * 64.0));
}
template <int P> void run(const char* name, int per_iter) {
  unsigned long long* d; float* s;
  (void)hipMalloc(&d, 8 * 1024); (void)hipMalloc(&s, 4096);
  const int iters = 500;
  for (int th : {256, 512}) {
    hipLaunchKernelGGL(k<P>, dim3(1), dim3(th), 0, 0, d, s, iters);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<P>, dim3(1), dim3(th), 0, 0, d, s, iters * 20);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h;
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s waves/SIMD %d: %8.2f ticks per unit (%d units per loop)  [%.0f ticks/us, %.2f ns per unit]\n", name, th / 256,
           (double)h / (iters * 20 * 8.0 * per_iter), per_iter, (double)h / (ms * 1e3), ms * 1e6 / (iters * 20 * 8.0 * per_iter));
  }
}
template <int P> void run_full(const char* name, int per_iter, int rnd) {
  // every CU busy: does the shader clock hold under chip-wide MFMA load?  (ticks/us = effective clock in MHz)
  unsigned long long* d; float* s;
  (void)hipMalloc(&d, 8 * 8192); (void)hipMalloc(&s, 4096);
  const int iters = 20000;
  hipLaunchKernelGGL(k<P>, dim3(1024), dim3(256), 0, 0, d, s, iters, rnd);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<P>, dim3(1024), dim3(256), 0, 0, d, s, iters, rnd);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1024];
  (void)hipMemcpy(h, d, 8 * 1024, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 1024; ++i) avg += (double)h[i]; avg /= 1024;
  // 1024 blocks over 256 CUs, one block per CU at a time (launch bounds 512 threads... the register use lets several
  // blocks share a CU: the per-block tick count and the wall time give the clock only together with the residency)
  printf("%-30s %s operands, chip-wide: %.2f ticks per unit per block, kernel %.3f ms, sum of block ticks / wall = %.0f MHz x blocks-in-flight/CU\n",
         name, rnd ? "random bf16" : "constant", avg / (iters * 8.0 * per_iter), ms, avg * 1024 / 256 / (ms * 1e3));
  double flops = 1024.0 * 4 * iters * 8.0 * per_iter * 32768.0;
  printf("   MFMA rate: %.0f TFLOP/s dense bf16\n", flops / (ms * 1e-3) / 1e12);
}
int main() {
  run_op<0>("A v, B v, C/D v"); run_op<1>("A a, B v, C/D v"); run_op<2>("A a, B a, C/D v");
  run_op<3>("A v, B v, C/D a"); run_op<4>("A a, B v, C/D a"); run_op<5>("A a, B a, C/D a");
  run_full<0>("mfma only", 8, 0);
  run_full<0>("mfma only", 8, 1);
  run_full<6>("mfma + 3 fma", 4, 0);
  run_full<6>("mfma + 3 fma", 4, 1);
  run<0>("mfma only (unit = mfma)", 8);
  run<1>("v_fma_f32 (unit = instr)", 24);
  run<2>("v_pk_fma_f32 (unit = instr)", 24);
  run<3>("s_add_u32 (unit = instr)", 24);
  run<4>("s_nop 0 (unit = instr)", 24);
  run<5>("s_waitcnt (unit = instr)", 24);
  run<20>("ds_read_b128 (unit = instr)", 8);
  run<6>("mfma + 3 fma", 4);
  run<7>("mfma + 6 fma", 4);
  run<8>("mfma + 4 salu", 4);
  run<9>("mfma + 8 salu", 4);
  run<10>("mfma + 3 fma + 4 salu", 4);
  run<11>("mfma + 3 pk_fma", 4);
  run<12>("mfma + 6 pk_fma", 4);
  run<13>("mfma + 2 exp + 3 fma", 4);
  run<14>("mfma + 2 ds_read_b128", 4);
  run<15>("mfma + 2 ds_read_b128 + 3 fma", 4);
  run<16>("mfma + 2 ds_read_b128 + 3 fma + 4 salu", 4);
  run<17>("mfma + 4 s_nop", 4);
  run<18>("mfma + 4 s_waitcnt", 4);
  run<19>("mfma + 4 cvt_pk", 4);
  return 0;
}
