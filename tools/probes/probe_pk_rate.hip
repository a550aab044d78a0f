// Probe: issue cost of packed fp32 VALU (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) against the scalar forms, alone
// and inside the softmax instruction mix (per 2 elements: 2 fma + 2 exp + 2 add + 1 cvt_pk  vs  1 pk_fma + 2 exp + 1 pk_add + 1 cvt_pk).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
typedef float f2 __attribute__((ext_vector_type(2)));
template <int P>
__global__ void k(unsigned long long* out, float* sink, int iters) {
  f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  unsigned int pk = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (P == 0) asm volatile(REP8("v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %1, %1, %1, %2\n v_pk_fma_f32 %2, %2, %2, %3\n v_pk_fma_f32 %3, %3, %3, %4\n v_pk_fma_f32 %4, %4, %4, %5\n v_pk_fma_f32 %5, %5, %5, %6\n v_pk_fma_f32 %6, %6, %6, %7\n v_pk_fma_f32 %7, %7, %7, %0\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (P == 1) asm volatile(REP8("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %6\n v_pk_add_f32 %6, %6, %7\n v_pk_add_f32 %7, %7, %0\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if (P == 2) asm volatile(REP8("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %4\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %0\n")
                             : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x));
    // softmax mix, scalar: per pair  fma fma exp exp add add cvt   (a6 / a7 are the two row sums)
    if (P == 3) asm volatile(REP8(
        "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_exp_f32 %0, %0\n v_add_f32 %6, %6, %2\n v_exp_f32 %1, %1\n v_add_f32 %7, %7, %3\n v_cvt_pk_bf16_f32 %8, %2, %3\n"
        "v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_exp_f32 %2, %2\n v_add_f32 %6, %6, %0\n v_exp_f32 %3, %3\n v_add_f32 %7, %7, %1\n v_cvt_pk_bf16_f32 %8, %0, %1\n")
                             : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y), "+v"(a3.x), "+v"(a3.y), "+v"(pk));
    // softmax mix, packed: per pair  pk_fma exp exp pk_add cvt
    if (P == 4) asm volatile(REP8(
        "v_pk_fma_f32 v[10:11], v[10:11], v[14:15], v[16:17]\n v_exp_f32 v10, v10\n v_pk_add_f32 v[18:19], v[18:19], v[12:13]\n v_exp_f32 v11, v11\n v_cvt_pk_bf16_f32 v20, v12, v13\n"
        "v_pk_fma_f32 v[12:13], v[12:13], v[14:15], v[16:17]\n v_exp_f32 v12, v12\n v_pk_add_f32 v[18:19], v[18:19], v[10:11]\n v_exp_f32 v13, v13\n v_cvt_pk_bf16_f32 v20, v10, v11\n")
                             ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20");
    // packed with broadcast op_sel (scalar constant in the low dword of the pair)
    if (P == 5) asm volatile(REP8("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %3, %3, %1, %2 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %4, %4, %1, %2 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %5, %5, %1, %2 op_sel_hi:[1,0,0]\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
    // the same mixes in the shadow of independent 32x32x16 MFMAs (one MFMA per 4 element pairs = the D = 128 forward's ratio)
    if (P == 6) asm volatile(REP8(
        "v_mfma_f32_32x32x16_bf16 a[0:15], v[30:33], v[34:37], a[0:15]\n"
        "v_fma_f32 v10, v10, v14, v16\n v_fma_f32 v11, v11, v14, v16\n v_exp_f32 v10, v10\n v_add_f32 v18, v18, v12\n v_exp_f32 v11, v11\n v_add_f32 v19, v19, v13\n v_cvt_pk_bf16_f32 v20, v12, v13\n"
        "v_fma_f32 v12, v12, v14, v16\n v_fma_f32 v13, v13, v14, v16\n v_exp_f32 v12, v12\n v_add_f32 v18, v18, v10\n v_exp_f32 v13, v13\n v_add_f32 v19, v19, v11\n v_cvt_pk_bf16_f32 v20, v10, v11\n"
        "v_mfma_f32_32x32x16_bf16 a[16:31], v[30:33], v[34:37], a[16:31]\n"
        "v_fma_f32 v10, v10, v14, v16\n v_fma_f32 v11, v11, v14, v16\n v_exp_f32 v10, v10\n v_add_f32 v18, v18, v12\n v_exp_f32 v11, v11\n v_add_f32 v19, v19, v13\n v_cvt_pk_bf16_f32 v20, v12, v13\n"
        "v_fma_f32 v12, v12, v14, v16\n v_fma_f32 v13, v13, v14, v16\n v_exp_f32 v12, v12\n v_add_f32 v18, v18, v10\n v_exp_f32 v13, v13\n v_add_f32 v19, v19, v11\n v_cvt_pk_bf16_f32 v20, v10, v11\n")
                             ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v30","v31","v32","v33","v34","v35","v36","v37",
                                 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
    if (P == 7) asm volatile(REP8(
        "v_mfma_f32_32x32x16_bf16 a[0:15], v[30:33], v[34:37], a[0:15]\n"
        "v_pk_fma_f32 v[10:11], v[10:11], v[14:15], v[16:17]\n v_exp_f32 v10, v10\n v_pk_add_f32 v[18:19], v[18:19], v[12:13]\n v_exp_f32 v11, v11\n v_cvt_pk_bf16_f32 v20, v12, v13\n"
        "v_pk_fma_f32 v[12:13], v[12:13], v[14:15], v[16:17]\n v_exp_f32 v12, v12\n v_pk_add_f32 v[18:19], v[18:19], v[10:11]\n v_exp_f32 v13, v13\n v_cvt_pk_bf16_f32 v20, v10, v11\n"
        "v_mfma_f32_32x32x16_bf16 a[16:31], v[30:33], v[34:37], a[16:31]\n"
        "v_pk_fma_f32 v[10:11], v[10:11], v[14:15], v[16:17]\n v_exp_f32 v10, v10\n v_pk_add_f32 v[18:19], v[18:19], v[12:13]\n v_exp_f32 v11, v11\n v_cvt_pk_bf16_f32 v20, v12, v13\n"
        "v_pk_fma_f32 v[12:13], v[12:13], v[14:15], v[16:17]\n v_exp_f32 v12, v12\n v_pk_add_f32 v[18:19], v[18:19], v[10:11]\n v_exp_f32 v13, v13\n v_cvt_pk_bf16_f32 v20, v10, v11\n")
                             ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v30","v31","v32","v33","v34","v35","v36","v37",
                                 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
    if (P == 8) asm volatile(REP8(
        "v_mfma_f32_32x32x16_bf16 a[0:15], v[30:33], v[34:37], a[0:15]\n"
        "v_mfma_f32_32x32x16_bf16 a[16:31], v[30:33], v[34:37], a[16:31]\n")
                             ::: "v30","v31","v32","v33","v34","v35","v36","v37",
                                 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  sink[threadIdx.x] = s.x + s.y + (float)pk;
}
template <int P> void run(const char* name, int threads, int per_iter) {
  unsigned long long* d; float* s;
  (void)hipMalloc(&d, 8 * 1024); (void)hipMalloc(&s, 4096);
  const int iters = 2000;
  hipLaunchKernelGGL(k<P>, dim3(1), dim3(threads), 0, 0, d, s, iters);
  hipLaunchKernelGGL(k<P>, dim3(1), dim3(threads), 0, 0, d, s, iters);
  unsigned long long h;
  (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-44s threads %4d: %.2f ticks / instruction, %.1f ticks / element pair\n", name, threads, (double)h / (iters * (double)per_iter),
         P >= 3 && P <= 4 ? (double)h / (iters * 16.0) : 0.0);
}
int main() {
  for (int th : {64, 256, 512}) {
    run<0>("pk_fma only", th, 64); run<1>("pk_add only", th, 64); run<2>("fma only", th, 64);
    run<3>("softmax mix scalar (7 instr / pair)", th, 112); run<4>("softmax mix packed (5 instr / pair)", th, 80);
    run<5>("pk_fma op_sel_hi broadcast", th, 32);
    run<8>("MFMA only (ticks / MFMA)", th, 16); run<6>("MFMA + scalar mix, 2 pairs each (ticks / MFMA)", th, 16); run<7>("MFMA + packed mix, 2 pairs each (ticks / MFMA)", th, 16);
  }
  return 0;
}
