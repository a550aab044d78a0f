#include <hip/hip_runtime.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* a, const unsigned* b, float* o) {
  unsigned x = a[threadIdx.x], y = b[threadIdx.x];
  float r1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, x), __builtin_bit_cast(h2, y), 1.0f, false);
  float r2 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, x), __builtin_bit_cast(b2, y), 1.0f, false);
  float r3 = 1.0f;
  asm("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(r3) : "v"(x), "v"(y));
  o[threadIdx.x * 3] = r1; o[threadIdx.x * 3 + 1] = r2; o[threadIdx.x * 3 + 2] = r3;
}
int main() {
  unsigned ha[64], hb[64]; float ho[192];
  for (int i = 0; i < 64; ++i) { ha[i] = 0x40003c00u; hb[i] = 0x42004000u; }   // f16: (1, 2) . (2, 3) = 8 ; as bf16 pairs: other values
  unsigned *da, *db; float* dout;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 768);
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
  hipMemcpy(ho, dout, 768, hipMemcpyDeviceToHost);
  printf("fdot2 f16 builtin %g (expect 9)  bf16 builtin %g  asm f16 %g\n", ho[0], ho[1], ho[2]);
  return 0;
}
