// Probe: is the scalar soffset of a raw buffer load included in the hardware range check?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* base, unsigned* out, int num_bytes, int soff) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, num_bytes, 0x00020000);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, soff, 0);
  out[threadIdx.x * 4 + 0] = v[0]; out[threadIdx.x * 4 + 1] = v[1];
  out[threadIdx.x * 4 + 2] = v[2]; out[threadIdx.x * 4 + 3] = v[3];
}
int main() {
  const int N = 4096;
  std::vector<unsigned> h(N);
  for (int i = 0; i < N; ++i) h[i] = 0x1000 + i;
  unsigned *d, *o;
  hipMalloc(&d, N * 4); hipMalloc(&o, 64 * 16);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> ho(256);
  // num_records = 1024 bytes (256 dwords). 64 lanes x 16 B = 1024 B per load.
  for (int soff : {0, 512, 1024, 2048}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 1024, soff);
    hipMemcpy(ho.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("soff=%4d: lane0=%x lane31=%x lane32=%x lane63=%x\n", soff, ho[0], ho[31 * 4], ho[32 * 4], ho[63 * 4]);
  }
  return 0;
}
