// Probe: buffer_load ... lds (LDS-DMA): lane-linear destination, and what out-of-range lanes write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* base, unsigned* out, int num_bytes, int soff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 512; i += 64) ((unsigned*)smem)[i] = 0xdead0000u + i;   // poison
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, num_bytes, 0x00020000);
  // permuted source: lane l loads chunk (l ^ 1)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem), 16, (threadIdx.x ^ 1) * 16, soff, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 1024), 16, threadIdx.x * 16, soff + 512, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
  const int N = 4096;
  std::vector<unsigned> h(N);
  for (int i = 0; i < N; ++i) h[i] = 0x1000 + i;
  unsigned *d, *o;
  (void)hipMalloc(&d, N * 4); (void)hipMalloc(&o, 512 * 4);
  (void)hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> ho(512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 2048, 0, d, o, 1024, 0);   // num_records = 1024 B
  (void)hipMemcpy(ho.data(), o, 2048, hipMemcpyDeviceToHost);
  printf("first load (lane^1 permuted source): lds dword[0]=%x [4]=%x [8]=%x [252]=%x\n", ho[0], ho[4], ho[8], ho[252]);
  printf("second load (soffset 512 -> lanes >= 32 out of range): lds dword[256]=%x [256+4*31]=%x [256+4*32]=%x [256+4*63]=%x\n",
         ho[256], ho[256 + 124], ho[256 + 128], ho[256 + 252]);
  return 0;
}
