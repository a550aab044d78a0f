// Probe: HBM read rate of an fp8 KV cache [tokens][32 heads][128 B] under two lane mappings (16-byte loads, nontemporal):
//   A "head-major workgroups": a wave instruction covers 8 tokens x 128 B of ONE head (rows 4 KiB apart)   <- the GEMV decode kernel
//   B "token-major":           a wave instruction covers 1 token x 8 heads = 1 KiB contiguous
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 32, ROW = 128;
template <int MODE>
__global__ void __launch_bounds__(256) k(const char* __restrict__ kv, int tokens_per_batch, int n_split, uint32_t* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  if (MODE == 0) {       // block = (batch, head, split); 4 waves take alternate 8-token groups
    const int id = blockIdx.x, split = id % n_split, head = (id / n_split) % H, b = id / (n_split * H);
    const int per = tokens_per_batch / n_split;
    const char* base = kv + ((size_t)b * tokens_per_batch + (size_t)split * per) * H * ROW + head * ROW;
    for (int t = wave * 8 + (lane >> 3); t < per; t += 32) {
      u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (size_t)t * H * ROW + (lane & 7) * 16));
      acc ^= x;
    }
  } else {               // block = (batch, split); wave w takes heads 8w..8w+7 of every token
    const int id = blockIdx.x, split = id % n_split, b = id / n_split;
    const int per = tokens_per_batch / n_split;
    const char* base = kv + ((size_t)b * tokens_per_batch + (size_t)split * per) * H * ROW + wave * 8 * ROW + lane * 16;
#pragma unroll 8
    for (int t = 0; t < per; ++t) {
      u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (size_t)t * H * ROW));
      acc ^= x;
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}
int main() {
  const int B = 128, T = 8192;
  const size_t bytes = (size_t)B * T * H * ROW;                  // 4.3 GB (one of K / V)
  char* kv; uint32_t* sink;
  (void)hipMalloc(&kv, bytes); (void)hipMalloc(&sink, 64);
  (void)hipMemset(kv, 1, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int ns : {1, 2, 4, 8, 16}) {
      const int grid = mode == 0 ? B * H * ns : B * ns;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, kv, T, ns, sink);
        else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, kv, T, ns, sink);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("%s splits %2d grid %6d: %.3f ms  %.0f GB/s\n", mode == 0 ? "A head-major (8 x 128 B per instr)" : "B token-major (1 KiB per instr)  ", ns, grid, best, bytes / best / 1e6);
    }
  return 0;
}
