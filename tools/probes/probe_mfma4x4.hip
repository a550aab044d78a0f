// v_mfma_f32_4x4x4_16b_{f16,bf16} with A = ones: does every lane get the sum of ITS OWN four B values in all four
// result registers?  (16 blocks of 4 lanes; D[i][j] = sum_k A[i][k] B[k][j], lane 4 b + j holds B[0..3][j] and D[0..3][j].)
// Also its issue / pipe cost next to v_add_f32 (clock64 around 64 of each).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, float* o, long long* t) {
    const int l = threadIdx.x;
    h4 b, ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    for (int i = 0; i < 4; ++i) b[i] = (_Float16)in[l * 4 + i];
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) o[l * 8 + i] = c[i];
    // bf16 form
    s4 bb, ob;
    for (int i = 0; i < 4; ++i) { bb[i] = (short)(__builtin_bit_cast(unsigned, in[l * 4 + i]) >> 16); ob[i] = 0x3F80; }
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ob, bb, d, 0, 0, 0);
    for (int i = 0; i < 4; ++i) o[l * 8 + 4 + i] = d[i];
    // timing: 64 dependent-free MFMAs into 4 accumulators vs 64 v_add
    f4 a0 = c, a1 = c, a2 = c, a3 = c;
    long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, b, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, b, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, b, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, b, a3, 0, 0, 0);
    }
    long long t1 = clock64();
    float s0 = in[l], s1 = in[l + 1], s2 = in[l + 2], s3 = in[l + 3];
    long long t2 = clock64();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4"
                     : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(in[l]));
    }
    long long t3 = clock64();
    if (l == 0) { t[0] = t1 - t0; t[1] = t3 - t2; }
    o[512 + l] = a0[0] + a1[0] + a2[0] + a3[0] + s0 + s1 + s2 + s3;
}
int main() {
    float hin[260], ho[1024]; long long ht[2];
    for (int i = 0; i < 260; ++i) hin[i] = (float)((i * 37) % 19) * 0.25f - 2.0f;
    float *din, *dout; long long* dt;
    hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(ho)); hipMalloc(&dt, 16);
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dt);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost); hipMemcpy(ht, dt, 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const float want = hin[l * 4] + hin[l * 4 + 1] + hin[l * 4 + 2] + hin[l * 4 + 3];
        for (int i = 0; i < 8; ++i) if (ho[l * 8 + i] != want) { if (bad < 8) printf("lane %d reg %d: %g want %g\n", l, i, ho[l * 8 + i], want); ++bad; }
    }
    printf("4x4x4 ones-sum: %s (%d mismatches); 64 mfma_4x4x4: %lld clocks, 64 v_add_f32: %lld clocks (s_memtime units)\n", bad ? "DIFFERENT" : "own-lane sums in all four registers", bad, ht[0], ht[1]);
    return 0;
}
