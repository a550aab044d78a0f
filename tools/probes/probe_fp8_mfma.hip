// (1) ds_read_b64_tr_b8: which LDS byte lands in which (lane, byte) - every lane passes its own address (lane * 8 in a 512-byte
//     region), the region holds byte = (address & 255) in pass A and (address >> 8) | marker in pass B.
// (2) v_mfma_f32_32x32x16_fp8_fp8 operand layout: assumed lane (i = l & 31, g = l >> 5) holds A[i][8 g + b] in byte b of its
//     64-bit operand, B[8 g + b][j = l & 31] likewise; C in the usual 32 x 32 map.  Checked against a host product of small
//     integers (exact in e4m3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __host__ inline uint8_t e4m3_of_small(int v) {   // v in -8..8 -> OCP e4m3fn code (exact)
    if (v == 0) return 0;
    const int s = v < 0; int a = s ? -v : v;
    int e = 0; while ((1 << (e + 1)) <= a) ++e;             // a in [2^e, 2^(e+1))
    const int m = ((a << 3) >> e) & 7;                       // 3 mantissa bits (a <= 8 -> exact for a = 1..8 except 5,7? 5 = 1.25 * 4 ok, 7 = 1.75 * 4 ok)
    return (uint8_t)((s << 7) | ((e + 7) << 3) | m);
}
__global__ void k(uint8_t* out_tr, float* out_c) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) { lds[i] = (uint8_t)(i & 255); lds[512 + i] = (uint8_t)(i >> 8); }
    __syncthreads();
    typedef __attribute__((address_space(3))) i32x2* lp;
    i32x2 a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(lds + l * 8));
    i32x2 b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(lds + 512 + l * 8));
    for (int j = 0; j < 8; ++j) {
        out_tr[l * 16 + j] = (uint8_t)(((j < 4 ? a[0] : a[1]) >> (8 * (j & 3))) & 255);
        out_tr[l * 16 + 8 + j] = (uint8_t)(((j < 4 ? b[0] : b[1]) >> (8 * (j & 3))) & 255);
    }
    // ---- MFMA layout ----
    const int i = l & 31, g = l >> 5;
    uint64_t av = 0, bv = 0;
    for (int bb = 0; bb < 8; ++bb) {
        const int kk = 8 * g + bb;
        av |= (uint64_t)e4m3_of_small(((i * 3 + kk * 5) % 9) - 4) << (8 * bb);      // A[i][kk]
        bv |= (uint64_t)e4m3_of_small(((i * 7 + kk * 2) % 7) - 3) << (8 * bb);      // B[kk][j = i]
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8((long)av, (long)bv, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out_c[l * 16 + r] = c[r];
}
int main() {
    uint8_t* dtr; float* dc; uint8_t htr[1024]; float hc[1024];
    hipMalloc(&dtr, 1024); hipMalloc(&dc, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dtr, dc);
    hipMemcpy(htr, dtr, 1024, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, 4096, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b8: result byte j of lane l <- (source lane, source byte) [each lane's 8 bytes sit at lane * 8]\n");
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 8; ++j) { const int addr = htr[l * 16 + j] | (htr[l * 16 + 8 + j] << 8); printf(" (%2d,%d)", addr >> 3, addr & 7); }
        printf("\n");
    }
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float want = 0.f;
        for (int kk = 0; kk < 16; ++kk) want += (float)(((row * 3 + kk * 5) % 9) - 4) * (float)(((col * 7 + kk * 2) % 7) - 3);
        if (hc[l * 16 + r] != want) { if (bad < 6) printf("C[%d][%d]: %g want %g\n", row, col, hc[l * 16 + r], want); ++bad; }
    }
    printf("mfma_f32_32x32x16_fp8_fp8 with byte b of lane (i, g) = A[i][8 g + b] / B[8 g + b][j]: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as assumed", bad);
    return 0;
}
