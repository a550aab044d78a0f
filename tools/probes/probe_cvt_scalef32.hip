// Probe: semantics of gfx950's v_cvt_scalef32_pk_{f16,bf16}_fp8 (2 x fp8-e4m3 -> 2 x 16-bit, one VALU op).
// Question: is the f32 `scale` operand a full multiplier or only its exponent, and is the result
// the exact fp8 value (x scale) for all 256 byte patterns?  Compared against v_cvt_pk_f32_fp8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float* ref, float* f16o, float* bf16o, float scale) {
    const unsigned b = threadIdx.x;                 // byte pattern 0..255
    const unsigned x = b | (b << 8) | (b << 16) | (b << 24);
    const f2 r = __builtin_amdgcn_cvt_pk_f32_fp8(x, false);
    const h2 a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x, scale, false);
    const h2 a2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(x, scale, true);
    const b2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x, scale, true);
    ref[b] = r[0];
    f16o[b] = (float)a[0] + 0.f * (float)a2[1];
    bf16o[b] = (float)c[1];
}
int main() {
    float *ref, *f16o, *bf16o;
    hipMallocManaged(&ref, 1024); hipMallocManaged(&f16o, 1024); hipMallocManaged(&bf16o, 1024);
    const float scales[] = {1.0f, 2.0f, 0.5f, 3.0f, 1.5f, 0.0078125f};
    for (float s : scales) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, ref, f16o, bf16o, s);
        hipDeviceSynchronize();
        int bad_full = 0, bad_exp = 0, badb_full = 0;
        const float s_exp = std::ldexp(1.0f, std::ilogb(s));
        for (int b = 0; b < 256; ++b) {
            if (std::isnan(ref[b])) continue;
            if (f16o[b] != ref[b] * s) bad_full++;
            if (f16o[b] != ref[b] * s_exp) bad_exp++;
            if (bf16o[b] != ref[b] * s) badb_full++;
        }
        printf("scale %g: f16 mismatches vs full multiply %d, vs exponent-only %d; bf16 vs full %d  (sample b=0x3a: ref %g f16 %g bf16 %g)\n",
               s, bad_full, bad_exp, badb_full, ref[0x3a], f16o[0x3a], bf16o[0x3a]);
    }
    return 0;
}
