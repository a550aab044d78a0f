// Measurement aid (NOT part of libfa_mi355.so): does data WRITTEN by one kernel stay in the 256 MB Infinity Cache (memory-side
// last-level cache, "MALL") for the NEXT kernel to read?  Background: the dS hand-off (profiles/r06_ds_handoff.txt) is break-even
// because its 2.15 GB of tiles go to HBM and come back; a producer / consumer schedule in chunks smaller than the cache would
// change that arithmetic - if the cache keeps written lines.
//   for N in sizes:   W(N): every workgroup streams its share of N bytes out (16 B per lane, whole lines)
//                     R(N): reads the same bytes back (and a second R right behind it: read-after-read)
//   reports ms and TB/s of R after W, R after R, and W itself.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/probes/probe_mall.hip -o tools/probes/probe_mall && tools/probes/probe_mall
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) wr(u32x4* p, size_t n16, unsigned v) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p[i] = u32x4{v, v + 1, v + 2, (unsigned)i};
}
__global__ void __launch_bounds__(256) rd(const u32x4* p, size_t n16, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) { const u32x4 x = p[i]; acc += x[0] ^ x[1] ^ x[2] ^ x[3]; }
    if (acc == 0x12345678u) sink[0] = acc;
}

static float ms_of(hipEvent_t a, hipEvent_t b) { float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
    const size_t MB = 1 << 20, maxb = 4096 * MB;
    u32x4* buf = nullptr; unsigned* sink = nullptr;
    if (hipMalloc(&buf, maxb) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e[4];
    for (auto& x : e) (void)hipEventCreate(&x);
    const int grid = 256 * 8;
    printf("%10s %12s %12s %12s   (TB/s)  W, R after W, R after R\n", "MiB", "W", "R_after_W", "R_after_R");
    const size_t sizes[] = {32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048, 4096};
    for (size_t s : sizes) {
        const size_t bytes = s * MB, n16 = bytes / 16;
        float w = 0, rw = 0, rr = 0;
        const int reps = 5;
        for (int r = 0; r < reps + 1; ++r) {
            // flush the caches of this buffer's lines: stream 1 GiB of another region through them
            hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, buf + (maxb - 1024 * MB) / 16, 1024 * MB / 16, sink);
            (void)hipEventRecord(e[0], 0);
            hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, buf, n16, (unsigned)r);
            (void)hipEventRecord(e[1], 0);
            hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, buf, n16, sink);
            (void)hipEventRecord(e[2], 0);
            hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, buf, n16, sink);
            (void)hipEventRecord(e[3], 0);
            (void)hipEventSynchronize(e[3]);
            if (r > 0) { w += ms_of(e[0], e[1]); rw += ms_of(e[1], e[2]); rr += ms_of(e[2], e[3]); }
        }
        w /= reps; rw /= reps; rr /= reps;
        printf("%10zu %8.4f ms %8.4f ms %8.4f ms   %6.2f %6.2f %6.2f\n", s, w, rw, rr, bytes / w / 1e9, bytes / rw / 1e9, bytes / rr / 1e9);
        fflush(stdout);
    }
    return 0;
}
