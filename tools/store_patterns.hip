// Store-pattern micro-benchmark behind mask_logits / the fused mask writers: the same bytes (E rows of ROW bytes, 16-byte stores, one
// kilobyte per wavefront and store instruction) written in different orders.  hipcc -O3 --offload-arch=gfx950 tools/store_patterns.hip -o tools/store_patterns
//   P0  fill-like: workgroup b writes bytes [b * 16 KB, +16 KB), four wavefronts x four interleaved kilobytes
//   P1  a wavefront per row, kilobytes in order                      (mask_logits, obs_small_kernel)
//   P2  P1, each wavefront starting at a different kilobyte of its row (rotated by 7 x row index)
//   P3  four wavefronts per row, kilobytes interleaved
//   P4  P1 with rows visited in a strided order (row = (w * 977) mod E): neighbours in time are far apart in memory
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void p0(uint4* out, size_t n16) {
    const size_t base = (size_t)blockIdx.x * 1024u + threadIdx.x;
    const uint4 v = make_uint4(1, 2, 3, 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const size_t k = base + (size_t)i * 256u; if (k < n16) out[k] = v; }
}
template <int MODE>
__global__ __launch_bounds__(256) void rows(uint4* out, uint32_t E, uint32_t row16) {     // row16: 16-byte groups per row
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint4 v = make_uint4(1, 2, 3, 4);
    const uint32_t spans = (row16 + 63u) / 64u;
    if (MODE == 3) {
        const uint32_t e = blockIdx.x;
        if (e >= E) return;
        uint4* row = out + (size_t)e * row16;
        for (uint32_t s = wv; s < spans; s += 4u) { const uint32_t g = s * 64u + lane; if (g < row16) row[g] = v; }
        return;
    }
    uint32_t e = blockIdx.x * 4u + wv;
    if (e >= E) return;
    if (MODE == 4) e = (uint32_t)(((uint64_t)e * 977u) % E);
    uint4* row = out + (size_t)e * row16;
    const uint32_t rot = MODE == 2 ? (e * 7u) % spans : 0u;
    for (uint32_t i = 0; i < spans; ++i) {
        uint32_t s = i + rot; if (s >= spans) s -= spans;
        const uint32_t g = s * 64u + lane;
        if (g < row16) row[g] = v;
    }
}

int main(int argc, char** argv) {
    const uint32_t E = argc > 1 ? atoi(argv[1]) : 65536, rowb = argc > 2 ? atoi(argv[2]) : 56688;
    const uint32_t row16 = rowb / 16;
    const size_t n16 = (size_t)E * row16;
    uint4* buf;
    CHECK(hipMalloc(&buf, n16 * 16));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CHECK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms * 100.0, n16 * 16.0 / (ms * 1e-4) / 1e12);
    };
    printf("E = %u rows of %u bytes (%.2f GB)\n", E, rowb, n16 * 16.0 / 1e9);
    time("P0 fill-like, 16 KB per workgroup", [&] { hipLaunchKernelGGL(p0, dim3((n16 + 1023) / 1024), dim3(256), 0, 0, buf, n16); });
    time("P1 wavefront per row", [&] { hipLaunchKernelGGL(rows<1>, dim3((E + 3) / 4), dim3(256), 0, 0, buf, E, row16); });
    time("P2 wavefront per row, rotated start", [&] { hipLaunchKernelGGL(rows<2>, dim3((E + 3) / 4), dim3(256), 0, 0, buf, E, row16); });
    time("P3 four wavefronts per row", [&] { hipLaunchKernelGGL(rows<3>, dim3(E), dim3(256), 0, 0, buf, E, row16); });
    time("P4 wavefront per row, rows in strided order", [&] { hipLaunchKernelGGL(rows<4>, dim3((E + 3) / 4), dim3(256), 0, 0, buf, E, row16); });
    CHECK(hipFree(buf));
    return 0;
}
