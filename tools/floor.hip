// Launch-floor probe for the step kernel's batch shape (not part of the product).
//
// Times, by hipGraph replay of 1000 dependent launches each, (1) an empty kernel, (2) a kernel that only moves the
// step kernel's per-env traffic (read 20 B action + 32 B header + 2 x 16 B rows at action-dependent addresses, write
// them back + 8 B results) with no simulation logic, at the same grid as mcbs::step_kernel (one lane per env,
// 256-thread workgroups).  The gap between (2) and the real kernel is what the simulation logic costs; the gap between
// (1) and (2) is what the two dependent memory round trips cost at this batch size.
//
//   hipcc -O3 --offload-arch=gfx950 tools/floor.hip -o tools/floor && tools/floor [envs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 0; }

// header columns: 8 x u32 per env, env-fastest; rows: 12 x 16 B per env (AoS); actions 5 x i32; out 8 B
__global__ __launch_bounds__(256) void traffic_kernel(uint32_t* __restrict__ hdr, uint4* __restrict__ rows,
                                                      const int* __restrict__ act, uint2* __restrict__ out, int E) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    uint32_t h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = hdr[(size_t)k * E + e];
    int a0 = act[e * 5 + 1], a1 = act[e * 5 + 2];
    uint4 r0 = rows[(size_t)e * 12 + (a0 % 12)];
    uint4 r1 = rows[(size_t)e * 12 + (a1 % 12)];
    r0.x += h[0]; r1.y += r0.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) hdr[(size_t)k * E + e] = h[k] + r1.y;
    rows[(size_t)e * 12 + (a0 % 12)] = r0;
    rows[(size_t)e * 12 + (a1 % 12)] = r1;
    out[e] = make_uint2(r0.x, r1.y);
}

// variant: rows as 12 env-fastest 8-byte columns, all loaded before the action is known (no dependent round trip),
// row picked by register select, changed rows stored back to their columns
__global__ __launch_bounds__(256) void traffic_cols_kernel(uint32_t* __restrict__ hdr, uint2* __restrict__ cols,
                                                           const int* __restrict__ act, uint2* __restrict__ out, int E) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    uint32_t h[8];
    uint2 r[12];
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = hdr[(size_t)k * E + e];
#pragma unroll
    for (int k = 0; k < 12; ++k) r[k] = cols[(size_t)k * E + e];
    int a0 = act[e * 5 + 1] % 12, a1 = act[e * 5 + 2] % 12;
    uint2 r0 = r[0], r1 = r[0];
#pragma unroll
    for (int k = 1; k < 12; ++k) { if (a0 == k) r0 = r[k]; if (a1 == k) r1 = r[k]; }
    r0.x += h[0]; r1.y += r0.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) hdr[(size_t)k * E + e] = h[k] + r1.y;
    cols[(size_t)a0 * E + e] = r0;
    cols[(size_t)a1 * E + e] = r1;
    out[e] = make_uint2(r0.x, r1.y);
}

// packed-batch shapes: header 16 + 16 + sets 16 (columns) + action 20, and the 96-byte body either as an array of structures
// (7 x 16-byte loads per lane at stride 96: what the step kernel does) or as seven env-fastest 16-byte columns
template <bool SOA>
__global__ __launch_bounds__(64) void packed_kernel(uint4* __restrict__ h, uint4* __restrict__ body, const int* __restrict__ act,
                                                    uint2* __restrict__ out, int E) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    uint4 h0 = h[e], h1 = h[(size_t)E + e], pk = h[(size_t)2 * E + e];
    uint4 b[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = SOA ? body[(size_t)k * E + e] : body[(size_t)e * 6 + k];
    int a0 = act[e * 5 + 1];
    uint32_t acc = h0.x + h1.y + pk.z + a0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc += b[k].x ^ b[k].w;
    h0.x = acc; h1.y = acc; pk.z = acc;
    h[e] = h0; h[(size_t)E + e] = h1; h[(size_t)2 * E + e] = pk;
    uint32_t* rows = reinterpret_cast<uint32_t*>(SOA ? body + (size_t)(3 + ((a0 >> 2) % 3)) * E + e : body + (size_t)e * 6 + 3 + ((a0 >> 2) % 3));
    rows[a0 & 3] = acc;
    out[e] = make_uint2(acc, acc);
}

template <class F> static float replay(hipStream_t st, int K, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < K; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1000.f / K;
}

int main(int argc, char** argv) {
    int E = argc > 1 ? atoi(argv[1]) : 65536;
    const int K = 1000;
    hipStream_t st; CK(hipStreamCreate(&st));
    uint32_t* hdr; uint4* rows; int* act; uint2* out;
    CK(hipMalloc(&hdr, (size_t)E * 32)); CK(hipMalloc(&rows, (size_t)E * 12 * 16));
    CK(hipMalloc(&act, (size_t)E * 20)); CK(hipMalloc(&out, (size_t)E * 8));
    CK(hipMemset(hdr, 0, (size_t)E * 32)); CK(hipMemset(rows, 0, (size_t)E * 12 * 16));
    std::vector<int> ha((size_t)E * 5);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = rand() % 12;
    CK(hipMemcpy(act, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    dim3 grid((E + 255) / 256), block(256);
    float t0 = replay(st, K, [&] { hipLaunchKernelGGL(empty_kernel, grid, block, 0, st, (int*)nullptr); });
    float t1 = replay(st, K, [&] { hipLaunchKernelGGL(traffic_kernel, grid, block, 0, st, hdr, rows, act, out, E); });
    dim3 grid64((E + 63) / 64), block64(64);
    float t2 = replay(st, K, [&] { hipLaunchKernelGGL(traffic_kernel, grid64, block64, 0, st, hdr, rows, act, out, E); });
    float t3 = replay(st, K, [&] { hipLaunchKernelGGL(traffic_cols_kernel, grid, block, 0, st, hdr, (uint2*)rows, act, out, E); });
    uint4* hp; uint4* bp;
    CK(hipMalloc(&hp, (size_t)E * 48)); CK(hipMalloc(&bp, (size_t)E * 96));
    CK(hipMemset(hp, 0, (size_t)E * 48)); CK(hipMemset(bp, 0, (size_t)E * 96));
    float t4 = replay(st, K, [&] { hipLaunchKernelGGL((packed_kernel<false>), grid64, block64, 0, st, hp, bp, act, out, E); });
    float t5 = replay(st, K, [&] { hipLaunchKernelGGL((packed_kernel<true>), grid64, block64, 0, st, hp, bp, act, out, E); });
    printf("{\"packed_aos_us\": %.3f, \"packed_soa_us\": %.3f}\n", t4, t5);
    printf("{\"envs\": %d, \"empty_us\": %.3f, \"traffic_only_us\": %.3f, \"traffic_only_wg64_us\": %.3f, \"bytes_per_env\": %d, "
           "\"row_columns_us\": %.3f}\n", E, t0, t1, t2, 20 + 64 + 64 + 8, t3);
    return 0;
}
