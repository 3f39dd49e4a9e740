// The decode kernel as shipped (marlon_amd/csrc/mcbs_aux.hip decode_kernel): every component is loaded unconditionally and the row is
// picked with selects — no divergent branch whose arms load into the same variables.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -save-temps -c tools/decode_repro/decode_select.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

struct State { const uint4* h0; uint32_t E; };
struct Cfg { uint32_t L, R, P; };

extern "C" __global__ __launch_bounds__(256) void decode_select(State S, const Cfg* __restrict__ Cp, uint32_t Nmax, uint32_t Cmax, const int64_t* md,
                                                                 const int64_t* discrete, int32_t* out, uint8_t* invalid) {
    const Cfg& C = *Cp;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S.E) return;
    const int64_t nd = (int64_t)(S.h0[e].z & 0xFFFFu);
    int64_t kind, a = 0, b = 0, c = 0, d = 0;
    if (md) {
        const int64_t* v = md + (size_t)e * 10;
        int64_t x[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) x[i] = v[i];
        kind = x[0];
        const bool k0 = kind == 0, k1 = kind == 1;
        a = k0 ? x[1] : (k1 ? x[3] : x[6]);
        b = k0 ? x[2] : (k1 ? x[4] : x[7]);
        c = k0 ? 0 : (k1 ? x[5] : x[8]);
        d = (k0 || k1) ? 0 : x[9];
    } else {
        const int64_t N = Nmax, P = C.P, Cm = Cmax, L = C.L, R = C.R;
        const int64_t connect_size = N * N * P * Cm, local_size = N * L;
        const int64_t idx = discrete[e];
        const bool is_c = idx < connect_size, is_l = !is_c && idx < connect_size + local_size;
        const int64_t rel = is_c ? idx : (is_l ? idx - connect_size : idx - connect_size - local_size);
        const int64_t inner = is_c ? Cm : (is_l ? L : R);
        const int64_t x0 = rel % inner, q = rel / inner;
        const int64_t qp = q / P;
        kind = is_c ? 2 : (is_l ? 0 : 1);
        a = is_c ? qp / N : (is_l ? q : q / N);
        b = is_c ? qp % N : (is_l ? x0 : q % N);
        c = is_c ? q % P : (is_l ? 0 : x0);
        d = is_c ? x0 : 0;
    }
    bool ok;
    if (kind == 0) ok = a < nd;
    else if (kind == 1 || kind == 2) ok = a < nd && b < nd;
    else ok = false;
    int32_t* o = out + (size_t)e * 5;
    o[0] = ok ? (int32_t)kind : 3;
    o[1] = (int32_t)a; o[2] = (int32_t)b; o[3] = (int32_t)c; o[4] = (int32_t)d;
    invalid[e] = ok ? 0 : 1;
}
