// Compile-only reproducer for the round-1 abort in mcbs_decode_attacker_actions (gpurun_out/gputest4.log, round 1):
// the decode kernel as it was first written, with a three-way if / else if / else over a LOADED value (the action kind).
// The faulting source was never committed; this is its reconstruction from the comment in marlon_amd/csrc/mcbs_aux.hip.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -save-temps -c tools/decode_repro/decode_branchy.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

struct State { const uint4* h0; uint32_t E; };
struct Cfg { uint32_t L, R, P; };

extern "C" __global__ __launch_bounds__(256) void decode_branchy(State S, const Cfg* __restrict__ Cp, uint32_t Nmax, uint32_t Cmax, const int64_t* md,
                                                                  const int64_t* discrete, int32_t* out, uint8_t* invalid) {
    const Cfg& C = *Cp;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S.E) return;
    const int64_t nd = (int64_t)(S.h0[e].z & 0xFFFFu);
    int64_t kind, a = 0, b = 0, c = 0, d = 0;
    if (md) {
        const int64_t* v = md + (size_t)e * 10;
        kind = v[0];
        if (kind == 0) { a = v[1]; b = v[2]; }
        else if (kind == 1) { a = v[3]; b = v[4]; c = v[5]; }
        else { a = v[6]; b = v[7]; c = v[8]; d = v[9]; }
    } else {
        const int64_t N = Nmax, P = C.P, Cm = Cmax, L = C.L, R = C.R;
        const int64_t connect_size = N * N * P * Cm, local_size = N * L;
        int64_t idx = discrete[e];
        if (idx < connect_size) {
            kind = 2;
            d = idx % Cm; idx /= Cm;
            c = idx % P; idx /= P;
            b = idx % N; a = idx / N;
        } else if (idx < connect_size + local_size) {
            idx -= connect_size;
            kind = 0;
            b = idx % L; a = idx / L;
        } else {
            idx -= connect_size + local_size;
            kind = 1;
            c = idx % R; idx /= R;
            b = idx % N; a = idx / N;
        }
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
