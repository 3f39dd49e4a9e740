// mcbs_logits.hip — on-device "action mask -> logits" for the Discrete attacker action space.
//
// MaskedDiscreteAttackerWrapper.action_masks() (marlon/baseline_models/env_wrappers/action_masking.py:90-110) hands MaskablePPO
// one bool per Discrete action — connect block ((src*N+tgt)*P+port)*C+cred, then local src*L+vuln, then remote
// (src*N+tgt)*R+vuln — and the policy turns it into `where(mask, logits, -1e8)` (sb3_contrib MaskableCategorical).  Materialised,
// that mask is N*N*P*C + N*L + N*N*R bytes per env and step (Chain-10 at 12/12: 14 172 B, 929 MB for 65 536 envs) written by the
// observation kernel only to be read back once.  This kernel applies it to the logits in place, straight from the 64-byte
// per-env digest the last observation left (owned-source bits by external index, discovered-node and cached-credential
// counts, blank flag: exactly what the mask bytes of THAT observation were computed from), so the mask never exists in memory:
//     logits[e, a] = mask(e, a) ? logits[e, a] : fill
// One workgroup per (env, slice of 256 * 4 * UNROLL actions); a thread owns UNROLL groups of 4 consecutive actions (16 bytes of
// fp32 / 8 bytes of bf16 logits): all its loads are issued before its first store.  Bound: HBM read + write of the logits.
#pragma once
#include "mcbs_device.h"
#include "mcbs_obs.hip"

namespace mcbs {

struct LogitsGeom {        // Discrete layout of the batch, set up on the host
    uint32_t A, M, ML, RL, C, N, L, R;     // total actions, connect block, local block, connect row length P*C, credentials, nodes, local / remote ids
    FastDiv dRL, dC, dN, dL, dR;
};

template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<uint16_t> { using type = ushort4; };     // bf16 / fp16 logits: 16-bit patterns, only moved or replaced

template <typename LT, int UNROLL, bool VEC>
__global__ __launch_bounds__(256) void mask_logits_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, const ObsDigest* __restrict__ digest,
                                                          LT* __restrict__ logits, size_t row_stride, LT fill, LogitsGeom G) {
    using V = typename Vec4<LT>::type;
    const uint32_t e = blockIdx.y;
    const ObsDigest d = digest[e];                       // uniform per workgroup: scalar loads
    const uint32_t n_disc = d.blank ? 0u : d.n_disc, n_creds = d.n_creds;
    LT* row = logits + (size_t)e * row_stride;
    auto own = [&](uint32_t s) -> bool { return s < G.N && ((d.own_ext[(s >> 6) & 3u] >> (s & 63u)) & 1ull); };
    auto pair_on = [&](uint32_t q) -> bool {             // row q = (source s, target t): s owned (hence discovered), t discovered
        const uint32_t s = fdiv(q, G.dN), t = q - s * G.N;
        return own(s) && t < n_disc;
    };
    const uint32_t base = (blockIdx.x * 256u * UNROLL + threadIdx.x) * 4u;
    V v[UNROLL];
    bool have[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {                   // all loads first
        const uint32_t a0 = base + (uint32_t)u * 1024u;
        have[u] = VEC && a0 + 4u <= G.A;                 // VEC: rows start 16-byte (8-byte for 16-bit logits) aligned
        if (have[u]) v[u] = *reinterpret_cast<const V*>(row + a0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const uint32_t a0 = base + (uint32_t)u * 1024u;
        if (a0 >= G.A) continue;
        bool m[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t a = a0 + j;
            bool on = false;
            if (a < G.M) {                               // connect[s][t][p][c] = on(s, t) && c < n_creds      (env.py:664-677)
                const uint32_t q = fdiv(a, G.dRL), r = a - q * G.RL, c = r - fdiv(r, G.dC) * G.C;
                on = c < n_creds && pair_on(q);
            } else if (a < G.M + G.ML) {                 // local[i][l] = owned(i) && vulnerability l applies to node i   (env.py:653-663)
                const uint32_t b = a - G.M, i = fdiv(b, G.dL), l = b - i * G.L;
                if (own(i) && i < n_disc) {
                    const uint8_t* body = S.body + (size_t)e * S.body_stride;
                    const mcbs_node_static* NS = reinterpret_cast<const mcbs_node_static*>(T.base + Cp->off_node);
                    on = (NS[body[S.off_disc + i]].local_mask >> l) & 1u;
                }
            } else if (a < G.A) {                        // remote[s][t][r] = on(s, t)
                on = pair_on(fdiv(a - G.M - G.ML, G.dR));
            }
            m[j] = on;
        }
        if (have[u]) {
            V o = v[u];
            o.x = m[0] ? o.x : fill; o.y = m[1] ? o.y : fill; o.z = m[2] ? o.z : fill; o.w = m[3] ? o.w : fill;
            *reinterpret_cast<V*>(row + a0) = o;
        } else {                                         // unaligned rows, and the last (partial) group of a row
            for (uint32_t j = 0; j < 4u && a0 + j < G.A; ++j)
                if (!m[j]) row[a0 + j] = fill;
        }
    }
}

} // namespace mcbs
