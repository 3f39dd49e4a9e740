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
// One workgroup per (env, slice of 256 * UNROLL groups); a thread owns UNROLL groups of 16 bytes of logits (4 fp32 / 8 bf16 actions):
// all its loads are issued before its first store, and the mask bits of a group come from ONE division chain (the connect block is
// periodic).  Bound: HBM read + write of the logits.
#pragma once
#include "mcbs_device.h"
#include "mcbs_obs.hip"

namespace mcbs {

struct LogitsGeom {        // Discrete layout of the batch, set up on the host
    uint32_t A, M, ML, RL, C, N, L, R;     // total actions, connect block, local block, connect row length P*C, credentials, nodes, local / remote ids
    FastDiv dRL, dC, dN, dL, dR;
};

// LT: float, or uint16_t for 16-bit logits (bf16 patterns are only moved or replaced).  GW: actions per group = per vector access
// (16 bytes: 4 fp32 / 8 bf16; 8-byte groups of 4 bf16 when the rows are only 8-byte aligned, e.g. Chain-10's 14 172 actions).
template <typename LT, uint32_t GW, int UNROLL, bool VEC>
__global__ __launch_bounds__(256) void mask_logits_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, const ObsDigest* __restrict__ digest,
                                                          LT* __restrict__ logits, size_t row_stride, LT fill, LogitsGeom G) {
    constexpr uint32_t NWORD = GW * (uint32_t)sizeof(LT) / 4u;      // dwords per group: 4 or 2
    static_assert(NWORD == 4u || NWORD == 2u, "group = 16 or 8 bytes");
    const uint32_t e = blockIdx.y;
    const ObsDigest d = digest[e];                       // uniform per workgroup: scalar loads
    const uint32_t n_disc = d.blank ? 0u : d.n_disc, n_creds = d.n_creds;
    LT* row = logits + (size_t)e * row_stride;
    auto own = [&](uint32_t s) -> bool { return s < G.N && ((d.own_ext[(s >> 6) & 3u] >> (s & 63u)) & 1ull); };
    auto pair_on = [&](uint32_t q) -> bool {             // row q = (source s, target t): s owned (hence discovered), t discovered
        const uint32_t s = fdiv(q, G.dN), t = q - s * G.N;
        return own(s) && t < n_disc;
    };
    auto mask_at = [&](uint32_t a) -> bool {             // one action, any region
        if (a < G.M) {                                   // connect[s][t][p][c] = on(s, t) && c < n_creds      (env.py:664-677)
            const uint32_t q = fdiv(a, G.dRL), r = a - q * G.RL, c = r - fdiv(r, G.dC) * G.C;
            return c < n_creds && pair_on(q);
        }
        if (a < G.M + G.ML) {                            // local[i][l] = owned(i) && vulnerability l applies to node i   (env.py:653-663)
            const uint32_t b = a - G.M, i = fdiv(b, G.dL), l = b - i * G.L;
            if (!(own(i) && i < n_disc)) return false;
            const uint8_t* body = S.body + (size_t)e * S.body_stride;
            const mcbs_node_static* NS = reinterpret_cast<const mcbs_node_static*>(T.base + Cp->off_node);
            return (NS[body[S.off_disc + i]].local_mask >> l) & 1u;
        }
        return a < G.A && pair_on(fdiv(a - G.M - G.ML, G.dR));   // remote[s][t][r] = on(s, t)
    };
    uint64_t pp = 0;                                     // credential pattern of one period, repeated to at least C + GW bits (uniform per workgroup)
    if (G.C + GW <= 64u) {
        const uint64_t one = n_creds >= 64u ? ~0ull : ((1ull << n_creds) - 1ull);
        for (uint32_t sh = 0; sh < 64u; sh += G.C) pp |= one << sh;
    }
    const uint32_t base = (blockIdx.x * 256u * UNROLL + threadIdx.x) * GW;
    uint32_t v[UNROLL][4];
    bool have[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {                   // all loads first
        const uint32_t a0 = base + (uint32_t)u * 256u * GW;
        have[u] = VEC && a0 + GW <= G.A;                 // VEC: rows start aligned to the group size
        v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0u;
        if (have[u]) {
            if constexpr (NWORD == 4u) { const uint4 t4 = *reinterpret_cast<const uint4*>(row + a0); v[u][0] = t4.x; v[u][1] = t4.y; v[u][2] = t4.z; v[u][3] = t4.w; }
            else { const uint2 t2 = *reinterpret_cast<const uint2*>(row + a0); v[u][0] = t2.x; v[u][1] = t2.y; }
        }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const uint32_t a0 = base + (uint32_t)u * 256u * GW;
        if (a0 >= G.A) continue;
        uint32_t m = 0;                                  // bit j: action a0 + j is allowed
        if (a0 + GW <= G.M && G.C + GW <= 64u && G.RL >= GW) {
            // the whole group lies in the connect block and one credential period plus a group fits 64 bits (Chain-10: C = 12, ToyCtf: 10):
            // `pp` = the periodic pattern "n_creds ones, C - n_creds zeros" as a bit string, so the group's credential bits are one shift;
            // the group touches at most two (source, target) rows, each on or off as a whole
            const uint32_t q0 = fdiv(a0, G.dRL), r0 = a0 - q0 * G.RL, c0 = r0 - fdiv(r0, G.dC) * G.C;
            const uint32_t first = G.RL - r0 < GW ? G.RL - r0 : GW;                  // actions of the group that belong to row q0
            const uint32_t lo = (1u << first) - 1u, all = (1u << GW) - 1u;
            const uint32_t rows = (pair_on(q0) ? lo : 0u) | ((first < GW && pair_on(q0 + 1u)) ? (all & ~lo) : 0u);
            m = (uint32_t)(pp >> c0) & rows;
        } else if (a0 + GW <= G.M && G.C >= 4u && G.RL >= GW) {
            // the whole group lies in the connect block (all but the last ~2 % of a row): one division chain per GROUP — the group
            // touches at most two (source, target) rows, and the credential index just counts on modulo C (RL is a multiple of C)
            const uint32_t q0 = fdiv(a0, G.dRL), r0 = a0 - q0 * G.RL, c0 = r0 - fdiv(r0, G.dC) * G.C;
            const bool on0 = pair_on(q0), on1 = pair_on(q0 + 1u);
#pragma unroll
            for (uint32_t j = 0; j < GW; ++j) {
                uint32_t c = c0 + j;
                c -= c >= G.C ? G.C : 0u;
                c -= c >= G.C ? G.C : 0u;
                const bool on = (r0 + j >= G.RL) ? on1 : on0;
                m |= (uint32_t)(on && c < n_creds) << j;
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < GW; ++j) m |= (uint32_t)mask_at(a0 + j) << j;
        }
        if (have[u]) {
            uint32_t o[4] = {v[u][0], v[u][1], v[u][2], v[u][3]};
            if constexpr (sizeof(LT) == 4) {
                const uint32_t f = __float_as_uint(fill);
#pragma unroll
                for (uint32_t w = 0; w < NWORD; ++w) o[w] = ((m >> w) & 1u) ? o[w] : f;
            } else {
                const uint32_t f = (uint32_t)fill, ff = f | (f << 16);
#pragma unroll
                for (uint32_t w = 0; w < NWORD; ++w) {    // two 16-bit patterns per dword
                    const uint32_t b2 = m >> (2u * w), keep = ((b2 & 1u) ? 0x0000FFFFu : 0u) | ((b2 & 2u) ? 0xFFFF0000u : 0u);
                    o[w] = (o[w] & keep) | (ff & ~keep);
                }
            }
            if constexpr (NWORD == 4u) *reinterpret_cast<uint4*>(row + a0) = make_uint4(o[0], o[1], o[2], o[3]);
            else *reinterpret_cast<uint2*>(row + a0) = make_uint2(o[0], o[1]);
        } else {                                         // unaligned rows, and the last (partial) group of a row
            for (uint32_t j = 0; j < GW && a0 + j < G.A; ++j)
                if (!((m >> j) & 1u)) row[a0 + j] = fill;
        }
    }
}

} // namespace mcbs
