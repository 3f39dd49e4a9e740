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
// The logits are never READ: masked-out actions are overwritten, allowed ones left alone.  Bound: HBM write of the masked-out logits
// (nearly all of them).
#pragma once
#include "mcbs_device.h"
#include "mcbs_obs.hip"

namespace mcbs {

struct LogitsGeom {        // Discrete layout of the batch, set up on the host
    uint32_t A, M, ML, RL, C, N, L, R;     // total actions, connect block, local block, connect row length P*C, credentials, nodes, local / remote ids
    FastDiv dRL, dC, dN, dL, dR;
};

// LT: float, or uint16_t for 16-bit logits (bf16 patterns are only replaced).  GW: actions per group = per vector store
// (16 bytes: 4 fp32 / 8 bf16; 8-byte groups of 4 bf16 when the rows are only 8-byte aligned, e.g. Chain-10's 14 172 actions).
//
// One WAVEFRONT per env (four independent ones per workgroup: no LDS, no barrier).  A span = the 64 groups one store instruction of
// the wavefront covers = 64 * GW consecutive actions, starting on a 128-byte line of memory.  Nearly every span holds no allowed action at all (a few dozen of Chain-10's
// 14 172 actions are allowed): for each chunk of 64 spans, lane k first decides whether span k is LIVE — some (source, target) row
// overlapping it is on, or it touches the local block — and one ballot turns that into a scalar mask; a span that is not live costs
// one scalar bit test and one store.  (Measured on the way here, 65 536 Chain-10 envs, fp32 / bf16 logits, us per launch: read-modify-write with one
// division chain per group 1 390 / -; write-only 1 008 / 882; row bits in LDS + span bits per four-wavefront workgroup 785 / 595 — the
// same time for half the bytes: bound by instruction issue and the two barriers; a persistent grid of such workgroups 950; this
// kernel 745 / 414.  Reference points on the same buffer: a plain fill 536 / 270, this store pattern with no mask work at all 700 / 367,
// torch.where with a materialised mask 1 440 / 830.  Non-temporal stores: no change.  tools/bench_logits.py.)
template <typename LT, uint32_t GW, bool VEC>
__global__ __launch_bounds__(256) void mask_logits_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, const ObsDigest* __restrict__ digest,
                                                          LT* __restrict__ logits, size_t row_stride, LT fill, LogitsGeom G) {
    constexpr uint32_t NWORD = GW * (uint32_t)sizeof(LT) / 4u;      // dwords per group: 4 or 2
    static_assert(NWORD == 4u || NWORD == 2u, "group = 16 or 8 bytes");
    constexpr uint32_t ALL = (1u << GW) - 1u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t e = blockIdx.y * 4u + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform
    if (e >= S.E) return;
    const ObsDigest d = digest[e];                       // uniform per wavefront: scalar loads
    const uint32_t n_disc = d.blank ? 0u : d.n_disc, n_creds = d.n_creds;
    const uint32_t remote0 = G.M + G.ML;
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
        if (a < remote0) {                               // local[i][l] = owned(i) && vulnerability l applies to node i   (env.py:653-663)
            const uint32_t b = a - G.M, i = fdiv(b, G.dL), l = b - i * G.L;
            if (!(own(i) && i < n_disc)) return false;
            const uint8_t* body = S.body + (size_t)e * S.body_stride;
            const mcbs_node_static* NS = reinterpret_cast<const mcbs_node_static*>(T.base + Cp->off_node);
            return (NS[body[S.off_disc + i]].local_mask >> l) & 1u;
        }
        return a < G.A && pair_on(fdiv(a - remote0, G.dR));   // remote[s][t][r] = on(s, t)
    };
    // bit j: the (source, target) row that action rel + j of a block of `rowlen`-long rows belongs to is on (rowlen >= GW: two rows at most)
    auto rows_mask = [&](uint32_t rel, const FastDiv& dRow, uint32_t rowlen, uint32_t& r0) -> uint32_t {
        const uint32_t q0 = fdiv(rel, dRow);
        r0 = rel - q0 * rowlen;
        const uint32_t first = rowlen - r0 < GW ? rowlen - r0 : GW, lo = (1u << first) - 1u;
        return (pair_on(q0) ? lo : 0u) | ((first < GW && pair_on(q0 + 1u)) ? (ALL & ~lo) : 0u);
    };
    uint64_t pp = 0;                                     // credential pattern of one period, repeated to at least C + GW bits (uniform)
    if (G.C + GW <= 64u) {
        const uint64_t one = n_creds >= 64u ? ~0ull : ((1ull << n_creds) - 1ull);
        for (uint32_t sh = 0; sh < 64u; sh += G.C) pp |= one << sh;
    }
    // Spans start on 128-byte lines of MEMORY, not of the row (rows are only 16- or 8-byte aligned: Chain-10's fp32 row is 56 688 B):
    // the env's groups are shifted down by `sh`, so that every store instruction writes whole lines and no line is shared by two
    // instructions; the first span is short.
    constexpr uint32_t GB = GW * (uint32_t)sizeof(LT);
    const uint32_t sh = VEC ? (uint32_t)((reinterpret_cast<uintptr_t>(row) / GB) % (128u / GB)) : 0u;
    const uint32_t nspan = ((G.A + GW - 1u) / GW + sh + 63u) / 64u;
    for (uint32_t c0s = blockIdx.x * 64u; c0s < nspan; c0s += gridDim.x * 64u) {      // chunks of 64 spans
        bool live = false;                               // lane k: span c0s + k holds an allowed action (or might)
        {
            const uint32_t g0 = (c0s + lane) * 64u;      // groups [g0 - sh, g0 + 64 - sh) of the row
            const uint32_t s0 = (g0 > sh ? g0 - sh : 0u) * GW, s1 = (g0 + 64u - sh) * GW < G.A ? (g0 + 64u - sh) * GW : G.A;      // [s0, s1)
            auto any_row = [&](uint32_t lo, uint32_t hi, const FastDiv& dRow) {                    // rows of actions lo .. hi (relative to their block)
                const uint32_t qa = fdiv(lo, dRow), qb = fdiv(hi, dRow);
                if (qb - qa > 8u) { live = true; return; }
                for (uint32_t q = qa; q <= qb; ++q) live |= pair_on(q);
            };
            if (s0 < s1) {
                if (s0 < G.M) any_row(s0, (s1 < G.M ? s1 : G.M) - 1u, G.dRL);
                if (s1 > G.M && s0 < remote0) live = true;                                         // local block: per-node vulnerability masks
                if (s1 > remote0) any_row((s0 > remote0 ? s0 : remote0) - remote0, s1 - 1u - remote0, G.dR);
            }
        }
        const uint64_t live_mask = __ballot(live);
        const uint32_t ns = nspan - c0s < 64u ? nspan - c0s : 64u;
#pragma unroll 4
        for (uint32_t j = 0; j < ns; ++j) {
            const uint32_t g = (c0s + j) * 64u + lane;
            const uint32_t a0 = (g - sh) * GW;
            if (g < sh || a0 >= G.A) continue;           // the first span's head, the last span's tail
            uint32_t m = 0, r0;                          // bit j: action a0 + j is allowed
            if (!((live_mask >> j) & 1ull)) {            // scalar test
            } else if (a0 + GW <= G.M && G.RL >= GW) {
                // the whole group lies in the connect block: at most two (source, target) rows, each on or off as a whole; within an
                // on row the credential index just counts on modulo C (RL is a multiple of C)
                const uint32_t rows = rows_mask(a0, G.dRL, G.RL, r0);
                if (rows) {
                    const uint32_t c0 = r0 - fdiv(r0, G.dC) * G.C;
                    if (G.C + GW <= 64u) {
                        // one credential period plus a group fits 64 bits (Chain-10: C = 12, ToyCtf: 10): `pp` = the periodic pattern
                        // "n_creds ones, C - n_creds zeros" as a bit string, so the group's credential bits are one shift
                        m = (uint32_t)(pp >> c0) & rows;
                    } else {
                        uint32_t c = c0;
#pragma unroll
                        for (uint32_t i = 0; i < GW; ++i) {
                            m |= (uint32_t)(c < n_creds) << i;
                            c = c + 1u == G.C ? 0u : c + 1u;
                        }
                        m &= rows;
                    }
                }
            } else if (a0 >= remote0 && a0 + GW <= G.A && G.R >= GW) {
                m = rows_mask(a0 - remote0, G.dR, G.R, r0);     // the whole group lies in the remote block: remote[s][t][r] = on(s, t)
            } else {
#pragma unroll
                for (uint32_t i = 0; i < GW; ++i) m |= (uint32_t)mask_at(a0 + i) << i;
            }
            // write-only: a group whose actions are all masked out is ONE vector store of `fill`, a group that is allowed as a whole is
            // left alone, a mixed group stores `fill` element by element — the logits are never read
            const uint32_t in_row = a0 + GW <= G.A ? ALL : (1u << (G.A - a0)) - 1u;
            const uint32_t off = ~m & in_row;            // bit i: action a0 + i is replaced
            if (VEC && off == ALL) {
                if constexpr (sizeof(LT) == 4) {
                    const uint32_t f = __float_as_uint(fill);
                    *reinterpret_cast<uint4*>(row + a0) = make_uint4(f, f, f, f);
                } else {
                    const uint32_t f = (uint32_t)fill, ff = f | (f << 16);
                    if constexpr (NWORD == 4u) *reinterpret_cast<uint4*>(row + a0) = make_uint4(ff, ff, ff, ff);
                    else *reinterpret_cast<uint2*>(row + a0) = make_uint2(ff, ff);
                }
            } else if (off) {
#pragma unroll
                for (uint32_t i = 0; i < GW; ++i)
                    if ((off >> i) & 1u) row[a0 + i] = fill;
            }
        }
    }
}

} // namespace mcbs
