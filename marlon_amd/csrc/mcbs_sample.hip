// mcbs_sample.hip — the random agents' action sampler, shared by sample_kernel (one batch of actions) and the step kernel's
// on-device rollout (mcbs_rollout_random).
//
// valid == 0: every component uniform in its bound (cyberbattle_env.py action space, :540-559), invalid and out-of-bound actions
// included.  valid != 0: the distribution of sample_valid_action (:959-1047): kind uniform among {local, remote, connect} (connect
// only once a credential is cached), source uniform among the nodes with privilege >= LocalUser, target uniform among the
// discovered nodes, vulnerability / port / credential uniform, whole action re-drawn until the action mask allows it.  (The
// reference draws from PCG64 streams; only the distribution is reproduced here — this is harness, not part of the step's
// parity.)  Philox keyed by (seed, global env id), counter (env id, step, draw block): the same (seed, step) gives the same action
// whether it is sampled by the stand-alone kernel or inside a rollout.
#pragma once
#include "mcbs_device.h"

namespace mcbs {

__device__ inline void sample_action(const DevState& S, const Topo& T, const StepCfg& C, uint32_t e, int valid, uint64_t seed, uint64_t step,
                                     uint32_t Nmax, uint32_t Cmax, int32_t (&a)[5]) {
    const uint64_t gid = C.env_id_base + e;
    uint32_t ctr = 0, buf[4], have = 0;
    auto next_u32 = [&]() -> uint32_t {
        if (!have) {
            philox4x32_10((uint32_t)gid, (uint32_t)step, (uint32_t)(step >> 32), ctr++, (uint32_t)seed ^ 0x5A17ACEDu,
                          (uint32_t)(seed >> 32) ^ (uint32_t)(gid >> 32), buf);
            have = 4;
        }
        return buf[--have];
    };
    auto below = [&](uint32_t n) -> uint32_t { return n ? (uint32_t)(((uint64_t)next_u32() * n) >> 32) : 0u; };
    a[0] = a[1] = a[2] = a[3] = a[4] = 0;
    if (!valid) {
        a[0] = (int32_t)below(3);
        if (a[0] == 0) { a[1] = below(Nmax); a[2] = below(C.L); }
        else if (a[0] == 1) { a[1] = below(Nmax); a[2] = below(Nmax); a[3] = below(C.R); }
        else { a[1] = below(Nmax); a[2] = below(Nmax); a[3] = below(C.P); a[4] = below(Cmax); }
        return;
    }
    const uint4 h0 = S.h0[e];
    const uint32_t n_disc = h0.z & 0xFFFFu, n_creds = h0.z >> 16, owned = h0.w & 0xFFFFu;
    const uint8_t* body = S.body + (size_t)e * S.body_stride;
    const uint8_t* dl = body + S.off_disc;
    const mcbs_node_static* NS = reinterpret_cast<const mcbs_node_static*>(T.base + C.off_node);
    // one pass over the discovery order up front (independent loads), instead of set look-ups in memory inside every attempt:
    // bit i of own_ext / inst_ext = the node at external index i has privilege >= LocalUser / has the agent installed
    uint64_t plo[4] = {0, 0, 0, 0}, phi[4] = {0, 0, 0, 0}, inst[4] = {0, 0, 0, 0};
    for (uint32_t w = 0; w < S.NW && w < 4u; ++w) { plo[w] = S.get(M_PLO, w, e); phi[w] = S.get(M_PHI, w, e); inst[w] = S.get(M_INST, w, e); }
    uint64_t own_ext[4] = {0, 0, 0, 0}, inst_ext[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < n_disc; ++i) {
        const uint32_t n = dl[i], w = (n >> 6) & 3u, b = n & 63u;
        own_ext[(i >> 6) & 3u] |= (((plo[w] | phi[w]) >> b) & 1ull) << (i & 63u);
        inst_ext[(i >> 6) & 3u] |= ((inst[w] >> b) & 1ull) << (i & 63u);
    }
    auto owned_source = [&]() -> uint32_t {      // external index of the k-th owned node, k uniform
        uint32_t k = below(owned);
        for (uint32_t c = 0; c < 4u; ++c) {
            uint64_t m = own_ext[c];
            const uint32_t pc = (uint32_t)__popcll(m);
            if (k >= pc) { k -= pc; continue; }
            while (k) { m &= m - 1ull; k -= 1u; }
            return c * 64u + (uint32_t)__builtin_ctzll(m);
        }
        return 0u;
    };
    for (int attempt = 0; attempt < 64; ++attempt) {
        const uint32_t kind = below(n_creds ? 3u : 2u);
        const uint32_t src = owned_source();
        const bool installed = (inst_ext[(src >> 6) & 3u] >> (src & 63u)) & 1ull;
        if (kind == 0) {
            const uint32_t v = below(C.L);
            a[0] = 0; a[1] = (int32_t)src; a[2] = (int32_t)v; a[3] = 0; a[4] = 0;
            if (installed && ((local_mask_of(C, NS, body, dl[src]) >> v) & 1u)) break;
        } else if (kind == 1) {
            a[0] = 1; a[1] = (int32_t)src; a[2] = (int32_t)below(n_disc); a[3] = (int32_t)below(C.R); a[4] = 0;
            if (installed) break;
        } else {
            a[0] = 2; a[1] = (int32_t)src; a[2] = (int32_t)below(n_disc); a[3] = (int32_t)below(C.P); a[4] = (int32_t)below(n_creds);
            if (installed) break;
        }
    }
}

} // namespace mcbs
