// mcbs_defend.hip — the learned defender's turn, batches created with MCBS_DEFENDER_EXTERNAL.
//
//   defender_kernel     : DefenderEnvWrapper.is_defender_action_valid (marlon/baseline_models/env_wrappers/defend_wrapper.py:329-412)
//                         then LearningDefender.executeAction (marlon/defender_agents/defender.py:31-107):
//                         DefenderAgentActions.on_attacker_step_taken (actions.py:714-746) and, if valid, the action.
//                         One lane per env; shares the re-imaging ring, the availability code and reimage() with the step kernel.
//   defender_obs_kernel : DefenderEnvWrapper.observe (defend_wrapper.py:492-534), one thread per (env, node); for topologies of up to
//                         32 nodes the turn kernels write it themselves (defender_observe_env), one launch per turn.
// Firewall rule LISTS are state objects of their own (several nodes / directions may hold the same Python list, and
// copy.deepcopy keeps that aliasing): per (env, list) 12 bits — for each of the six names the learned defender can name
// (RDP, SSH, HTTPS, HTTP, su, sudo) "a rule with that name exists" and "the first such rule is ALLOW".  That is everything
// block_traffic (remove every rule with the name) and allow_traffic (append an ALLOW rule when the examined list has
// none — to the node's incoming list in both cases, defender.py:68) can observe or change.
#pragma once
#include "mcbs_device.h"
#include "mcbs_step.hip"

namespace mcbs {

// One env's defender turn (env e < S.E).  -> ok (the action was valid), avail (availability after the tick), evicted
template <int WT>
__device__ __forceinline__ void defender_turn(const DevState& S, const Topo& T, const StepCfg& C, const int64_t* actions, uint32_t e,
                                              bool& ok_out, double& avail_out, bool& evicted_out) {
    const uint4 h0 = S.h0[e];
    double2 h1 = S.h1[e];
    uint8_t* body = S.body + (size_t)e * S.body_stride;
    Lane<WT> ln{S, C, T.hot, e, body, h0.z & 0xFFFFu, h0.z >> 16, h0.w & 0xFFFFu, h0.w >> 16, {}, 0u, 0ull, 0u, 0u, 0u, false, true, 0u, 0u,
                0.0, MCBS_OUT_NONE, 0, 0, 0};
    uint64_t m0[M_COUNT][WT];
#pragma unroll
    for (int k = 0; k < M_COUNT; ++k)
#pragma unroll
        for (int w = 0; w < WT; ++w) {
            const bool wanted = k == M_INST || k == M_RUN || k == M_PLO || k == M_PHI;
            m0[k][w] = wanted ? S.get(k, (uint32_t)w, e) : 0ull;
            ln.m[k][w] = m0[k][w];
        }
    uint64_t back[WT], fresh[WT];
#pragma unroll
    for (int w = 0; w < WT; ++w) { back[w] = S.ring[((ln.dclk & 15u) * WT + (uint32_t)w) * S.E + e]; fresh[w] = 0ull; }

    // ---- is_defender_action_valid, on the state BEFORE this turn's tick ----
    // all twelve components are loaded, then picked with selects: a divergent branch on the loaded kind whose arms load different
    // components into the same variables is what ROCm 7.2's compiler mis-structurised in decode_kernel (tools/decode_repro/README.md)
    const int64_t* ap = actions + (size_t)e * 12;
    int64_t a[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) a[i] = ap[i];
    const int kind = (int)a[0];
    if (kind <= -2) {                                                // env not taking part in this turn (e.g. its episode just ended)
        ok_out = false; avail_out = h1.y; evicted_out = false;
        return;
    }
    const int node = kind == 0 ? (int)a[1] : kind == 1 ? (int)a[2] : kind == 2 ? (int)a[5] : kind == 3 ? (int)a[8] : kind == 4 ? (int)a[10] : -1;
    const int rule = kind == 1 ? (int)a[3] : (int)a[6];
    const bool incoming = (kind == 1 ? a[4] : a[7]) != 0;
    const int service = kind == 3 ? (int)a[9] : (int)a[11];
    bool ok = kind < 0;                                              // the empty action is a valid no-op (defend_wrapper.py:207-212)
    uint16_t* fwp = reinterpret_cast<uint16_t*>(body + S.off_fw);
    uint32_t l_examined = 0, l_in = 0, fw = 0;
    if (kind >= 0 && kind <= 4 && node >= 0 && node < (int)S.N && rget<WT>(ln.m[M_RUN], (uint32_t)node)) {
        const HotNode* hn = ln.NS((uint32_t)node);
        const uint32_t lists = reinterpret_cast<const uint32_t*>(T.hot + C.hot_fwlist)[node];
        l_in = lists & 0xFFFFu;
        l_examined = incoming ? l_in : (lists >> 16);
        fw = fwp[l_examined];
        if (kind == 0) ok = (hn->flags & MCBS_NODE_REIMAGABLE) != 0;
        else if (kind == 1) ok = rule >= 0 && rule < 6 && ((fw >> rule) & 1u);                           // firewall_rule_exists
        else if (kind == 2) ok = true;
        else ok = service >= 0 && service < (int)hn->svc_cnt;                                           // service_exists
    }
    // ---- executeAction: the tick first, always ----
    h1.y = ln.defender_tick(back);
    if (ok && kind == 0) ln.reimage((uint32_t)node, fresh);
    else if (ok && kind == 1) fwp[l_examined] = (uint16_t)(fw & ~((1u | (1u << 6)) << rule));            // every rule with that name is removed
    else if (ok && kind == 2 && rule >= 0 && rule < 6 && !((fw >> rule) & 1u)) {
        // the examined list has no such rule: an ALLOW rule is appended to the node's INCOMING list; it becomes that list's
        // first (only) match unless the list already has a rule with the name
        const uint32_t fin = fwp[l_in];
        if (!((fin >> rule) & 1u)) fwp[l_in] = (uint16_t)(fin | ((1u | (1u << 6)) << rule));
    }
    // stop_service / start_service: valid, and without effect in the reference (defender.py:45-48 vs actions.py:782-794)
#pragma unroll
    for (int w = 0; w < WT; ++w)
        if (fresh[w] != back[w]) S.ring[((ln.dclk & 15u) * WT + (uint32_t)w) * S.E + e] = fresh[w];
    ln.dclk = (ln.dclk + 1u) & 0xFFFFu;
    S.h0[e] = make_uint4(h0.x, h0.y, h0.z, ln.owned | (ln.dclk << 16));
    S.h1[e] = h1;
    if (ln.dirty) {
#pragma unroll
        for (int k = 0; k < M_COUNT; ++k) {
            if (!((ln.dirty >> k) & 1u)) continue;
#pragma unroll
            for (int w = 0; w < WT; ++w)
                if (WT == 1 || ln.m[k][w] != m0[k][w]) S.put(k, (uint32_t)w, e, ln.m[k][w]);
        }
    }
    ok_out = ok; avail_out = h1.y;
    evicted_out = C.defender_goal_eviction && ln.owned == 0;          // __defender_goal_reached (env.py:1112-1116)
}

// DefenderEnvWrapper.observe (defend_wrapper.py:492-534) written by the TURN kernel's own workgroup, for topologies of up to 32 nodes: the
// separate observation launch (a thread per (env, node), three dependent loads, 13 byte stores) took as long as the turn itself (6.5 of
// the 12 us of a DefenderVecEnv.step at 16 384 ToyCtf envs).  Each lane parks its env's bits in LDS after its turn (agent-installed word;
// per node the six rule-name bits of the incoming and the outgoing list, read back from its own stores), and the workgroup's 128 envs —
// one contiguous region of each output array — leave as 16-byte stores, thread by thread.  (A first version that had every lane store
// its own env's ~80 bytes / shorts was SLOWER than the extra launch: 14.4 against 12.2 us — 64 scattered partial lines per instruction.)
struct DefObs {
    int8_t* infected; int8_t* fw_in; int8_t* fw_out; int8_t* services;
    uint32_t n_services, fused;
    FastDiv dN, d6N, dS;                 // divisors: nodes, six bytes per node, services
};
constexpr uint32_t DEF_WG = 128u, DEF_NMAX = 32u, DEF_SMAX = 256u;
struct DefStage { uint32_t inst[DEF_WG]; uint16_t fw[DEF_WG * DEF_NMAX]; uint8_t svc[DEF_SMAX]; };      // 8.9 KB per workgroup

template <int WT>
__device__ __forceinline__ void defender_observe_wg(const DevState& S, const Topo& T, const DefObs& o, const uint32_t e, const bool valid, DefStage& st) {
    const uint32_t tid = threadIdx.x, N = S.N, e0 = blockIdx.x * DEF_WG;
    const uint32_t n_env = S.E - e0 < DEF_WG ? S.E - e0 : DEF_WG;
    if (valid) {
        const uint8_t* body = S.body + (size_t)e * S.body_stride;
        const uint16_t* fw = reinterpret_cast<const uint16_t*>(body + S.off_fw);
        const mcbs_node_static* NS = reinterpret_cast<const mcbs_node_static*>(T.base + T.H().off_node);
        st.inst[tid] = (uint32_t)S.get(M_INST, 0u, e);                   // N <= 32: the low bits of word 0
        for (uint32_t n = 0; n < N; ++n) {                               // (uniform: node statics through the scalar cache)
            const uint32_t lists = NS[n].fw_lists;
            st.fw[tid * DEF_NMAX + n] = (uint16_t)((fw[lists & 0xFFFFu] & 63u) | ((fw[lists >> 16] & 63u) << 8));
        }
    }
    if (o.services) {                                                    // services never change state (see below): the initial flags
        const mcbs_service* sv = reinterpret_cast<const mcbs_service*>(T.base + T.H().off_service);
        for (uint32_t k = tid; k < o.n_services; k += DEF_WG) st.svc[k] = sv[k].running ? 1 : 0;
    }
    __syncthreads();
    // byte j of a 16-byte chunk held in four dwords
#define MCBS_PUT_BYTE(w, j, v) w[(j) >> 2] |= (uint32_t)(v) << (8u * ((j) & 3u))
    if (o.infected) {
        int8_t* p = o.infected + (size_t)e0 * N;
        const uint32_t R = n_env * N;
        for (uint32_t b0 = tid * 16u; b0 < R; b0 += DEF_WG * 16u) {
            uint32_t env = fdiv(b0, o.dN), node = b0 - env * N, bits = st.inst[env], w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) {
                MCBS_PUT_BYTE(w, j, (bits >> node) & 1u);
                if (++node == N) { node = 0; env = env + 1u < DEF_WG ? env + 1u : env; bits = st.inst[env]; }
            }
            if (b0 + 16u <= R) *reinterpret_cast<uint4*>(p + b0) = make_uint4(w[0], w[1], w[2], w[3]);
            else {
#pragma unroll
                for (uint32_t j = 0; j < 16u; ++j)
                    if (b0 + j < R) p[b0 + j] = (int8_t)((w[j >> 2] >> (8u * (j & 3u))) & 0xFFu);
            }
        }
    }
    auto stream_fw = [&](int8_t* dst, const uint32_t shift) {            // six 0 / 1 bytes per (env, node): two per step (6 is even)
        int8_t* p = dst + (size_t)e0 * 6u * N;
        const uint32_t R = n_env * 6u * N;
        for (uint32_t b0 = tid * 16u; b0 < R; b0 += DEF_WG * 16u) {
            uint32_t env = fdiv(b0, o.d6N);
            const uint32_t rem = b0 - env * 6u * N;
            uint32_t node = rem / 6u, k = rem - node * 6u, f = (uint32_t)st.fw[env * DEF_NMAX + node] >> shift, w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t j = 0; j < 16u; j += 2u) {
                MCBS_PUT_BYTE(w, j, (f >> k) & 1u);
                MCBS_PUT_BYTE(w, j + 1u, (f >> (k + 1u)) & 1u);
                k += 2u;
                if (k == 6u) {
                    k = 0;
                    if (++node == N) { node = 0; env = env + 1u < DEF_WG ? env + 1u : env; }
                    f = (uint32_t)st.fw[env * DEF_NMAX + node] >> shift;
                }
            }
            if (b0 + 16u <= R) *reinterpret_cast<uint4*>(p + b0) = make_uint4(w[0], w[1], w[2], w[3]);
            else {
#pragma unroll
                for (uint32_t j = 0; j < 16u; j += 2u)
                    if (b0 + j < R) *reinterpret_cast<uint16_t*>(p + b0 + j) = (uint16_t)((w[j >> 2] >> (8u * (j & 3u))) & 0xFFFFu);
            }
        }
    };
    if (o.fw_in) stream_fw(o.fw_in, 0u);
    if (o.fw_out) stream_fw(o.fw_out, 8u);
    if (o.services && o.n_services) {
        const uint32_t Sv = o.n_services, R = n_env * Sv;
        int8_t* p = o.services + (size_t)e0 * Sv;
        for (uint32_t b0 = tid * 16u; b0 < R; b0 += DEF_WG * 16u) {
            uint32_t k = b0 - fdiv(b0, o.dS) * Sv, w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) {
                MCBS_PUT_BYTE(w, j, st.svc[k]);
                if (++k == Sv) k = 0;
            }
            if (b0 + 16u <= R) *reinterpret_cast<uint4*>(p + b0) = make_uint4(w[0], w[1], w[2], w[3]);
            else {
#pragma unroll
                for (uint32_t j = 0; j < 16u; ++j)
                    if (b0 + j < R) p[b0 + j] = (int8_t)((w[j >> 2] >> (8u * (j & 3u))) & 0xFFu);
            }
        }
    }
#undef MCBS_PUT_BYTE
}

template <int WT>
__global__ __launch_bounds__(128) void defender_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, const int64_t* actions,
                                                       uint8_t* valid_out, double* avail_out, uint8_t* evicted_out, DefObs obs) {
    __shared__ DefStage stage;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e < S.E;
    if (valid) {
        bool ok, evicted;
        double avail;
        defender_turn<WT>(S, T, *Cp, actions, e, ok, avail, evicted);
        if (valid_out) valid_out[e] = ok ? 1 : 0;
        if (avail_out) avail_out[e] = avail;
        if (evicted_out) evicted_out[e] = evicted ? 1 : 0;
    }
    if (obs.fused) defender_observe_wg<WT>(S, T, obs, e, valid, stage);        // (uniform: the whole workgroup)
}

// DefenderEnvWrapper.step's reward shaping (defend_wrapper.py:228-282) for one env, from the turn's results (in registers or loaded)
__device__ __forceinline__ void defender_shape(const mcbs_defender_wrapper_buffers& w, const mcbs_defender_wrapper_cfg& c, uint32_t e, bool valid, double avail,
                                               bool won) {
    if (valid) w.valid_action_count[e] += 1; else w.invalid_action_count[e] += 1;
    double reward = (valid ? 0.0 : 1.0) * c.invalid_action_penalty;
    reward = reward - (w.attacker_has_cyber_reward[e] ? (double)w.attacker_last_cyber_reward[e] : 0.0);
    const double worsening = w.prev_availability[e] - avail;
    const bool breached = avail < c.maintain_sla, had = w.has_breached_sla[e] != 0;
    const bool first = breached && !had;
    reward = reward + (first ? 1.0 : 0.0) * c.loss_reward;
    const bool again = breached && had && worsening > 0.0;
    reward = reward + (again ? __dmul_rn(-c.sla_worsening_penalty_scale, worsening) : 0.0);   // (a separately rounded product, as on the host)
    bool terminated = c.reset_on_constraint_broken ? first : false;
    w.has_breached_sla[e] = breached ? 1 : 0;
    w.prev_availability[e] = avail;
    if (won) reward = c.winning_reward;
    terminated = terminated || won;
    const int32_t t = w.timesteps[e] + 1;
    w.timesteps[e] = t;
    w.reward[e] = reward;
    w.terminated[e] = terminated ? 1 : 0;
    w.truncated[e] = t >= c.max_timesteps ? 1 : 0;
    w.breached[e] = breached ? 1 : 0;
    w.won[e] = won ? 1 : 0;
}

__global__ __launch_bounds__(256) void defender_wrapper_post_kernel(uint32_t E, mcbs_defender_wrapper_buffers w, mcbs_defender_wrapper_cfg c) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    defender_shape(w, c, e, w.valid[e] != 0, w.availability[e], w.evicted[e] != 0);
}

// mcbs_defender_wrapper_step: the defender's turn AND the wrapper's reward shaping in one launch (the shaping takes the turn's results
// from registers; valid / availability / evicted are still written for the caller)
template <int WT>
__global__ __launch_bounds__(128) void defender_turn_post_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, const int64_t* actions,
                                                                 mcbs_defender_wrapper_buffers w, mcbs_defender_wrapper_cfg c, DefObs obs) {
    __shared__ DefStage stage;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e < S.E;
    if (valid) {
        bool ok, evicted;
        double avail;
        defender_turn<WT>(S, T, *Cp, actions, e, ok, avail, evicted);
        const_cast<uint8_t*>(w.valid)[e] = ok ? 1 : 0;
        const_cast<double*>(w.availability)[e] = avail;
        const_cast<uint8_t*>(w.evicted)[e] = evicted ? 1 : 0;
        defender_shape(w, c, e, ok, avail, evicted);
    }
    if (obs.fused) defender_observe_wg<WT>(S, T, obs, e, valid, stage);        // (uniform: the whole workgroup)
}

__global__ __launch_bounds__(256) void defender_obs_kernel(DevState S, Topo T, int8_t* infected, int8_t* fw_in, int8_t* fw_out,
                                                           int8_t* services, uint32_t n_services) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= S.E * S.N) return;
    const uint32_t e = g / S.N, n = g - e * S.N;
    if (infected) infected[g] = (int8_t)S.has(M_INST, n, e);
    const uint16_t* fw = reinterpret_cast<const uint16_t*>(S.body + (size_t)e * S.body_stride + S.off_fw);
    const uint32_t lists = reinterpret_cast<const mcbs_node_static*>(T.base + T.H().off_node)[n].fw_lists;
    const uint32_t fin = fw[lists & 0xFFFFu], fout = fw[lists >> 16];
    for (uint32_t k = 0; k < 6u; ++k) {
        if (fw_in) fw_in[(size_t)g * 6 + k] = (int8_t)((fin >> k) & 1u);
        if (fw_out) fw_out[(size_t)g * 6 + k] = (int8_t)((fout >> k) & 1u);
    }
    if (services && n == 0) {                                          // services never change state (see above): the initial flags
        const mcbs_service* sv = reinterpret_cast<const mcbs_service*>(T.base + T.H().off_service);
        for (uint32_t k = 0; k < n_services; ++k) services[(size_t)e * n_services + k] = (int8_t)(sv[k].running ? 1 : 0);
    }
}

} // namespace mcbs
