// mcbs_defend.hip — the learned defender's turn, batches created with MCBS_DEFENDER_EXTERNAL.
//
//   defender_kernel     : DefenderEnvWrapper.is_defender_action_valid (marlon/baseline_models/env_wrappers/defend_wrapper.py:329-412)
//                         then LearningDefender.executeAction (marlon/defender_agents/defender.py:31-107):
//                         DefenderAgentActions.on_attacker_step_taken (actions.py:714-746) and, if valid, the action.
//                         One lane per env; shares the re-imaging ring, the availability code and reimage() with the step kernel.
//   defender_obs_kernel : DefenderEnvWrapper.observe (defend_wrapper.py:492-534), one thread per (env, node).
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

template <int WT>
__global__ __launch_bounds__(128) void defender_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, const int64_t* actions,
                                                       uint8_t* valid_out, double* avail_out, uint8_t* evicted_out) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S.E) return;
    bool ok, evicted;
    double avail;
    defender_turn<WT>(S, T, *Cp, actions, e, ok, avail, evicted);
    if (valid_out) valid_out[e] = ok ? 1 : 0;
    if (avail_out) avail_out[e] = avail;
    if (evicted_out) evicted_out[e] = evicted ? 1 : 0;
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
                                                                 mcbs_defender_wrapper_buffers w, mcbs_defender_wrapper_cfg c) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S.E) return;
    bool ok, evicted;
    double avail;
    defender_turn<WT>(S, T, *Cp, actions, e, ok, avail, evicted);
    const_cast<uint8_t*>(w.valid)[e] = ok ? 1 : 0;
    const_cast<double*>(w.availability)[e] = avail;
    const_cast<uint8_t*>(w.evicted)[e] = evicted ? 1 : 0;
    defender_shape(w, c, e, ok, avail, evicted);
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
