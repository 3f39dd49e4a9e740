// mcbs_ere.hip — the ExternalRandomEvents defender (MCBS_DEFENDER_RANDOM_EVENTS), SURVEY.md section 8f-3.
//
// ExternalRandomEvents.step (cyberbattle/_env/defender.py:58-148) runs five passes over EVERY node, every step; each node draws
// numpy.random.random() <= 0.1 and, when that fires, random.choice(...) once or twice:
//   1 patch_vulnerabilities_at_random : pop a random key of the node's own vulnerability dictionary
//   2 stop_service_at_random          : DefenderAgentActions.stop_service(node, random service's name) — every service of that name
//   3 plant_vulnerabilities_at_random : add a random library vulnerability the node's dictionary does not hold (numpy.setdiff1d:
//                                       chosen from the NAME-SORTED difference)
//   4 firewall_change_remove          : remove the first rule equal to a random rule of the incoming or the outgoing list
//   5 firewall_change_add             : append ALLOW on a random SAMPLE_IDENTIFIERS port to the incoming or outgoing list, unless the
//                                       INCOMING list already holds that rule (both branches test the incoming list, :145-148)
// All of that is per-env state the other defenders never touch; it lives in an overlay behind the env's body (StepCfg::ere_*).
// One lane per env like the rest of the step; this path is about exactness, not speed (≈10 N Philox doubles per env-step).
// random.choice(seq) is seq[floor(u * len(seq))] here and in the fixtures' patched generators (appendix C of SURVEY.md).
#pragma once
#include "mcbs_device.h"

namespace mcbs {

struct EreView {
    uint8_t* body; const StepCfg& C; const uint8_t* blob; uint32_t N;
    __device__ __forceinline__ uint8_t* keys(uint32_t n) const { return body + C.ere_off_keys + n * C.ere_key_cap; }
    __device__ __forceinline__ uint8_t& kcnt(uint32_t n) const { return body[C.ere_off_kcnt + n]; }
    __device__ __forceinline__ uint64_t& present(uint32_t n) const { return reinterpret_cast<uint64_t*>(body + C.ere_off_present)[n]; }
    __device__ __forceinline__ uint32_t& svc(uint32_t n) const { return reinterpret_cast<uint32_t*>(body + C.ere_off_svc)[n]; }
    __device__ __forceinline__ uint16_t* list(uint32_t l) const { return reinterpret_cast<uint16_t*>(body + C.ere_off_fw + (C.ere_lists[l] & 0xFFFFu)); }
    __device__ __forceinline__ uint32_t list_cap(uint32_t l) const { return C.ere_lists[l] >> 16; }
    __device__ __forceinline__ const mcbs_node_static* NS() const { return reinterpret_cast<const mcbs_node_static*>(blob + C.off_node); }
    __device__ __forceinline__ const mcbs_service* SV() const { return reinterpret_cast<const mcbs_service*>(blob + C.off_service); }

    // __is_passing_firewall_rules (actions.py:504-515) on the env's own list: the first rule naming the port decides
    __device__ bool passes(uint32_t l, uint32_t name) const {
        const uint16_t* L = list(l);
        const uint32_t cnt = L[0];
        for (uint32_t i = 0; i < cnt; ++i) if ((L[1 + i] & 0xFFu) == name) return (L[1 + i] >> 8) & 1u;
        return false;
    }
    // _check_service_running_and_authorized (actions.py:608-621) with the env's own running flags
    __device__ bool authorized(uint32_t node, uint32_t port, uint32_t cred) const {
        const mcbs_node_static& t = NS()[node];
        const uint16_t* allowed = reinterpret_cast<const uint16_t*>(blob + C.off_allowed);
        const uint32_t run = svc(node);
        for (uint32_t i = 0; i < t.svc_cnt; ++i) {
            const mcbs_service& s = SV()[t.svc_off + i];
            if (!((run >> i) & 1u) || s.port != port) continue;
            for (uint32_t k = 0; k < s.allowed_cnt; ++k) if (allowed[s.allowed_off + k] == cred) return true;
        }
        return false;
    }
    // on_attacker_step_taken (actions.py:728-746): every machine is Running under this defender (it never re-images)
    __device__ double availability() const {
        double total_nodes = 0.0, avail = 0.0;
        for (uint32_t n = 0; n < N; ++n) {
            const mcbs_node_static& t = NS()[n];
            const uint32_t run = svc(n);
            double total = 0.0, running = 0.0;
            for (uint32_t i = 0; i < t.svc_cnt; ++i) {
                const double w = SV()[t.svc_off + i].sla_weight;
                total += w;
                running = __dadd_rn(running, __dmul_rn(w, (double)((run >> i) & 1u)));
            }
            total_nodes += t.sla_weight;
            // product and sum rounded separately, as Python floats do (actions.py:743); a fused multiply-add rounds once and can
            // differ in the last bit for non-dyadic weights (the Makefile also passes -ffp-contract=off)
            avail = __dadd_rn(avail, __dmul_rn((1.0 + running) / (1.0 + total), t.sla_weight));
        }
        return avail / total_nodes;
    }
};

template <class Draw>   // Draw: double operator()(): the next double of this env's stream for this step
__device__ void random_events_step(const EreView& V, Draw draw) {
    const double p = 0.1;
    const uint32_t N = V.N;
    const mcbs_ere_tables& E = *reinterpret_cast<const mcbs_ere_tables*>(V.blob + V.C.off_ere);
    const uint8_t* lib_sorted = reinterpret_cast<const uint8_t*>(&E) + E.off_lib_sorted;
    auto pick = [&](uint32_t n) -> uint32_t { const uint32_t i = (uint32_t)floor(draw() * (double)n); return i < n ? i : n - 1u; };
    auto rule_index = [&](uint16_t* L, uint32_t rule) -> int {
        const uint32_t cnt = L[0];
        for (uint32_t i = 0; i < cnt; ++i) if (L[1 + i] == rule) return (int)i;
        return -1;
    };
    auto remove_random = [&](uint32_t l) {           // rule = random.choice(list); list.remove(rule): the FIRST equal one goes
        uint16_t* L = V.list(l);
        const uint32_t cnt = L[0], c = pick(cnt);
        const int i = rule_index(L, L[1 + c]);
        for (uint32_t k = (uint32_t)i; k + 1u < cnt; ++k) L[1 + k] = L[2 + k];
        L[0] = (uint16_t)(cnt - 1u);
    };
    for (uint32_t n = 0; n < N; ++n) {               // 1 patch
        const bool fire = draw() <= p;
        const uint32_t cnt = V.kcnt(n);
        if (fire && cnt > 0u) {
            uint8_t* k = V.keys(n);
            const uint32_t c = pick(cnt);
            V.present(n) &= ~(1ull << k[c]);
            for (uint32_t j = c; j + 1u < cnt; ++j) k[j] = k[j + 1u];
            V.kcnt(n) = (uint8_t)(cnt - 1u);
        }
    }
    for (uint32_t n = 0; n < N; ++n) {               // 2 stop a service (and its namesakes)
        const bool fire = draw() <= p;
        const mcbs_node_static& t = V.NS()[n];
        if (fire && t.svc_cnt > 0u) {
            const uint32_t port = V.SV()[t.svc_off + pick(t.svc_cnt)].port;
            uint32_t run = V.svc(n);
            for (uint32_t i = 0; i < t.svc_cnt; ++i) if (V.SV()[t.svc_off + i].port == port) run &= ~(1u << i);
            V.svc(n) = run;
        }
    }
    for (uint32_t n = 0; n < N; ++n) {               // 3 plant
        const bool fire = draw() <= p;
        const uint64_t have = V.present(n);
        uint32_t n_new = 0;
        for (uint32_t i = 0; i < E.n_library; ++i) n_new += !((have >> lib_sorted[i]) & 1ull);
        if (fire && n_new > 0u) {
            uint32_t c = pick(n_new), col = 0;
            for (uint32_t i = 0; i < E.n_library; ++i)
                if (!((have >> lib_sorted[i]) & 1ull)) { if (c == 0u) { col = lib_sorted[i]; break; } c -= 1u; }
            const uint32_t cnt = V.kcnt(n);
            V.keys(n)[cnt] = (uint8_t)col;
            V.kcnt(n) = (uint8_t)(cnt + 1u);
            V.present(n) = have | (1ull << col);
        }
    }
    for (uint32_t n = 0; n < N; ++n) {               // 4 remove a firewall rule
        const bool fire = draw() <= p;
        const uint32_t ids = V.NS()[n].fw_lists, lin = ids & 0xFFFFu, lout = ids >> 16;
        const uint32_t nin = V.list(lin)[0], nout = V.list(lout)[0];
        if (fire && nout > 0u && nin > 0u) {
            const bool incoming = draw() <= 0.5;
            remove_random(incoming ? lin : lout);
        } else if (fire && nout > 0u) remove_random(lout);
        else if (fire && nin > 0u) remove_random(lin);
    }
    for (uint32_t n = 0; n < N; ++n) {               // 5 add a firewall rule
        const bool fire = draw() <= p;
        if (!fire) continue;
        const uint32_t rule = (uint32_t)E.sample_name[pick(7u)] | 0x100u;           // ALLOW
        const bool incoming = draw() <= 0.5;
        const uint32_t ids = V.NS()[n].fw_lists, lin = ids & 0xFFFFu, lout = ids >> 16;
        if (rule_index(V.list(lin), rule) >= 0) continue;
        uint16_t* L = V.list(incoming ? lin : lout);
        const uint32_t cnt = L[0];
        if (cnt >= V.list_cap(incoming ? lin : lout)) continue;                     // capacity (MCBS_FW_GROWTH): the rule is dropped
        L[1 + cnt] = (uint16_t)rule;
        L[0] = (uint16_t)(cnt + 1u);
    }
}

} // namespace mcbs
