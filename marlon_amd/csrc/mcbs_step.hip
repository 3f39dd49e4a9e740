// mcbs_step.hip — the hot path: CyberBattleEnv.step for a whole batch in one launch (gfx950).
//
// One LANE per environment (a wavefront advances 64 envs).  The transition of one env is a short,
// branchy chain of table look-ups with almost no intra-env parallelism (all
// integer / boolean work, O(1) per step once the per-env sets are bit masks), so spending a whole
// wavefront on one env would idle 63 lanes; wave-level cooperation is used where it exists — the
// coalesced re-initialisation of an env that just ended (ballot over the lanes that need a reset, then
// all 64 lanes copy the reset image) and the observation kernels (mcbs_obs.hip).  Header columns are
// laid out env-fastest so every header access of a wave is one contiguous burst; node rows are gathered.
//
// Rules restated from the reference (citations = /root/reference/src/CyberBattleSim/cyberbattle/...):
//   _env/cyberbattle_env.py : step 1145-1185, __execute_action 707-751, index translation 584-601,
//                             discovery / credential cache append 863-907, goals 1080-1116
//   simulation/actions.py   : exploit_local 473-502, exploit_remote 425-471, __process_outcome 325-423,
//                             __mark_node_as_owned 251-275, __mark_discovered_entities 277-310,
//                             connect_to_remote_machine 524-606, reimage_node 700-712,
//                             on_attacker_step_taken 714-746, penalties / rewards 49-93
//   _env/defender.py        : ScanAndReimageCompromisedMachines.step 42-55
// Time stamps of the reference (datetime.now(), actions.py:273,407,711) are replaced by the two bits
// they decide: "ever owned" (last_owned_at is not None) and, per vulnerability slot, "attacked ever" /
// "attacked since the node's last re-imaging"; agent_installed doubles as "currently owned"
// (actions.py:517-522), see DESIGN.md "Logical time".
#pragma once
#include "mcbs_device.h"

namespace mcbs {

// ------------------------------ bit-mask columns [word][env] ------------------------------
__device__ __forceinline__ bool mtest(const uint64_t* col, uint32_t E, uint32_t e, uint32_t bit) {
    return (col[(size_t)(bit >> 6) * E + e] >> (bit & 63u)) & 1ull;
}
__device__ __forceinline__ bool mtestset(uint64_t* col, uint32_t E, uint32_t e, uint32_t bit) {
    uint64_t* p = &col[(size_t)(bit >> 6) * E + e];
    const uint64_t w = *p, m = 1ull << (bit & 63u);
    if (w & m) return true;
    *p = w | m;
    return false;
}
__device__ __forceinline__ void mset(uint64_t* col, uint32_t E, uint32_t e, uint32_t bit) {
    col[(size_t)(bit >> 6) * E + e] |= 1ull << (bit & 63u);
}
__device__ __forceinline__ void mclear(uint64_t* col, uint32_t E, uint32_t e, uint32_t bit) {
    col[(size_t)(bit >> 6) * E + e] &= ~(1ull << (bit & 63u));
}

// ------------------------------ per-lane working set ------------------------------
struct Lane {
    const DevState& S;
    const StepCfg& C;
    const uint8_t* tb;   // topology blob
    uint32_t e;
    uint8_t* body;
    uint32_t n_disc, n_creds, owned, imaging;
    // result of the attacker's action
    double raw;
    int okind, olevel, new_nodes, new_creds;

    __device__ __forceinline__ const mcbs_node_static* NS(uint32_t n) const {
        return reinterpret_cast<const mcbs_node_static*>(tb + C.off_node) + n;
    }
    __device__ __forceinline__ const mcbs_vuln_slot* SL(uint32_t n, uint32_t s) const {
        return reinterpret_cast<const mcbs_vuln_slot*>(tb + C.off_slot) + (size_t)n * C.V + s;
    }
    __device__ __forceinline__ const mcbs_payload* PL(uint32_t i) const {
        return reinterpret_cast<const mcbs_payload*>(tb + C.off_payload) + i;
    }
    __device__ __forceinline__ Row* row(uint32_t n) const { return reinterpret_cast<Row*>(body) + n; }
    __device__ __forceinline__ uint8_t* disc_list() const { return body + S.off_disc; }
    __device__ __forceinline__ uint16_t* cred_list() const { return reinterpret_cast<uint16_t*>(body + S.off_cred); }

    __device__ __forceinline__ void done_with(double r, int kind) { raw = r; okind = kind; }

    // __mark_node_as_owned (actions.py:251-275) on the row held in registers.
    // Returns "was owned at some point before" (last_owned_at is not None); `already` = currently owned.
    __device__ __forceinline__ bool mark_owned(uint32_t n, uint32_t level, uint64_t& props, uint32_t& misc, bool& already) {
        const bool ever = mtest(S.m_ever, S.E, e, n);
        already = mtest(S.m_inst, S.E, e, n);
        if (!already) {
            mset(S.m_inst, S.E, e, n);
            const uint32_t priv = misc & 0xFFu;
            const uint32_t np = priv > level ? priv : level;       // model.escalate
            if (np >= 1u && priv == 0u) { mset(S.m_priv, S.E, e, n); owned += 1; }
            misc = (misc & ~0xFFu) | np;
            props |= NS(n)->props;                                  // all (non-tag) properties become known
            if (!ever) mset(S.m_ever, S.E, e, n);
        }
        return ever;
    }

    // __process_outcome (actions.py:325-423) with __mark_discovered_entities (277-310) and the env-side
    // appends of cyberbattle_env.py:863-907 fused (both sides keep the same sets, in the same order).
    __device__ void process_outcome(uint32_t tgt, uint32_t col, double failed_penalty) {
        if (!mtest(S.m_run, S.E, e, tgt)) return done_with(0.0, MCBS_OUT_NONE);             // MACHINE_NOT_RUNNING
        const uint32_t s = (tb + C.off_slot_of)[(size_t)tgt * (C.L + C.R) + col];
        if (s == 0xFFu) return done_with(-5.0, MCBS_OUT_NONE);                               // SUPSPICIOUSNESS
        const mcbs_vuln_slot* v = SL(tgt, s);
        Row* rp = row(tgt);
        const uint4 r0 = *reinterpret_cast<const uint4*>(rp);
        uint64_t props = (uint64_t)r0.x | ((uint64_t)r0.y << 32);
        uint32_t ever = r0.z, since = r0.w, misc = rp->misc;
        const uint32_t tags = (misc >> 8) & 0xFu;
        const uint32_t kind = v->kind;
        if (!((v->precond_tt >> tags) & 1u)) return done_with(failed_penalty, MCBS_OUT_EXPLOIT_FAILED);

        int r = 0;
        if (kind == MCBS_OUT_PRIVILEGE_ESCALATION) {
            const uint32_t level = v->level;
            olevel = (int)level;
            if ((tags >> level) & 1u) return done_with(-1.0, MCBS_OUT_PRIVILEGE_ESCALATION);  // REPEAT, nothing recorded
            bool already;
            if (!mark_owned(tgt, level, props, misc, already)) r += NS(tgt)->value;
            misc |= (1u << level) << 8;
        } else if (kind == MCBS_OUT_LATERAL_MOVE) {
            bool already;
            if (!mark_owned(tgt, 1u, props, misc, already)) r += NS(tgt)->value;
        } else if (kind == MCBS_OUT_PROBE_SUCCEEDED) {
            const uint64_t pm = v->probe_mask;
            r += 2 * __popcll(pm & ~props);
            props |= pm;
        }
        const uint32_t bit = 1u << s;
        if (ever & bit) { if (since & bit) r -= 1; } else r += 7;
        ever |= bit; since |= bit;
        *reinterpret_cast<uint4*>(rp) = make_uint4((uint32_t)props, (uint32_t)(props >> 32), ever, since);
        rp->misc = misc;

        int nn = 0, nc = 0;
        if (kind == MCBS_OUT_LEAKED_CREDENTIALS || kind == MCBS_OUT_LEAKED_NODES) {
            const uint32_t off = v->payload_off, cnt = v->payload_cnt;
            for (uint32_t i = 0; i < cnt; ++i) {
                const mcbs_payload p = *PL(off + i);
                if (!mtestset(S.m_disc, S.E, e, p.node)) { disc_list()[n_disc++] = (uint8_t)p.node; nn++; }
                if (kind == MCBS_OUT_LEAKED_CREDENTIALS) {
                    if (!mtestset(S.m_gath, S.E, e, p.cred)) nc++;
                    if (!mtestset(S.m_cach, S.E, e, p.triple)) { cred_list()[n_creds++] = p.triple; new_creds++; }
                }
            }
        }
        new_nodes = nn;
        r += 5 * nn + 3 * nc;
        done_with((double)r - v->cost, (int)kind);
    }

    // connect_to_remote_machine (actions.py:524-606); cred_idx already checked against the cache length
    __device__ void connect(uint32_t src, uint32_t tgt, uint32_t port, uint32_t cred_idx) {
        if (!mtest(S.m_inst, S.E, e, src)) return done_with(-1.0, MCBS_OUT_NONE);
        // target is discovered and the credential gathered by construction (both come from this env's own lists)
        const uint32_t triple = cred_list()[cred_idx];
        const uint32_t cred = (reinterpret_cast<const mcbs_triple*>(tb + C.off_triple) + triple)->cred;
        const mcbs_node_static* t = NS(tgt);
        if (!((NS(src)->fw_out_allow >> port) & 1u)) return done_with(-10.0, MCBS_OUT_NONE);  // BLOCKED_BY_LOCAL_FIREWALL
        if (!((t->fw_in_allow >> port) & 1u)) return done_with(-10.0, MCBS_OUT_NONE);         // BLOCKED_BY_REMOTE_FIREWALL
        if (!((t->listen >> port) & 1u)) return done_with(-10.0, MCBS_OUT_NONE);              // SCANNING_UNOPEN_PORT
        if (!mtest(S.m_run, S.E, e, tgt)) return done_with(0.0, MCBS_OUT_NONE);               // MACHINE_NOT_RUNNING
        bool authorized = false;                                                              // actions.py:608-621
        const mcbs_service* sv = reinterpret_cast<const mcbs_service*>(tb + C.off_service) + t->svc_off;
        const uint16_t* allowed = reinterpret_cast<const uint16_t*>(tb + C.off_allowed);
        for (uint32_t i = 0; i < t->svc_cnt; ++i) {
            if (!sv[i].running || sv[i].port != port) continue;
            for (uint32_t k = 0; k < sv[i].allowed_cnt; ++k) authorized |= (allowed[sv[i].allowed_off + k] == cred);
        }
        if (!authorized) return done_with(-10.0, MCBS_OUT_NONE);                              // WRONG_PASSWORD
        Row* rp = row(tgt);
        const uint4 r0 = *reinterpret_cast<const uint4*>(rp);
        uint64_t props = (uint64_t)r0.x | ((uint64_t)r0.y << 32);
        uint32_t misc = rp->misc;
        bool already;
        const bool ever = mark_owned(tgt, 1u, props, misc, already);
        if (already) return done_with(-1.0, MCBS_OUT_LATERAL_MOVE);                            // REPEAT
        *reinterpret_cast<uint2*>(rp) = make_uint2((uint32_t)props, (uint32_t)(props >> 32));
        rp->misc = misc;
        done_with(ever ? 0.0 : (double)t->value, MCBS_OUT_LATERAL_MOVE);
    }

    // __execute_action (cyberbattle_env.py:707-751).  Returns true on the OutOfBoundIndexError path.
    __device__ bool attacker(const int32_t* a) {
        raw = 0.0; okind = MCBS_OUT_NONE; olevel = 0; new_nodes = 0; new_creds = 0;
        const int kind = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4];
        const int nd = (int)n_disc;
        if (kind == 0) {
            if (a1 < 0 || a1 >= nd || a2 < 0 || a2 >= (int)C.L) return true;
            const uint32_t src = disc_list()[a1];
            if (!mtest(S.m_inst, S.E, e, src)) done_with(-1.0, MCBS_OUT_NONE);                 // INVALID_ACTION
            else process_outcome(src, (uint32_t)a2, -20.0);                                    // LOCAL_EXPLOIT_FAILED
        } else if (kind == 1) {
            if (a1 < 0 || a1 >= nd || a2 < 0 || a2 >= nd || a3 < 0 || a3 >= (int)C.R) return true;
            const uint32_t src = disc_list()[a1], tgt = disc_list()[a2];
            if (!mtest(S.m_inst, S.E, e, src)) done_with(-1.0, MCBS_OUT_NONE);
            else process_outcome(tgt, C.L + (uint32_t)a3, -50.0);                              // FAILED_REMOTE_EXPLOIT
        } else if (kind == 2) {
            if (a4 < 0 || a4 >= (int)n_creds) { done_with(-1.0, MCBS_OUT_NONE); return false; } // env.py:736-737
            if (a1 < 0 || a1 >= nd || a2 < 0 || a2 >= nd || a3 < 0 || a3 >= (int)C.P) return true;
            connect(disc_list()[a1], disc_list()[a2], (uint32_t)a3, (uint32_t)a4);
        } else {
            return true;
        }
        return false;
    }

    // ---- defender ----
    __device__ __forceinline__ double draw(uint32_t i, uint32_t step, const StepIO& io) const {
        if (C.rng_kind == MCBS_RNG_TAPE) return (io.tape && i < io.tape_dps) ? io.tape[(size_t)e * io.tape_dps + i] : 0.0;
        const uint64_t gid = C.env_id_base + e;
        uint32_t r[4];
        philox4x32_10((uint32_t)gid, S.episode[e], step, i >> 1, (uint32_t)C.seed,
                      (uint32_t)(C.seed >> 32) ^ (uint32_t)(gid >> 32), r);
        return (i & 1u) ? to_double53(r[2], r[3]) : to_double53(r[0], r[1]);
    }

    __device__ __forceinline__ uint64_t valid_bits(uint32_t w) const {
        const uint32_t rem = S.N - w * 64u;
        return rem >= 64u ? ~0ull : ((1ull << rem) - 1ull);
    }

    // on_attacker_step_taken (actions.py:714-746) -> availability
    __device__ double defender_tick() {
        if (imaging) {
            for (uint32_t w = 0; w < S.NW; ++w) {
                uint64_t* prun = &S.m_run[(size_t)w * S.E + e];
                const uint64_t run = *prun;
                uint64_t im = ~run & valid_bits(w), nrun = run;
                while (im) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(im);
                    im &= im - 1;
                    uint32_t* pm = &row(w * 64u + b)->misc;
                    const uint32_t misc = *pm;
                    if ((misc >> 16) & 0xFFu) *pm = misc - (1u << 16);
                    else { nrun |= 1ull << b; imaging -= 1; }
                }
                if (nrun != run) *prun = nrun;
            }
        }
        if (!imaging) return C.full_availability;
        double s;
        if (C.avail_any_order) {          // exact in any order: subtract the terms of the nodes being re-imaged
            s = C.full_sum;
            for (uint32_t w = 0; w < S.NW; ++w) {
                uint64_t im = ~S.m_run[(size_t)w * S.E + e] & valid_bits(w);
                while (im) { const uint32_t b = (uint32_t)__builtin_ctzll(im); im &= im - 1; s -= NS(w * 64u + b)->avail_term; }
            }
        } else {                          // the reference's node-order sum
            s = 0.0;
            for (uint32_t n = 0; n < S.N; ++n) if (mtest(S.m_run, S.E, e, n)) s += NS(n)->avail_term;
        }
        return s / C.total_sla_weight;
    }

    // ScanAndReimageCompromisedMachines.step (defender.py:42-55) + reimage_node (actions.py:700-712)
    __device__ void defender_scan(uint32_t step, const StepIO& io) {
        if (step % C.scan_frequency) return;
        uint32_t det = 0;
        for (uint32_t i = 0; i < C.scan_capacity; ++i) {
            int n = (int)floor(draw(i, step, io) * (double)S.N);
            if (n >= (int)S.N) n = (int)S.N - 1;
            if (!mtest(S.m_run, S.E, e, (uint32_t)n) || !mtest(S.m_inst, S.E, e, (uint32_t)n)) continue;
            const double d = draw(C.scan_capacity + det, step, io);
            det += 1;
            if (!(d <= C.scan_probability) || !(NS((uint32_t)n)->flags & MCBS_NODE_REIMAGABLE)) continue;
            Row* rp = row((uint32_t)n);
            const uint32_t misc = rp->misc;
            rp->misc = (misc & 0x0000FF00u) | (15u << 16);       // privilege NoAccess, tags kept, REIMAGING_DURATION
            rp->since = 0;                                       // every earlier attack now predates last_reimaging
            mclear(S.m_inst, S.E, e, (uint32_t)n);
            if (misc & 0xFFu) { mclear(S.m_priv, S.E, e, (uint32_t)n); owned -= 1; }
            mclear(S.m_run, S.E, e, (uint32_t)n);
            imaging += 1;
        }
    }
};

// PHASE 0: whole step.  PHASE 1: attacker's action only (raw reward parked in S.pending).
// PHASE 2: defender, goals, outputs, auto-reset (after the observation kernels ran).
template <int PHASE>
__global__ __launch_bounds__(128) void step_kernel(DevState S, Topo T, StepCfg C, StepIO io) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    bool need_reset = false;
    if (e < S.E) {
        const uint4 h0 = S.h0[e];
        uint32_t step = h0.x, flags = h0.y;
        if (flags & (F_DONE | F_TRUNC)) {
            // step after done: the reference raises RuntimeError (env.py:1146-1147); the batch leaves the env untouched
            if (PHASE != 1) {
                io.reward[e] = 0.0f;
                io.terminated[e] = (uint8_t)(flags & F_DONE);
                if (io.truncated) io.truncated[e] = (uint8_t)((flags & F_TRUNC) ? 1 : 0);
                if (io.availability) io.availability[e] = S.h1[e].y;
                if (io.step_count) io.step_count[e] = (int32_t)step;
                if (io.oob) io.oob[e] = 0;
                if (io.raw_reward) io.raw_reward[e] = 0.0f;
            }
        } else {
            Lane ln{S, C, T.base, e, S.body + (size_t)e * S.body_stride, h0.z & 0xFFFFu, h0.z >> 16, h0.w & 0xFFFFu, h0.w >> 16,
                    0.0, MCBS_OUT_NONE, 0, 0, 0};
            bool oob;
            if (PHASE != 2) {
                step += 1;
                const int32_t* ap = io.actions + (size_t)e * 5;
                const int32_t a[5] = {ap[0], ap[1], ap[2], ap[3], ap[4]};
                oob = ln.attacker(a);
                flags = (oob ? F_OOB : 0u) | ((uint32_t)ln.okind << F_KIND_SHIFT) | ((uint32_t)ln.olevel << F_LEVEL_SHIFT) |
                        ((uint32_t)ln.new_nodes << F_NEWNODES_SHIFT) | ((uint32_t)ln.new_creds << F_NEWCREDS_SHIFT);
                if (oob) ln.raw = 0.0;
            } else {
                oob = (flags & F_OOB) != 0;
                ln.raw = S.pending[e];
            }
            if (PHASE == 1) {
                S.pending[e] = ln.raw;
                S.h0[e] = make_uint4(step, flags, ln.n_disc | (ln.n_creds << 16), ln.owned | (ln.imaging << 16));
            } else {
                double2 h1 = S.h1[e];
                double reward = 0.0;
                bool done = false;
                if (!oob) {
                    const bool has_def = C.defender_kind != MCBS_DEFENDER_NONE;
                    if (has_def) {
                        h1.y = ln.defender_tick();
                        ln.defender_scan(step, io);
                    }
                    // goals (env.py:1080-1116) on the state AFTER the defender acted, availability from BEFORE its scan
                    bool attacker_goal = C.has_attacker_goal != 0;
                    if (attacker_goal) {
                        if (h1.x < C.goal_reward) attacker_goal = false;
                        else if (ln.owned < C.goal_own_atleast) attacker_goal = false;
                        else if ((double)ln.owned / (double)S.N < C.goal_own_atleast_percent) attacker_goal = false;
                        else if (has_def && h1.y >= C.goal_low_availability) attacker_goal = false;
                    }
                    const bool sla_broken = has_def && h1.y < C.maintain_sla;
                    const bool evicted = C.defender_goal_eviction && ln.owned == 0;
                    if (attacker_goal || sla_broken) { done = true; reward = C.winning_reward; }
                    else if (evicted) { done = true; reward = C.losing_reward; }
                    else reward = ln.raw > 0.0 ? ln.raw : 0.0;                                  // max(0, reward), env.py:1169
                }
                h1.x += reward;
                const bool trunc = !done && C.max_episode_steps && step >= C.max_episode_steps;
                io.reward[e] = (float)reward;
                io.terminated[e] = done ? 1 : 0;
                if (io.truncated) io.truncated[e] = trunc ? 1 : 0;
                if (io.availability) io.availability[e] = h1.y;
                if (io.step_count) io.step_count[e] = (int32_t)step;
                if (io.oob) io.oob[e] = oob ? 1 : 0;
                if (io.raw_reward) io.raw_reward[e] = (float)ln.raw;
                if ((done || trunc) && C.auto_reset) need_reset = true;
                else {
                    flags |= (done ? F_DONE : 0u) | (trunc ? F_TRUNC : 0u);
                    S.h0[e] = make_uint4(step, flags, ln.n_disc | (ln.n_creds << 16), ln.owned | (ln.imaging << 16));
                    S.h1[e] = h1;
                }
            }
        }
    }
    if (PHASE != 1) {
        // Envs that just ended are re-initialised by the whole wavefront: ballot the lanes that need it, then all
        // 64 lanes copy the reset image of one env at a time with 16-byte accesses (coalesced), instead of one
        // lane writing N rows serially.
        const uint64_t rm = __ballot(need_reset);
        if (rm) {
            __threadfence_block();   // the owner lane's row stores must land before other lanes overwrite them
            const uint32_t lane = threadIdx.x & 63u;
            const uint32_t wave_base = e - lane;
            uint64_t m = rm;
            while (m) {
                const uint32_t l = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                uint8_t* dst = S.body + (size_t)(wave_base + l) * S.body_stride;
                for (uint32_t off = lane * 16u; off < S.body_stride; off += 64u * 16u)
                    *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(S.init_body + off);
            }
            if (need_reset) reset_header(S, T, e, S.episode[e] + 1u);
        }
    }
}

template __global__ void step_kernel<0>(DevState, Topo, StepCfg, StepIO);
template __global__ void step_kernel<1>(DevState, Topo, StepCfg, StepIO);
template __global__ void step_kernel<2>(DevState, Topo, StepCfg, StepIO);

} // namespace mcbs
