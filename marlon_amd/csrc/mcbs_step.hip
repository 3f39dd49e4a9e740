// mcbs_step.hip — the hot path: CyberBattleEnv.step for a whole batch in one launch (gfx950).
//
// One LANE per environment (a wavefront advances 64 envs).  The transition of one env is a short,
// branchy chain of table look-ups with almost no intra-env parallelism (all
// integer / boolean work, O(1) per step once the per-env sets are bit masks), so spending a whole
// wavefront on one env would idle 63 lanes; wave-level cooperation is used where it exists — the
// coalesced re-initialisation of an env that just ended (ballot over the lanes that need a reset, then
// all 64 lanes copy the reset image), the cooperative staging of the topology tables in LDS, and the
// observation kernels (mcbs_obs.hip).
//
// At BASELINE.json's batch (65 536 envs = 1 024 wavefronts on 1 024 SIMDs) every wavefront is resident at
// once, so the launch takes as long as ONE wavefront's dependent chain of memory accesses.  The kernel is
// therefore organised by dependency level, not by reference function:
//   level 1 (addresses depend on the env index only, issued back to back, all coalesced along the env axis):
//           header uint4, action row, first 16 discovery-order entries, first 16 credential-cache entries,
//           the agent-installed and running node masks, {cum_reward, availability};
//           meanwhile the workgroup copies the topology tables into LDS;
//   level 2 (address depends on the action): the target node's 32-byte row (and list entries beyond the
//           first 16 for large topologies);
//   then pure ALU + LDS look-ups (vulnerability slot, payload, firewall / service tables), and one round
//   of stores (row, masks, header, outputs).  Rare events (discoveries, ownership changes, re-imaging)
//   touch further mask columns / rows lazily.
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

// ------------------------------ bit-mask columns [word][env] in memory ------------------------------
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

// ------------------------------ node masks held in registers (NWT words) ------------------------------
template <int NWT>
__device__ __forceinline__ bool rget(const uint64_t (&m)[NWT], uint32_t n) {
    uint64_t w = m[0];
#pragma unroll
    for (int i = 1; i < NWT; ++i) if ((n >> 6) == (uint32_t)i) w = m[i];
    return (w >> (n & 63u)) & 1ull;
}
template <int NWT>
__device__ __forceinline__ void rset(uint64_t (&m)[NWT], uint32_t n) {
#pragma unroll
    for (int i = 0; i < NWT; ++i) if ((n >> 6) == (uint32_t)i) m[i] |= 1ull << (n & 63u);
}
template <int NWT>
__device__ __forceinline__ void rclear(uint64_t (&m)[NWT], uint32_t n) {
#pragma unroll
    for (int i = 0; i < NWT; ++i) if ((n >> 6) == (uint32_t)i) m[i] &= ~(1ull << (n & 63u));
}

__device__ __forceinline__ uint32_t pick4(const uint4& v, uint32_t i) {
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

// ------------------------------ per-lane working set ------------------------------
template <int NWT>
struct Lane {
    const DevState& S;
    const StepCfg& C;
    const uint8_t* tb;   // topology tables (LDS copy or the blob in global memory)
    uint32_t e;
    uint8_t* body;
    uint32_t n_disc, n_creds, owned, imaging;
    uint64_t inst[NWT], run[NWT];
    // the target node's row, in registers
    uint64_t props;
    uint32_t ever, since, misc;
    bool row_dirty;
    // result of the attacker's action
    double raw;
    int okind, olevel, new_nodes, new_creds;

    __device__ __forceinline__ const mcbs_node_static* NS(uint32_t n) const {
        return reinterpret_cast<const mcbs_node_static*>(tb + C.off_node) + n;
    }
    __device__ __forceinline__ Row* row(uint32_t n) const { return reinterpret_cast<Row*>(body + S.off_rows) + n; }
    __device__ __forceinline__ uint8_t* disc_list() const { return body + S.off_disc; }
    __device__ __forceinline__ uint16_t* cred_list() const { return reinterpret_cast<uint16_t*>(body + S.off_cred); }
    __device__ __forceinline__ void done_with(double r, int kind) { raw = r; okind = kind; }

    // __mark_node_as_owned (actions.py:251-275) on the row held in registers.
    // Returns "was owned at some point before" (last_owned_at is not None); `already` = currently owned.
    __device__ __forceinline__ bool mark_owned(uint32_t n, uint32_t level, bool& already) {
        already = rget<NWT>(inst, n);
        if (already) return true;                                   // currently owned implies owned before
        const bool ever_owned = mtestset(S.m_ever, S.E, e, n);
        rset<NWT>(inst, n);
        const uint32_t priv = misc & 0xFFu;
        const uint32_t np = priv > level ? priv : level;           // model.escalate
        if (np >= 1u && priv == 0u) { mset(S.m_priv, S.E, e, n); owned += 1; }
        misc = (misc & ~0xFFu) | np;
        props |= NS(n)->props;                                      // all (non-tag) properties become known
        row_dirty = true;
        return ever_owned;
    }

    // __process_outcome (actions.py:325-423) with __mark_discovered_entities (277-310) and the env-side
    // appends of cyberbattle_env.py:863-907 fused (both sides keep the same sets, in the same order).
    __device__ __forceinline__ void process_outcome(uint32_t tgt, uint32_t col, double failed_penalty) {
        if (!rget<NWT>(run, tgt)) return done_with(0.0, MCBS_OUT_NONE);                       // MACHINE_NOT_RUNNING
        const uint32_t s = (tb + C.off_slot_of)[(size_t)tgt * (C.L + C.R) + col];
        if (s == 0xFFu) return done_with(-5.0, MCBS_OUT_NONE);                               // SUPSPICIOUSNESS
        const uint4* vp = reinterpret_cast<const uint4*>(tb + C.off_slot + ((size_t)tgt * C.V + s) * sizeof(mcbs_vuln_slot));
        const uint4 v0 = vp[0], v1 = vp[1];   // {cost lo,hi, probe lo,hi} {payload_off, cnt | tt << 16, code_off, code_len | kind << 16 | level << 24}
        const uint32_t tags = (misc >> 8) & 0xFu;
        const uint32_t kind = (v1.w >> 16) & 0xFFu, level = v1.w >> 24;
        if (!(((v1.y >> 16) >> tags) & 1u)) return done_with(failed_penalty, MCBS_OUT_EXPLOIT_FAILED);

        int r = 0;
        if (kind == MCBS_OUT_PRIVILEGE_ESCALATION) {
            olevel = (int)level;
            if ((tags >> level) & 1u) return done_with(-1.0, MCBS_OUT_PRIVILEGE_ESCALATION);  // REPEAT, nothing recorded
            bool already;
            if (!mark_owned(tgt, level, already)) r += NS(tgt)->value;
            misc |= (1u << level) << 8;
        } else if (kind == MCBS_OUT_LATERAL_MOVE) {
            bool already;
            if (!mark_owned(tgt, 1u, already)) r += NS(tgt)->value;
        } else if (kind == MCBS_OUT_PROBE_SUCCEEDED) {
            const uint64_t pm = (uint64_t)v0.z | ((uint64_t)v0.w << 32);
            r += 2 * __popcll(pm & ~props);
            props |= pm;
        }
        const uint32_t bit = 1u << s;
        if (ever & bit) { if (since & bit) r -= 1; } else r += 7;
        ever |= bit; since |= bit;
        row_dirty = true;

        int nn = 0, nc = 0;
        if (kind == MCBS_OUT_LEAKED_CREDENTIALS || kind == MCBS_OUT_LEAKED_NODES) {
            const uint32_t off = v1.x, cnt = v1.y & 0xFFFFu;
            const uint2* pl = reinterpret_cast<const uint2*>(tb + C.off_payload) + off;
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint2 p = pl[i];                   // {node | cred << 16, triple | port << 16}
                const uint32_t pn = p.x & 0xFFFFu;
                if (!mtestset(S.m_disc, S.E, e, pn)) { disc_list()[n_disc++] = (uint8_t)pn; nn++; }
                if (kind == MCBS_OUT_LEAKED_CREDENTIALS) {
                    if (!mtestset(S.m_gath, S.E, e, p.x >> 16)) nc++;
                    if (!mtestset(S.m_cach, S.E, e, p.y & 0xFFFFu)) { cred_list()[n_creds++] = (uint16_t)(p.y & 0xFFFFu); new_creds++; }
                }
            }
        }
        new_nodes = nn;
        r += 5 * nn + 3 * nc;
        const double cost = __hiloint2double((int)v0.y, (int)v0.x);
        done_with((double)r - cost, (int)kind);
    }

    // connect_to_remote_machine (actions.py:524-606); the credential index was checked against the cache length
    __device__ __forceinline__ void connect(uint32_t src, uint32_t tgt, uint32_t port, uint32_t triple) {
        if (!rget<NWT>(inst, src)) return done_with(-1.0, MCBS_OUT_NONE);
        // target is discovered and the credential gathered by construction (both come from this env's own lists)
        const uint32_t cred = (reinterpret_cast<const mcbs_triple*>(tb + C.off_triple) + triple)->cred;
        const mcbs_node_static* t = NS(tgt);
        if (!((NS(src)->fw_out_allow >> port) & 1u)) return done_with(-10.0, MCBS_OUT_NONE);  // BLOCKED_BY_LOCAL_FIREWALL
        if (!((t->fw_in_allow >> port) & 1u)) return done_with(-10.0, MCBS_OUT_NONE);         // BLOCKED_BY_REMOTE_FIREWALL
        if (!((t->listen >> port) & 1u)) return done_with(-10.0, MCBS_OUT_NONE);              // SCANNING_UNOPEN_PORT
        if (!rget<NWT>(run, tgt)) return done_with(0.0, MCBS_OUT_NONE);                        // MACHINE_NOT_RUNNING
        bool authorized = false;                                                              // actions.py:608-621
        const mcbs_service* sv = reinterpret_cast<const mcbs_service*>(tb + C.off_service) + t->svc_off;
        const uint16_t* allowed = reinterpret_cast<const uint16_t*>(tb + C.off_allowed);
        const uint32_t nsv = t->svc_cnt;
        for (uint32_t i = 0; i < nsv; ++i) {
            if (!sv[i].running || sv[i].port != port) continue;
            const uint32_t ao = sv[i].allowed_off, ac = sv[i].allowed_cnt;
            for (uint32_t k = 0; k < ac; ++k) authorized |= (allowed[ao + k] == cred);
        }
        if (!authorized) return done_with(-10.0, MCBS_OUT_NONE);                              // WRONG_PASSWORD
        bool already;
        const bool ever_owned = mark_owned(tgt, 1u, already);
        if (already) return done_with(-1.0, MCBS_OUT_LATERAL_MOVE);                            // REPEAT
        done_with(ever_owned ? 0.0 : (double)t->value, MCBS_OUT_LATERAL_MOVE);
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
    __device__ __forceinline__ double defender_tick() {
        if (imaging) {
#pragma unroll
            for (int w = 0; w < NWT; ++w) {
                if ((uint32_t)w >= S.NW) break;
                uint64_t im = ~run[w] & valid_bits(w);
                while (im) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(im);
                    im &= im - 1;
                    uint32_t* pm = &row(w * 64u + b)->misc;
                    const uint32_t m = *pm;
                    if ((m >> 16) & 0xFFu) *pm = m - (1u << 16);
                    else { run[w] |= 1ull << b; imaging -= 1; }
                }
            }
        }
        if (!imaging) return C.full_availability;
        double s;
        if (C.avail_any_order) {          // exact in any order: subtract the terms of the nodes being re-imaged
            s = C.full_sum;
#pragma unroll
            for (int w = 0; w < NWT; ++w) {
                if ((uint32_t)w >= S.NW) break;
                uint64_t im = ~run[w] & valid_bits(w);
                while (im) { const uint32_t b = (uint32_t)__builtin_ctzll(im); im &= im - 1; s -= NS(w * 64u + b)->avail_term; }
            }
        } else {                          // the reference's node-order sum
            s = 0.0;
            for (uint32_t n = 0; n < S.N; ++n) if (rget<NWT>(run, n)) s += NS(n)->avail_term;
        }
        return s / C.total_sla_weight;
    }

    // ScanAndReimageCompromisedMachines.step (defender.py:42-55) + reimage_node (actions.py:700-712)
    __device__ __forceinline__ void defender_scan(uint32_t step, const StepIO& io) {
        if (step % C.scan_frequency) return;
        uint32_t det = 0;
        for (uint32_t i = 0; i < C.scan_capacity; ++i) {
            int n = (int)floor(draw(i, step, io) * (double)S.N);
            if (n >= (int)S.N) n = (int)S.N - 1;
            if (!rget<NWT>(run, (uint32_t)n) || !rget<NWT>(inst, (uint32_t)n)) continue;
            const double d = draw(C.scan_capacity + det, step, io);
            det += 1;
            if (!(d <= C.scan_probability) || !(NS((uint32_t)n)->flags & MCBS_NODE_REIMAGABLE)) continue;
            Row* rp = row((uint32_t)n);
            const uint32_t m = rp->misc;
            rp->misc = (m & 0x0000FF00u) | (15u << 16);          // privilege NoAccess, tags kept, REIMAGING_DURATION
            rp->since = 0;                                       // every earlier attack now predates last_reimaging
            rclear<NWT>(inst, (uint32_t)n);
            if (m & 0xFFu) { mclear(S.m_priv, S.E, e, (uint32_t)n); owned -= 1; }
            rclear<NWT>(run, (uint32_t)n);
            imaging += 1;
        }
    }
};

// PHASE 0: whole step.  PHASE 1: attacker's action only (raw reward parked in S.pending).
// PHASE 2: defender, goals, outputs, auto-reset (after the observation kernels ran).
// NWT: node-mask words held in registers (1, 2 or 4).  TOPO_LDS: topology tables staged in LDS.
template <int PHASE, int NWT, bool TOPO_LDS>
__global__ __launch_bounds__(256) void step_kernel(DevState S, Topo T, StepCfg C, StepIO io) {
    extern __shared__ uint4 topo_lds[];
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = e < S.E;
    const uint32_t ec = active ? e : 0u;                // clamp so inactive lanes read valid memory and take no branch

    // ---------------- level 1: loads whose addresses depend on the env index only ----------------
    uint8_t* body = S.body + (size_t)ec * S.body_stride;
    const uint4 h0 = S.h0[ec];
    uint4 a03 = make_uint4(0, 0, 0, 0);
    uint32_t a4 = 0;
    uint4 dhead = make_uint4(0, 0, 0, 0), chead0 = dhead, chead1 = dhead;
    if (PHASE != 2) {
        const uint32_t* ap = reinterpret_cast<const uint32_t*>(io.actions) + (size_t)ec * 5;
        a03 = make_uint4(ap[0], ap[1], ap[2], ap[3]);
        a4 = ap[4];
        dhead = *reinterpret_cast<const uint4*>(body + S.off_disc);
        chead0 = *reinterpret_cast<const uint4*>(body + S.off_cred);
        chead1 = *reinterpret_cast<const uint4*>(body + S.off_cred + 16);
    }
    uint64_t inst0[NWT], run0[NWT];
#pragma unroll
    for (int w = 0; w < NWT; ++w) {
        const bool have = (uint32_t)w < S.NW;
        inst0[w] = have ? S.m_inst[(size_t)w * S.E + ec] : 0ull;
        run0[w] = have ? S.m_run[(size_t)w * S.E + ec] : 0ull;
    }
    double2 h1 = make_double2(0.0, 0.0);
    if (PHASE != 1) h1 = S.h1[ec];

    const uint8_t* tb = T.base;
    if (TOPO_LDS) {                                     // cooperative copy of the topology tables, 16 bytes per lane
        const uint4* src = reinterpret_cast<const uint4*>(T.base);
        for (uint32_t i = threadIdx.x; i < C.lds_bytes / 16u; i += blockDim.x) topo_lds[i] = src[i];
        __syncthreads();
        tb = reinterpret_cast<const uint8_t*>(topo_lds);
    }

    bool need_reset = false;
    uint32_t step = h0.x, flags = h0.y;
    if (active && (flags & (F_DONE | F_TRUNC))) {
        // step after done: the reference raises RuntimeError (env.py:1146-1147); the batch leaves the env untouched
        if (PHASE != 1) {
            io.reward[e] = 0.0f;
            io.terminated[e] = (uint8_t)(flags & F_DONE);
            if (io.truncated) io.truncated[e] = (uint8_t)((flags & F_TRUNC) ? 1 : 0);
            if (io.availability) io.availability[e] = h1.y;
            if (io.step_count) io.step_count[e] = (int32_t)step;
            if (io.oob) io.oob[e] = 0;
            if (io.raw_reward) io.raw_reward[e] = 0.0f;
        }
    } else if (active) {
        Lane<NWT> ln{S, C, tb, e, body, h0.z & 0xFFFFu, h0.z >> 16, h0.w & 0xFFFFu, h0.w >> 16, {}, {}, 0ull, 0u, 0u, 0u, false,
                     0.0, MCBS_OUT_NONE, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < NWT; ++w) { ln.inst[w] = inst0[w]; ln.run[w] = run0[w]; }
        bool oob = false;
        if (PHASE != 2) {
            // ---------------- __execute_action (cyberbattle_env.py:707-751) ----------------
            step += 1;
            const int kind = (int)a03.x, a1 = (int)a03.y, a2 = (int)a03.z, a3 = (int)a03.w, a4i = (int)a4;
            const int nd = (int)ln.n_disc, ncr = (int)ln.n_creds;
            bool skip = false;                                       // connect with a credential index outside the cache
            if (kind == 0) oob = a1 < 0 || a1 >= nd || a2 < 0 || a2 >= (int)C.L;
            else if (kind == 1) oob = a1 < 0 || a1 >= nd || a2 < 0 || a2 >= nd || a3 < 0 || a3 >= (int)C.R;
            else if (kind == 2) {
                skip = a4i < 0 || a4i >= ncr;                        // env.py:736-737, before any node look-up
                oob = !skip && (a1 < 0 || a1 >= nd || a2 < 0 || a2 >= nd || a3 < 0 || a3 >= (int)C.P);
            } else oob = true;
            if (skip) ln.done_with(-1.0, MCBS_OUT_NONE);
            else if (!oob) {
                auto node_of = [&](int ext) -> uint32_t {
                    if (ext < 16) return (pick4(dhead, (uint32_t)ext >> 2) >> (8u * ((uint32_t)ext & 3u))) & 0xFFu;
                    return ln.disc_list()[ext];
                };
                const uint32_t src = node_of(a1);
                const uint32_t tgt = kind == 0 ? src : node_of(a2);
                // ---------------- level 2: the target row ----------------
                const Row* rp = ln.row(tgt);
                const uint4 r0 = *reinterpret_cast<const uint4*>(rp);
                ln.misc = rp->misc;
                ln.props = (uint64_t)r0.x | ((uint64_t)r0.y << 32);
                ln.ever = r0.z; ln.since = r0.w;
                if (kind == 2) {
                    uint32_t triple;
                    if (a4i < 16) {
                        const uint32_t d = a4i < 8 ? pick4(chead0, (uint32_t)a4i >> 1) : pick4(chead1, ((uint32_t)a4i - 8u) >> 1);
                        triple = (d >> (16u * ((uint32_t)a4i & 1u))) & 0xFFFFu;
                    } else triple = ln.cred_list()[a4i];
                    ln.connect(src, tgt, (uint32_t)a3, triple);
                } else if (!rget<NWT>(ln.inst, src)) ln.done_with(-1.0, MCBS_OUT_NONE);          // INVALID_ACTION
                else ln.process_outcome(tgt, kind == 0 ? (uint32_t)a2 : C.L + (uint32_t)a3, kind == 0 ? -20.0 : -50.0);
                if (ln.row_dirty) {
                    Row* wp = ln.row(tgt);
                    *reinterpret_cast<uint4*>(wp) = make_uint4((uint32_t)ln.props, (uint32_t)(ln.props >> 32), ln.ever, ln.since);
                    wp->misc = ln.misc;
                }
            }
            if (oob) { ln.raw = 0.0; ln.okind = MCBS_OUT_NONE; ln.olevel = 0; ln.new_nodes = 0; ln.new_creds = 0; }
            flags = (oob ? F_OOB : 0u) | ((uint32_t)ln.okind << F_KIND_SHIFT) | ((uint32_t)ln.olevel << F_LEVEL_SHIFT) |
                    ((uint32_t)ln.new_nodes << F_NEWNODES_SHIFT) | ((uint32_t)ln.new_creds << F_NEWCREDS_SHIFT);
        } else {
            oob = (flags & F_OOB) != 0;
            ln.raw = S.pending[e];
        }
        if (PHASE == 1) {
            S.pending[e] = ln.raw;
            S.h0[e] = make_uint4(step, flags, ln.n_disc | (ln.n_creds << 16), ln.owned | (ln.imaging << 16));
        } else {
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
        if (!need_reset) {
#pragma unroll
            for (int w = 0; w < NWT; ++w) {
                if ((uint32_t)w >= S.NW) break;
                if (ln.inst[w] != inst0[w]) S.m_inst[(size_t)w * S.E + e] = ln.inst[w];
                if (ln.run[w] != run0[w]) S.m_run[(size_t)w * S.E + e] = ln.run[w];
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

} // namespace mcbs
