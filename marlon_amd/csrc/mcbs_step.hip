// mcbs_step.hip — the hot path: CyberBattleEnv.step for a whole batch in one launch (gfx950).
//
// One LANE per environment (a wavefront advances 64 envs).  The transition of one env is a short,
// branchy chain of table look-ups with almost no intra-env parallelism (all
// integer / boolean work, O(1) per step once the per-env sets are bit masks), so spending a whole
// wavefront on one env would idle 63 lanes; wave-level cooperation is used where it exists — the
// coalesced re-initialisation of an env that just ended (ballot over the lanes that need a reset, then
// all 64 lanes copy the reset image) and the observation kernels (mcbs_obs.hip).
//
// At BASELINE.json's batch (65 536 envs = 1 024 wavefronts on 1 024 SIMDs) every wavefront is resident at
// once, so the launch takes as long as ONE wavefront's dependent chain of memory accesses (two thirds of a
// wavefront's cycles are parked at s_waitcnt: profiles/round2_notes.md).  The kernel is therefore organised by
// dependency level, not by reference function:
//   prologue: ONE batch of kernel-argument loads, then
//   level 1 (addresses depend on the env index only, issued back to back, all coalesced along the env axis):
//           header uint4, action row, first 16 discovery-order entries, first 16 credential-cache entries,
//           EVERY set of the env as u64 bit-mask words (discovered, agent installed, ever owned, running,
//           privilege bit-planes, gathered credentials, cached credential triples; packed batches: one uint4 and
//           every 4-byte node row), {cum_reward, availability}; the config words (action-space bounds, goal
//           constants) are fetched through the config pointer AFTER these are in flight;
//   level 2 (address depends on the action / header): the target node's 16-byte row, the re-imaging ring slot
//           of this defender tick (list entries beyond the first 16 for large topologies come first), and the
//           EIGHT loads from the topology's hot image — node record, source firewall word, authorisation word
//           (indexed by triple id), the 64-byte vulnerability descriptor with the first four leak entries inline —
//           through L1 / L2, fenced so that they go out together (leak entries 4..7: one more, wave-uniformly
//           skipped, batch);
//   ONE explicit wait for everything, in straight-line code; then pure register work and
//   one round of stores (row, sets, list appends, header, outputs) with NO load behind any store: on gfx9 vector
//   loads and stores retire in order on one counter, so a load behind a store waits for the write
//   acknowledgement, and a wait placed under divergent control flow can only be vmcnt(0).  "Rare" events are not
//   rare per wavefront (64 envs), so nothing on those paths may cost another trip to memory: a first version that
//   tested the ever-owned / discovered / credential sets lazily in memory spent 9 of its 10 dependency levels there
//   (profiles/round1_notes.md).  Round 1 staged the hot image in LDS per workgroup; reading it through L1 / L2
//   with one-wavefront workgroups measured faster at every BASELINE shape (TOPO_LDS = true survives behind
//   MCBS_LDS_TOPO=1 for the comparison).
//
// Control flow.  The lanes of a wavefront hold different action kinds, outcomes and validity, so every `if` of the
// reference that the compiler keeps as a branch is paid by the whole wave (compare, exec-mask save, branch, restore:
// four single-issue instructions at one wave per SIMD) whether or not any lane takes it.  The attacker's action is
// therefore written as straight-line predicated arithmetic: indices are clamped so that every look-up is safe for
// every lane, both the connect-side and the exploit-side tables are read, every check of the reference becomes a
// boolean, and the state changes are masked by the conjunction (`go`).  Loads and stores are unconditional wherever
// storing the unchanged value back is harmless.  What remains as real branches: the payload loop of leaked
// credentials / nodes, list entries past the first 16, the in-env defender, and the wave-level auto-reset.
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
#include "mcbs_ere.hip"
#include "mcbs_sample.hip"

namespace mcbs {

// ------------------------------ sets held in registers (WT words each) ------------------------------
// (word selection is written as mask arithmetic, not as a select chain: the compiler turns a chain of selects over m[0..WT)
// into a dynamically indexed stack array, i.e. scratch memory round trips, for WT == 4)
__device__ __forceinline__ uint64_t word_is(uint32_t n, int i) { return 0ull - (uint64_t)((n >> 6) == (uint32_t)i); }
template <int WT>
__device__ __forceinline__ bool rget(const uint64_t (&m)[WT], uint32_t n) {
    uint64_t w = WT == 1 ? m[0] : 0ull;
    if (WT > 1) {
#pragma unroll
        for (int i = 0; i < WT; ++i) w |= m[i] & word_is(n, i);
    }
    return (w >> (n & 63u)) & 1ull;
}
template <int WT>
__device__ __forceinline__ void rset(uint64_t (&m)[WT], uint32_t n) {
#pragma unroll
    for (int i = 0; i < WT; ++i) m[i] |= (1ull << (n & 63u)) & (WT == 1 ? ~0ull : word_is(n, i));
}
template <int WT>
__device__ __forceinline__ void rclear(uint64_t (&m)[WT], uint32_t n) {
#pragma unroll
    for (int i = 0; i < WT; ++i) m[i] &= ~((1ull << (n & 63u)) & (WT == 1 ? ~0ull : word_is(n, i)));
}
template <int WT>   // returns the previous value of the bit
__device__ __forceinline__ bool rtestset(uint64_t (&m)[WT], uint32_t n) {
    const bool was = rget<WT>(m, n);
    rset<WT>(m, n);
    return was;
}
// predicated forms for the straight-line attacker path: the bit of element n as a WT-word set, empty when !on
template <int WT>
__device__ __forceinline__ void rbit(uint64_t (&b)[WT], uint32_t n, bool on) {
    const uint64_t bit = on ? (1ull << (n & 63u)) : 0ull;
#pragma unroll
    for (int i = 0; i < WT; ++i) b[i] = bit & (WT == 1 ? ~0ull : word_is(n, i));
}

// entry i (< 16) of a 16-byte vector of u8, and of two 16-byte vectors of u16 (entries 0..7 | 8..15); written with 64-bit
// shifts so that the compiler keeps the vectors in registers (a select tree over the four dwords becomes a stack array)
__device__ __forceinline__ uint32_t byte_of(const uint4& v, uint32_t i) {
    const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    return (uint32_t)(((i & 8u) ? hi : lo) >> ((i & 7u) * 8u)) & 0xFFu;
}
__device__ __forceinline__ uint32_t half_of(const uint4& a, const uint4& b, uint32_t i) {
    const uint64_t a0 = (uint64_t)a.x | ((uint64_t)a.y << 32), a1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
    const uint64_t b0 = (uint64_t)b.x | ((uint64_t)b.y << 32), b1 = (uint64_t)b.z | ((uint64_t)b.w << 32);
    const uint64_t wa = (i & 4u) ? a1 : a0, wb = (i & 4u) ? b1 : b0;
    return (uint32_t)(((i & 8u) ? wb : wa) >> ((i & 3u) * 16u)) & 0xFFFFu;
}

// dword i (< 16) of four 16-byte vectors, as 64-bit selects and one shift (see byte_of)
__device__ __forceinline__ uint32_t dword_of(const uint4& r0, const uint4& r1, const uint4& r2, const uint4& r3, uint32_t i) {
#define MCBS_P64(a, b) ((uint64_t)(a) | ((uint64_t)(b) << 32))
    const uint64_t p0 = MCBS_P64(r0.x, r0.y), p1 = MCBS_P64(r0.z, r0.w), p2 = MCBS_P64(r1.x, r1.y), p3 = MCBS_P64(r1.z, r1.w);
    const uint64_t p4 = MCBS_P64(r2.x, r2.y), p5 = MCBS_P64(r2.z, r2.w), p6 = MCBS_P64(r3.x, r3.y), p7 = MCBS_P64(r3.z, r3.w);
#undef MCBS_P64
    const bool b1 = i & 2u, b2 = i & 4u, b3 = i & 8u;
    const uint64_t q0 = b1 ? p1 : p0, q1 = b1 ? p3 : p2, q2 = b1 ? p5 : p4, q3 = b1 ? p7 : p6;
    const uint64_t s0 = b2 ? q1 : q0, s1 = b2 ? q3 : q2;
    return (uint32_t)((b3 ? s1 : s0) >> ((i & 1u) * 32u));
}

// entry i (< 16) of a 16-byte vector of u8 / of two 16-byte vectors of u16 := v (64-bit mask arithmetic, see byte_of)
__device__ __forceinline__ void put_byte(uint4& v, uint32_t i, uint32_t x) {
    uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    const uint32_t sh = (i & 7u) * 8u;
    const uint64_t clr = ~(0xFFull << sh), val = (uint64_t)(x & 0xFFu) << sh;
    if (i & 8u) hi = (hi & clr) | val; else lo = (lo & clr) | val;
    v = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
__device__ __forceinline__ void put_half(uint4& a, uint4& b, uint32_t i, uint32_t x) {
    uint64_t w[4] = {(uint64_t)a.x | ((uint64_t)a.y << 32), (uint64_t)a.z | ((uint64_t)a.w << 32), (uint64_t)b.x | ((uint64_t)b.y << 32),
                     (uint64_t)b.z | ((uint64_t)b.w << 32)};
    const uint32_t sh = (i & 3u) * 16u, q = (i >> 2) & 3u;
    const uint64_t clr = ~(0xFFFFull << sh), val = (uint64_t)(x & 0xFFFFu) << sh;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) w[k] = k == q ? ((w[k] & clr) | val) : w[k];
    a = make_uint4((uint32_t)w[0], (uint32_t)(w[0] >> 32), (uint32_t)w[1], (uint32_t)(w[1] >> 32));
    b = make_uint4((uint32_t)w[2], (uint32_t)(w[2] >> 32), (uint32_t)w[3], (uint32_t)(w[3] >> 32));
}

// Hook points of step_body for kernels that wrap more work around the same step in the same launch (mcbs_wrapper_fused.hip); every
// instantiation below that does not name a hook uses this empty one and compiles to exactly the code it had without them.
struct NoHook {
    static constexpr bool kAction = false;   // the action row is produced by the hook (decode of a policy action) instead of read from io.actions
    static constexpr bool kObs = false;      // the hook assembles the observation between the attacker's action and the defender's turn
    static constexpr bool kFinish = false;   // the hook runs after the step's stores (wrapper bookkeeping, auto-reset, streaming the observation)
};

// ------------------------------ per-lane working set ------------------------------
template <int WT>
struct Lane {
    const DevState& S;
    const StepCfg& C;
    const uint8_t* tb;   // hot image (LDS copy, or Topo::hot in global memory)
    uint32_t e;
    uint8_t* body;
    uint32_t n_disc, n_creds, owned, dclk;
    uint64_t m[M_COUNT][WT];
    uint32_t dirty;      // defender paths: bit k = set k changed and must be written back
    // the target node's row, in registers
    uint64_t props;      // discovered properties (60 bits)
    uint32_t ever, since, tags;
    bool row_dirty;      // (unused by the step kernel: the row is always written back)
    // learned-defender tier: this env's firewall state of source / target (12 bits per rule list, see mcbs_defend.hip), else unused
    bool learned;
    uint32_t fw_src, fw_tgt;
    // result of the attacker's action
    double raw;
    int okind, olevel, new_nodes, new_creds;
    // wide cached-triple set (DevState::cach): this lane's LDS column, [word * wide_stride], staged around a credential leak
    uint64_t* wide_lds = nullptr;
    uint32_t wide_stride = 0;
    // packed batches, whole step: the env's 4-byte node rows as the step loaded them and the target's row as the attacker left it, so that
    // re-imaging a node is a pure STORE (a read-modify-write would be a load behind the step's stores: a write-acknowledgement round trip)
    bool rows_in_regs = false;
    uint4 prw0 = {0, 0, 0, 0}, prw1 = {0, 0, 0, 0}, prw2 = {0, 0, 0, 0}, prw3 = {0, 0, 0, 0};
    uint32_t ptgt = 0xFFFFFFFFu, pword = 0;
    const uint8_t* ere_blob = nullptr;   // the topology blob (ExternalRandomEvents reads its cold tables)
    // fused wrapper step (mcbs_wrapper_fused.hip, act<.., REC = true>): the heads of the two lists kept up to date in registers while the
    // leak entries are appended, so that the observation can be assembled without reading the lists back
    uint4 rec_dh = {0, 0, 0, 0}, rec_c0 = {0, 0, 0, 0}, rec_c1 = {0, 0, 0, 0};

    __device__ __forceinline__ const HotNode* NS(uint32_t n) const { return reinterpret_cast<const HotNode*>(tb + C.hot_node) + n; }
    __device__ __forceinline__ Row* row(uint32_t n) const { return reinterpret_cast<Row*>(body + S.off_rows) + n; }
    __device__ __forceinline__ uint8_t* disc_list() const { return body + S.off_disc; }
    __device__ __forceinline__ uint16_t* cred_list() const { return reinterpret_cast<uint16_t*>(body + S.off_cred); }

    __device__ __forceinline__ uint32_t privilege(uint32_t n) const {
        return (uint32_t)rget<WT>(m[M_PLO], n) | ((uint32_t)rget<WT>(m[M_PHI], n) << 1);
    }
    __device__ __forceinline__ void set_privilege(uint32_t n, uint32_t p) {
        if (p & 1u) rset<WT>(m[M_PLO], n); else rclear<WT>(m[M_PLO], n);
        if (p & 2u) rset<WT>(m[M_PHI], n); else rclear<WT>(m[M_PHI], n);
        dirty |= (1u << M_PLO) | (1u << M_PHI);
    }

    // One action of one env: AgentActions.exploit_local_vulnerability / exploit_remote_vulnerability (actions.py:473-502,
    // 425-471) -> __process_outcome (325-423) with __mark_discovered_entities (277-310), and connect_to_remote_machine
    // (524-606), as ONE predicated flow (see "Control flow" above).
    //   X      : the env executes this action (live env, indices in range, credential index inside the cache)
    //   raw_nx : the raw reward when !X (0 out of bounds, -1 credential index outside the cache; env.py:736-737)
    //   kind   : 0 local, 1 remote, 2 connect; `col` = vulnerability column (exploits; 0 for connect),
    //            `port` / `triple` = connect arguments (0 for exploits).  Every index is valid for every lane.
    // WIDE_OK = false: the batch cannot have a wide cached-triple set (packed layout); compiles its handling out.
    // DK: the batch's defender kind; MCBS_DEFENDER_RANDOM_EVENTS consults the env's own vulnerability / service / firewall state
    template <bool WIDE_OK, int DK, bool REC = false>
    __device__ __forceinline__ void act(bool X, double raw_nx, int kind, uint32_t src, uint32_t tgt, uint32_t col, uint32_t port, uint32_t triple) {
        const bool k2 = kind == 2;
        // ---- look-ups of both flavours (LDS) ----
        const uint4* tp = reinterpret_cast<const uint4*>(NS(tgt));
        const uint4 t0 = tp[0], t1 = tp[1];              // {props lo, hi, value, fw_in_allow} {fw_out_allow, listen, svc, flags}
        const uint64_t t_props = (uint64_t)t0.x | ((uint64_t)t0.y << 32);
        const int t_value = (int)t0.z;
        const uint32_t src_fw_out = NS(src)->fw_out_allow;
        const uint64_t auth = reinterpret_cast<const uint64_t*>(tb + C.hot_auth)[(tgt * C.P + port) * C.auth_words + (triple >> 6)];
        const uint4* dp = reinterpret_cast<const uint4*>(tb + C.hot_desc + (tgt * (C.L + C.R) + col) * (uint32_t)sizeof(HotDesc));
        const uint4 d0 = dp[0], d1 = dp[1];   // {cost lo,hi, probe lo,hi} {payload_off, cnt | tt << 16, kind | level << 8 | slot << 16, -}
        const uint4 d2 = dp[2], d3 = dp[3];   // the first four payload entries {node | cred << 16, triple | port << 16} x 2, x 2
        __builtin_amdgcn_sched_barrier(0);    // all eight table loads go out together: left alone, the scheduler sinks the descriptor's
                                              // four below the first uses of the node record, i.e. behind a wait — one more round trip

        const bool src_owned = rget<WT>(m[M_INST], src);
        const bool running = rget<WT>(m[M_RUN], tgt);
        const bool already = rget<WT>(m[M_INST], tgt);

        // ---- connect_to_remote_machine checks, in the reference's order: firewalls (-10), listening (-10), running (0),
        // credentials (-10: WRONG_PASSWORD); target discovered / credential gathered hold by construction (own lists) ----
        bool out_ok = (src_fw_out >> port) & 1u, in_ok = (t0.w >> port) & 1u;
        if (learned) {                                       // a manageable rule name: the env's own rule state decides
#pragma unroll
            for (uint32_t r = 0; r < 6u; ++r) {
                const bool mine = C.rule_port[r] == port;
                out_ok = mine ? (((fw_src >> r) & (fw_src >> (6u + r)) & 1u) != 0) : out_ok;   // fw_src: the source's OUTGOING list
                in_ok = mine ? (((fw_tgt >> r) & (fw_tgt >> (6u + r)) & 1u) != 0) : in_ok;     // fw_tgt: the target's INCOMING list
            }
        }
        bool authorized = (auth >> (triple & 63u)) & 1ull;                                      // actions.py:608-621, precomputed per (node, port, cached credential)
        uint64_t own_present = ~0ull;
        if (DK == MCBS_DEFENDER_RANDOM_EVENTS) {               // (slow path by design: loops over the env's own tables in memory)
            const EreView V{body, C, ere_blob, S.N};
            own_present = V.present(tgt) | C.ere_lib_cols;
            if (X & k2) {
                const uint32_t* lists = reinterpret_cast<const uint32_t*>(tb + C.hot_fwlist);
                out_ok = V.passes(lists[src] >> 16, port);
                in_ok = V.passes(lists[tgt] & 0xFFFFu, port);
                authorized = V.authorized(tgt, port, (reinterpret_cast<const mcbs_triple*>(tb + C.hot_triple) + triple)->cred);
            }
        }
        const bool reach = out_ok & in_ok & (bool)((t1.y >> port) & 1u);                        // not BLOCKED_BY_*_FIREWALL, not SCANNING_UNOPEN_PORT
        const bool c_proceed = reach & running & authorized;
        const double c_fail_raw = (reach & !running) ? 0.0 : -10.0;

        // ---- exploit checks: MACHINE_NOT_RUNNING 0 > SUPSPICIOUSNESS -5 > LOCAL_EXPLOIT_FAILED -20 / FAILED_REMOTE_EXPLOIT -50 > REPEAT -1 ----
        const uint32_t vk = d1.z & 0xFFu, level = (d1.z >> 8) & 0xFFu;
        const bool present = (vk != 0xFFu) & (bool)((own_present >> (col & 63u)) & 1ull);       // (random events: the key may have been patched away)
        const bool pre_ok = ((d1.y >> 16) >> tags) & 1u;                                        // precondition on (static props, tags)
        const bool esc = !k2 & (vk == MCBS_OUT_PRIVILEGE_ESCALATION);
        const bool repeat_esc = esc & (bool)((tags >> level) & 1u);                             // tag already on the node
        const bool x_proceed = running & present & pre_ok & !repeat_esc;
        const double x_fail_raw = !running ? 0.0 : (!present ? -5.0 : (!pre_ok ? (kind == 0 ? -20.0 : -50.0) : -1.0));
        const int x_fail_kind = (running & present) ? (!pre_ok ? MCBS_OUT_EXPLOIT_FAILED : MCBS_OUT_PRIVILEGE_ESCALATION) : MCBS_OUT_NONE;
        const int x_lvl = (running & present & pre_ok & esc) ? (int)level : 0;

        const bool so = X & src_owned;                          // else INVALID_ACTION (-1), checked first
        const bool go = so & (k2 ? c_proceed : x_proceed);      // the action takes effect
        const bool xb = go & !k2;                               // ... and is an exploit
        const uint32_t vk_ok = k2 ? (uint32_t)MCBS_OUT_LATERAL_MOVE : vk;
        const uint32_t own_level = esc ? level : 1u;

        // ---- __mark_node_as_owned (actions.py:251-275): `already` = currently owned (agent installed); a node that is
        // owned now was owned before, so only a change of ownership consults / sets the ever-owned bit ----
        const bool newly = go & !already & (k2 | esc | (vk == MCBS_OUT_LATERAL_MOVE));
        const bool first_time = newly & !rget<WT>(m[M_EVER], tgt);
        const uint32_t priv = privilege(tgt);
        const uint32_t np = priv > own_level ? priv : own_level;   // model.escalate
        const bool chg = newly & (np != priv);
        uint64_t bn[WT], bc[WT];
        rbit<WT>(bn, tgt, newly);
        rbit<WT>(bc, tgt, chg);
#pragma unroll
        for (int w = 0; w < WT; ++w) {
            m[M_EVER][w] |= bn[w];
            m[M_INST][w] |= bn[w];
            m[M_PLO][w] = (m[M_PLO][w] & ~bc[w]) | ((np & 1u) ? bc[w] : 0ull);
            m[M_PHI][w] = (m[M_PHI][w] & ~bc[w]) | ((np & 2u) ? bc[w] : 0ull);
        }
        owned += (chg & (priv == 0u)) ? 1u : 0u;
        props |= newly ? t_props : 0ull;                        // all (non-tag) properties become known
        tags |= (xb & esc) ? (1u << own_level) : 0u;

        // ---- exploit bookkeeping (actions.py:386-423) ----
        int r = first_time ? t_value : 0;                       // (connect uses it below)
        const uint64_t probe = (xb & (vk == MCBS_OUT_PROBE_SUCCEEDED)) ? ((uint64_t)d0.z | ((uint64_t)d0.w << 32)) : 0ull;
        int rx = r + 2 * __popcll(probe & ~props);
        props |= probe;
        const uint32_t sb = xb ? (1u << ((d1.z >> 16) & 0x1Fu)) : 0u;
        rx += (ever & sb) ? ((since & sb) ? -1 : 0) : (xb ? 7 : 0);
        ever |= sb; since |= sb;
        const bool creds = vk == MCBS_OUT_LEAKED_CREDENTIALS;
        const uint32_t cnt = (xb & (creds | (vk == MCBS_OUT_LEAKED_NODES))) ? (d1.y & 0xFFFFu) : 0u;
        const uint2* pl = reinterpret_cast<const uint2*>(tb + C.hot_payload) + d1.x;
        uint32_t nn = 0, nc = 0, ncache = 0;
        const bool wide = WIDE_OK && S.wide != 0u;       // uniform: the cached-triple set lives in memory, staged in LDS for the loop
        const bool stage = wide && cnt != 0u && creds;
        if (stage)
            for (uint32_t w = 0; w < S.TW; ++w) wide_lds[w * wide_stride] = S.cach[(size_t)w * S.E + e];
        // The first four entries came with the descriptor; entries 4..7 are fetched now, before the first store (a load behind a store
        // waits for the store's write acknowledgement), wave-uniformly skipped when no lane has that many; longer lists: the tail loop.
        constexpr uint32_t PF = 8;
        uint2 pre[PF] = {make_uint2(d2.x, d2.y), make_uint2(d2.z, d2.w), make_uint2(d3.x, d3.y), make_uint2(d3.z, d3.w),
                         make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u)};
#pragma unroll
        for (uint32_t i = 4; i < PF; ++i)
            if (__ballot(i < cnt)) pre[i] = pl[i < cnt ? i : 0u];
        // Every load of the step has been issued by now and the stores start below.  Wait for ALL of them here, in straight-line code:
        // the leak entries run under divergent control flow, and a wait the compiler has to place INSIDE it (it cannot prove that an
        // entry's operands have landed) is `s_waitcnt vmcnt(0)` with the previous entry's stores in flight — a write-acknowledgement
        // round trip per leaked entry (the vector memory counter retires loads and stores in order).
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0), expcnt / lgkmcnt untouched
        auto leak = [&](const uint2 p) {                 // one LeakedCredentials / LeakedNodesId entry {node | cred << 16, triple | port << 16}
            const uint32_t pn = p.x & 0xFFFFu, pc = p.x >> 16, pt = p.y & 0xFFFFu;
            // appends go to the slot past the list's end whether or not the element is new (the lists have one slack slot):
            // the count only advances for a new element, so a stale write is overwritten or never read
            const bool new_n = !rget<WT>(m[M_DISC], pn);
            const bool new_g = creds & !rget<WT>(m[M_GATH], pc);
            bool new_c;
            if (wide) {
                uint64_t* w = wide_lds + (pt >> 6) * wide_stride;
                const uint64_t old = stage ? *w : 0ull, bit = 1ull << (pt & 63u);
                new_c = creds & !(old & bit);
                if (stage) *w = old | bit;
            } else new_c = creds & !rget<WT>(m[M_CACH], pt);
            disc_list()[n_disc] = (uint8_t)pn;
            cred_list()[n_creds] = (uint16_t)pt;
            if (REC) {                                   // (packed batches: both lists have fewer than 16 entries)
                put_byte(rec_dh, n_disc & 15u, pn);
                put_half(rec_c0, rec_c1, n_creds & 15u, pt);
            }
            uint64_t b0[WT], b1[WT], b2[WT];
            rbit<WT>(b0, pn, new_n); rbit<WT>(b1, pc, new_g); rbit<WT>(b2, pt & (WT * 64u - 1u), new_c && !wide);
#pragma unroll
            for (int w = 0; w < WT; ++w) { m[M_DISC][w] |= b0[w]; m[M_GATH][w] |= b1[w]; m[M_CACH][w] |= b2[w]; }
            n_disc += new_n; nn += new_n; nc += new_g; n_creds += new_c; ncache += new_c;
        };
#pragma unroll
        for (uint32_t i = 0; i < PF; ++i) {
            if (!__ballot(i < cnt)) break;
            if (i < cnt) leak(pre[i]);
        }
        for (uint32_t i = PF; i < cnt; ++i) leak(pl[i]);
        if (stage)
            for (uint32_t w = 0; w < S.TW; ++w) S.cach[(size_t)w * S.E + e] = wide_lds[w * wide_stride];
        rx += 5 * (int)nn + 3 * (int)nc;
        const double x_raw = (double)rx - __hiloint2double((int)d0.y, (int)d0.x);
        const double c_raw = already ? -1.0 : (double)r;                                       // REPEAT, else the node's value the first time

        raw = !X ? raw_nx : (!src_owned ? -1.0 : (!go ? (k2 ? c_fail_raw : x_fail_raw) : (k2 ? c_raw : x_raw)));
        okind = go ? (int)vk_ok : ((so & !k2) ? x_fail_kind : MCBS_OUT_NONE);
        olevel = (so & !k2) ? x_lvl : 0;
        new_nodes = (int)nn;
        new_creds = (int)ncache;
    }

    // ---- defender ----
    // One Philox4x32-10 block yields the two doubles 2b and 2b+1 of a step; the block last computed is kept, so consecutive draws
    // (scan draws 0..k-1, then the detection draws) cost one block per PAIR
    uint32_t rng_block = 0xFFFFFFFFu, rng_w[4] = {0u, 0u, 0u, 0u};
    __device__ __forceinline__ double draw(uint32_t i, uint32_t step, uint32_t episode, const StepIO& io) {
        if (C.rng_kind == MCBS_RNG_TAPE) return (io.tape && i < io.tape_dps) ? io.tape[(size_t)e * io.tape_dps + i] : 0.0;
        if ((i >> 1) != rng_block) {
            const uint64_t gid = C.env_id_base + e;
            philox4x32_10((uint32_t)gid, episode, step, i >> 1, (uint32_t)C.seed, (uint32_t)(C.seed >> 32) ^ (uint32_t)(gid >> 32), rng_w);
            rng_block = i >> 1;
        }
        return (i & 1u) ? to_double53(rng_w[2], rng_w[3]) : to_double53(rng_w[0], rng_w[1]);
    }

    __device__ __forceinline__ uint64_t valid_bits(uint32_t w) const {
        if (w * 64u >= S.N) return 0ull;
        const uint32_t rem = S.N - w * 64u;
        return rem >= 64u ? ~0ull : ((1ull << rem) - 1ull);
    }
    __device__ __forceinline__ double avail_term(uint32_t n) const { return reinterpret_cast<const double*>(tb + C.hot_avail)[n]; }

    // on_attacker_step_taken (actions.py:714-746): nodes whose re-imaging started 16 defender ticks ago are back
    // (REIMAGING_DURATION 15 -> 0, then Running); `back` = this tick's ring slot.  Returns the availability.
    __device__ __forceinline__ double defender_tick(const uint64_t (&back)[WT]) {
        uint32_t imaging = 0;
#pragma unroll
        for (int w = 0; w < WT; ++w) {
            if (back[w]) { m[M_RUN][w] |= back[w]; dirty |= 1u << M_RUN; }
            imaging += __popcll(~m[M_RUN][w] & valid_bits(w));
        }
        if (!imaging) return C.full_availability;
        double s;
        if (C.avail_uniform) {            // every node has the same term and sums are exact in any order (all reference samples: 1.0): no table read
            s = C.full_sum - (double)imaging * C.avail_term0;
        } else if (C.avail_any_order) {   // exact in any order: subtract the terms of the nodes being re-imaged
            s = C.full_sum;
#pragma unroll
            for (int w = 0; w < WT; ++w) {
                uint64_t im = ~m[M_RUN][w] & valid_bits(w);
                while (im) { const uint32_t b = (uint32_t)__builtin_ctzll(im); im &= im - 1; s -= avail_term(w * 64u + b); }
            }
        } else {                          // the reference's node-order sum
            s = 0.0;
            for (uint32_t n = 0; n < S.N; ++n) if (rget<WT>(m[M_RUN], n)) s += avail_term(n);
        }
        return s / C.total_sla_weight;
    }

    // ScanAndReimageCompromisedMachines.step (defender.py:42-55) + reimage_node (actions.py:700-712).
    // Nodes re-imaged now are collected in `fresh`: they come back 16 ticks from now (same ring slot).
    __device__ __forceinline__ void defender_scan(uint32_t step, uint32_t episode, const StepIO& io, uint64_t (&fresh)[WT]) {
        if (step % C.scan_frequency) return;
        uint32_t det = 0;
        uint64_t reim[WT];                // re-imagable nodes as set words from the config (scalar loads), not a table read behind the step's stores
#pragma unroll
        for (int w = 0; w < WT; ++w) reim[w] = C.reimagable[w];
        for (uint32_t i = 0; i < C.scan_capacity; ++i) {
            int n = (int)floor(draw(i, step, episode, io) * (double)S.N);
            if (n >= (int)S.N) n = (int)S.N - 1;
            if (!rget<WT>(m[M_RUN], (uint32_t)n) || !rget<WT>(m[M_INST], (uint32_t)n)) continue;
            const double d = draw(C.scan_capacity + det, step, episode, io);
            det += 1;
            if (!(d <= C.scan_probability) || !rget<WT>(reim, (uint32_t)n)) continue;
            reimage((uint32_t)n, fresh);
        }
    }

    // reimage_node (actions.py:700-712): agent removed, privilege NoAccess, Imaging for REIMAGING_DURATION ticks; tags,
    // discovered properties and credentials stay (quirk Q6)
    __device__ __forceinline__ void reimage(uint32_t n, uint64_t (&fresh)[WT]) {
        if (S.packed) {                                       // every earlier attack now predates last_reimaging: attacked-since := 0
            const uint32_t keep = (1u << (S.tiny_p + 4u + S.tiny_v)) - 1u;
            uint32_t* rw = reinterpret_cast<uint32_t*>(body + S.off_rows) + n;
            if (rows_in_regs) *rw = (n == ptgt ? pword : dword_of(prw0, prw1, prw2, prw3, n & 15u)) & keep;
            else *rw &= keep;
        } else row(n)->since = 0;
        rclear<WT>(m[M_INST], n);
        if (privilege(n)) { set_privilege(n, 0u); owned -= 1; }
        rclear<WT>(m[M_RUN], n);
        rset<WT>(fresh, n);
        dirty |= (1u << M_INST) | (1u << M_RUN);
    }
};

// PHASE 0: whole step.  PHASE 1: attacker's action only (raw reward parked in S.pending).
// PHASE 2: defender, goals, outputs, auto-reset (after the observation kernels ran).
// WT: words per set held in registers (1, 2 or 4; >= NW, SW, TW).  TOPO_LDS: topology tables staged in LDS.
// DEFK: MCBS_DEFENDER_* (none / in-env ScanAndReimage / external learned defender).
// MANY: the in-kernel step loop of mcbs_step_many / mcbs_rollout_random (step_many_kernel below); `roll` = the random agent of
// mcbs_rollout_random (mode 0: actions are read from io.actions).
template <int PHASE, int WTP, bool TOPO_LDS, int DEFK, bool MANY, class Hook>
__device__ __forceinline__ void step_body(const DevState& S, const Topo& T, const StepCfg* __restrict__ Cp, const StepIO& io, const RollArgs& roll, Hook& hook) {
    constexpr bool PK = WTP == 0;           // packed batch: the eight sets are 16-bit fields of one uint4 per env
    constexpr int WT = PK ? 1 : WTP;
    const StepCfg& C = *Cp;   // in device memory: fields are fetched by scalar loads where they are used, not all up front
    extern __shared__ uint4 topo_lds[];
#ifdef MCBS_DIAG
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st_[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_NOWAIT(i) do { __builtin_amdgcn_sched_barrier(0); st_[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i)
#define STAMP_NOWAIT(i)
#endif
    STAMP_NOWAIT(0);
    // action-space bounds as scalars, fetched now: left alone, the compiler turns `k1 ? C.R : C.P` into a per-lane VECTOR load
    // from the config (select of two loads -> load of the selected address), i.e. one more memory round trip after level 1
    // (fetched and pinned AFTER the level-1 vector loads are issued, below: pinned here, the kernel waited for the kernel-argument
    // load, then for this load through the config pointer, before its first vector load went out — a quarter of a wavefront's life)
    // Workgroups of the L1 / L2 variant are always one wavefront (mcbs_api.hip): the env index needs no hidden-argument load
    const uint32_t bdim = TOPO_LDS ? blockDim.x : 64u;
    const uint32_t e = blockIdx.x * bdim + threadIdx.x;
    const bool active = e < S.E;
    const uint32_t ec = active ? e : 0u;                // clamp so inactive lanes read valid memory and take no branch
    constexpr bool has_def = DEFK == MCBS_DEFENDER_SCAN_AND_REIMAGE;   // in-env defender, resolved at launch
    constexpr bool learned = DEFK == MCBS_DEFENDER_EXTERNAL;           // firewall rules are per-env state (learned defender)
    constexpr bool ere = DEFK == MCBS_DEFENDER_RANDOM_EVENTS;          // ExternalRandomEvents: per-env vulnerability / service / firewall state
    constexpr bool def_avail = has_def || ere;                         // a defender agent exists: availability goals and the SLA constraint apply

    // MANY: mcbs_step_many — n_steps consecutive steps of every env in ONE launch (action batches [n_steps, E, 5], outputs
    // [n_steps, E]); each iteration is the whole step below against the state the previous one stored (a lane reads its own
    // stores in program order; the reset copy is by lanes of the same wavefront).  Envs never interact, so no grid-wide
    // synchronisation is involved; what goes away is the per-launch cost between dependent steps.
    const uint32_t n_it = MANY ? io.n_steps : 1u;
    const uint8_t* tb = TOPO_LDS ? reinterpret_cast<const uint8_t*>(topo_lds) : T.hot;
    for (uint32_t it = 0; it < n_it; ++it) {
    StepIO iok = io;
    if (MANY) {
        iok.actions = io.actions ? io.actions + (size_t)it * S.E * 5u : nullptr;
        iok.reward = io.reward + (size_t)it * S.E;
        iok.terminated = io.terminated + (size_t)it * S.E;
    }
    // ---------------- level 1: loads whose addresses depend on the env index only ----------------
    uint8_t* body = S.body + (size_t)ec * S.body_stride;
    const uint4 h0 = S.h0[ec];
    uint4 a03 = make_uint4(0, 0, 0, 0);
    uint32_t a4 = 0;
    uint4 dhead = make_uint4(0, 0, 0, 0), chead0 = dhead, chead1 = dhead;
    if (MANY && roll.mode) {                            // on-device random agent: this step's action comes from the env's own state
        int32_t ra[5];
        sample_action(S, T, C, ec, roll.mode == 2u, roll.seed, roll.step0 + it, roll.nmax, roll.cmax, ra);
        a03 = make_uint4((uint32_t)ra[0], (uint32_t)ra[1], (uint32_t)ra[2], (uint32_t)ra[3]);
        a4 = (uint32_t)ra[4];
        if (io.actions && active) {
            int32_t* o = const_cast<int32_t*>(iok.actions) + (size_t)e * 5;
            o[0] = ra[0]; o[1] = ra[1]; o[2] = ra[2]; o[3] = ra[3]; o[4] = ra[4];
        }
    } else if constexpr (Hook::kAction) {
        hook.load_action(ec);                            // the policy's own encoding; decoded below, once the header has landed
    } else if (PHASE != 2) {
        const uint32_t* ap = reinterpret_cast<const uint32_t*>(iok.actions) + (size_t)ec * 5;
        a03 = make_uint4(ap[0], ap[1], ap[2], ap[3]);
        a4 = ap[4];
    }
    if (PHASE != 2) {
        dhead = *reinterpret_cast<const uint4*>(body + S.off_disc);
        chead0 = *reinterpret_cast<const uint4*>(body + S.off_cred);
        chead1 = *reinterpret_cast<const uint4*>(body + S.off_cred + 16);
    }
    uint4 rw0 = make_uint4(0, 0, 0, 0), rw1 = rw0, rw2 = rw0, rw3 = rw0;
    if (PK && PHASE != 2) {                              // packed batch: every 4-byte node row of the env (<= 64 bytes)
        const uint4* rp = reinterpret_cast<const uint4*>(body + S.off_rows);
        rw0 = rp[0]; rw1 = rp[1]; rw2 = rp[2];
        if (S.N > 12u) rw3 = rp[3];
    }
    uint64_t m0[M_COUNT][WT];        // every set is stored padded to WT words: no bounds to test, all loads independent
    if (PK) {
        const uint4 pk = reinterpret_cast<const uint4*>(S.masks)[ec];
        const uint32_t f[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
        for (int k = 0; k < M_COUNT; ++k) m0[k][0] = (f[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
    } else {
#pragma unroll
        for (int k = 0; k < M_COUNT; ++k) {
            const bool wanted = PHASE != 2 || (k != M_GATH && k != M_CACH && k != M_DISC && k != M_EVER);
#pragma unroll
            for (int w = 0; w < WT; ++w) m0[k][w] = wanted ? S.masks[((uint32_t)k * WT + (uint32_t)w) * S.E + ec] : 0ull;
        }
    }
    double2 h1 = make_double2(0.0, 0.0);
    uint32_t episode = 0;
    double pending = 0.0;
    if (PHASE != 1) {
        h1 = S.h1[ec];
        // (packed batches: always — the auto-reset of an env that ends needs episode + 1, and a load in the reset tail would sit behind
        // the step's stores)
        if (PK || (def_avail && C.rng_kind == MCBS_RNG_PHILOX)) episode = S.episode[ec];
    }
    if (PHASE == 2) pending = S.pending[ec];

    if constexpr (Hook::kFinish) hook.level1(ec);                  // its own level-1 loads (wrapper counters ...) go out behind the step's
    STAMP_NOWAIT(1);   // level-1 loads issued
    uint32_t cL = C.L, cR = C.R, cP = C.P;
    // the goal / termination constants too (one pin for all, so that the loads go out together): fetched where they are used they
    // were five serial scalar-load waits on the way to the step's stores; here their latency disappears behind the vector loads'
    unsigned long long g_reward = __double_as_longlong(C.goal_reward), g_low = __double_as_longlong(C.goal_low_availability),
                       g_sla = __double_as_longlong(C.maintain_sla),
                       g_win = __double_as_longlong(C.winning_reward), g_lose = __double_as_longlong(C.losing_reward);
    uint32_t g_has = C.has_attacker_goal, g_own = C.goal_own_atleast, g_evict = C.defender_goal_eviction, g_auto = C.auto_reset, g_max = C.max_episode_steps,
             g_pctmin = C.goal_own_pct_min;
    if (PHASE != 1)
        asm volatile("" : "+s"(cL), "+s"(cR), "+s"(cP), "+s"(g_reward), "+s"(g_low), "+s"(g_pctmin), "+s"(g_sla), "+s"(g_win), "+s"(g_lose), "+s"(g_has),
                          "+s"(g_own), "+s"(g_evict), "+s"(g_auto), "+s"(g_max));
    else asm volatile("" : "+s"(cL), "+s"(cR), "+s"(cP));
    if (!TOPO_LDS) tb = T.hot;
    if (TOPO_LDS && it == 0u) {                         // cooperative copy of the hot topology image, 16 bytes per lane
        const uint4* src = reinterpret_cast<const uint4*>(T.hot);
        const uint32_t nvec = C.hot_bytes / 16u, bd = bdim;
        if (PK || nvec <= 2u * bd) {                    // small image (Chain-10: 1.1 passes): the plain loop is the fastest here; packed
                                                        // batches (<= 16 nodes) never have a large one, and their kernel keeps exactly this code
            for (uint32_t i = threadIdx.x; i < nvec; i += bd) topo_lds[i] = src[i];
        } else {                                        // large image (Chain-100: 8 passes): four loads in flight per lane, not one —
            for (uint32_t i = threadIdx.x; i < nvec; i += 4u * bd) {   // a plain loop waits for each pass before issuing the next
                const uint32_t i1 = i + bd, i2 = i + 2u * bd, i3 = i + 3u * bd, last = nvec - 1u;
                const uint4 v0 = src[i], v1 = src[i1 < nvec ? i1 : last], v2 = src[i2 < nvec ? i2 : last], v3 = src[i3 < nvec ? i3 : last];
                topo_lds[i] = v0;
                if (i1 < nvec) topo_lds[i1] = v1;
                if (i2 < nvec) topo_lds[i2] = v2;
                if (i3 < nvec) topo_lds[i3] = v3;
            }
        }
        __syncthreads();
    }

    STAMP(2);          // level-1 loads and the LDS copy have landed
    bool need_reset = false;
    float hk_reward = 0.0f;
    bool hk_done = false;
    if (active) {   // (inactive lanes of the last wavefront still take part in the wave-level reset copy below)
    const uint32_t old_flags = h0.y;
    if constexpr (Hook::kAction) hook.decode(S, C, h0, a03, a4);
    const bool ended = (old_flags & (F_DONE | F_TRUNC)) != 0;   // step after done: the reference raises RuntimeError
                                                                // (env.py:1146-1147); the batch leaves the env untouched
    // skip actions (MCBS_ACTION_SKIP) leave the env untouched too; a split step remembers them in F_SKIP for phase 2
    const bool skip_env = PHASE == 2 ? (old_flags & F_SKIP) != 0 : (int)a03.x == MCBS_ACTION_SKIP;
    const bool live = !ended & !skip_env;

    Lane<WT> ln{S, C, tb, ec, body, h0.z & 0xFFFFu, h0.z >> 16, h0.w & 0xFFFFu, h0.w >> 16, {}, 0u, 0ull, 0u, 0u, 0u, false, learned, 0u, 0u,
                0.0, MCBS_OUT_NONE, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < M_COUNT; ++k)
#pragma unroll
        for (int w = 0; w < WT; ++w) ln.m[k][w] = m0[k][w];
    ln.ere_blob = T.base;
    if (Hook::kObs) { ln.rec_dh = dhead; ln.rec_c0 = chead0; ln.rec_c1 = chead1; }
    if (!PK && S.wide) {                                // this lane's LDS column for the wide cached-triple set, behind the hot image
        ln.wide_lds = reinterpret_cast<uint64_t*>(topo_lds + (TOPO_LDS ? C.hot_bytes / 16u : 0u)) + threadIdx.x;
        ln.wide_stride = bdim;
    }
    // level 2 (needs the header): this defender tick's ring slot
    uint64_t back[WT];
#pragma unroll
    for (int w = 0; w < WT; ++w)
        back[w] = (PHASE != 1 && has_def) ? S.ring[((ln.dclk & 15u) * WT + (uint32_t)w) * S.E + ec] : 0ull;

    uint32_t step = h0.x, flags = old_flags;
    bool oob = false;
    if (PHASE != 2) {
        // ---------------- __execute_action (cyberbattle_env.py:707-751), index checks as booleans ----------------
        const int kind = (int)a03.x;
        const uint32_t a1 = a03.y, a2 = a03.z, a3 = a03.w;      // unsigned compares reject negative indices as well
        const bool k0 = kind == 0, k1 = kind == 1, k2 = kind == 2;
        const bool skip = k2 & (a4 >= ln.n_creds);              // connect with a credential index outside the cache: env.py:736-737,
                                                                // before any node look-up
        const bool bad = (a1 >= ln.n_disc) | (a2 >= (k0 ? cL : ln.n_disc)) | (!k0 & (a3 >= (k1 ? cR : cP)));
        oob = live & (!(k0 | k1 | k2) | (!skip & bad));
        const bool X = live & !skip & !oob & (k0 | k1 | k2);
        // indices every lane may use: its own when the action executes, entry 0 otherwise
        const uint32_t i1 = X ? a1 : 0u, i2 = (X & !k0) ? a2 : i1, i4 = (X & k2) ? a4 : 0u;
        uint32_t src = byte_of(dhead, i1 & 15u), tgt = byte_of(dhead, i2 & 15u);
        uint32_t triple = half_of(chead0, chead1, i4 & 15u);
        if (!PK && (i1 | i2 | i4) >= 16u) {                     // large topologies: entries past the first 16 of a list
            const uint32_t s2 = ln.disc_list()[i1], t2 = ln.disc_list()[i2], c2 = ln.cred_list()[i4];
            src = i1 >= 16u ? s2 : src; tgt = i2 >= 16u ? t2 : tgt; triple = i4 >= 16u ? c2 : triple;
        }
        // ---------------- level 2: the target row (packed batches: already here) ----------------
        uint4 r0;
        if (PK) {
            const uint32_t w = dword_of(rw0, rw1, rw2, rw3, tgt), vm = (1u << S.tiny_v) - 1u;
            r0 = make_uint4(w & ((1u << S.tiny_p) - 1u), ((w >> S.tiny_p) & 0xFu) << 28, (w >> (S.tiny_p + 4u)) & vm,
                            (w >> (S.tiny_p + 4u + S.tiny_v)) & vm);
        } else r0 = *reinterpret_cast<const uint4*>(ln.row(tgt));
        if (learned) {
            const uint16_t* fw = reinterpret_cast<const uint16_t*>(body + S.off_fw);
            const uint32_t* lists = reinterpret_cast<const uint32_t*>(tb + C.hot_fwlist);
            ln.fw_src = fw[lists[src] >> 16]; ln.fw_tgt = fw[lists[tgt] & 0xFFFFu];
        }
        const uint64_t pt = (uint64_t)r0.x | ((uint64_t)r0.y << 32);
        ln.props = pt & ROW_PROPS_MASK; ln.tags = (uint32_t)(pt >> 60);
        ln.ever = r0.z; ln.since = r0.w;
        STAMP(3);  // row landed
        ln.template act<!PK, DEFK, Hook::kObs>(X, skip ? -1.0 : 0.0, kind, src, tgt, X ? (k0 ? a2 : (k1 ? cL + a3 : 0u)) : 0u, (X & k2) ? a3 : 0u, triple);
        // unchanged rows are written back as they were
        if (PK) {
            const uint32_t w = S.tiny_pack(ln.props, ln.tags, ln.ever, ln.since);
            reinterpret_cast<uint32_t*>(body + S.off_rows)[tgt] = w;
            if (PHASE == 0) { ln.rows_in_regs = true; ln.prw0 = rw0; ln.prw1 = rw1; ln.prw2 = rw2; ln.prw3 = rw3; ln.ptgt = tgt; ln.pword = w; }
        }
        else {
            const uint64_t wpt = ln.props | ((uint64_t)ln.tags << 60);
            *reinterpret_cast<uint4*>(ln.row(tgt)) = make_uint4((uint32_t)wpt, (uint32_t)(wpt >> 32), ln.ever, ln.since);
        }
        STAMP(4);      // attacker logic and row store done
        const uint32_t nf = (oob ? F_OOB : 0u) | ((uint32_t)ln.okind << F_KIND_SHIFT) | ((uint32_t)ln.olevel << F_LEVEL_SHIFT) |
                            ((uint32_t)ln.new_nodes << F_NEWNODES_SHIFT) | ((uint32_t)ln.new_creds << F_NEWCREDS_SHIFT);
        flags = live ? nf : ((PHASE == 1 && !ended && skip_env) ? (old_flags | F_SKIP) : old_flags);
        step += live ? 1u : 0u;
        // the observation is assembled here: after the attacker's action, before the defender's turn (env.py:1153 vs 1156-1158)
        if constexpr (Hook::kObs) hook.stage_obs(S, C, ln, flags, !skip_env);
    } else {
        oob = live & ((old_flags & F_OOB) != 0);
        ln.raw = pending;
        flags = old_flags & ~F_SKIP;
    }

    if (PHASE == 1) {
        if (live) S.pending[e] = ln.raw;
        S.h0[e] = make_uint4(step, flags, ln.n_disc | (ln.n_creds << 16), ln.owned | (ln.dclk << 16));
    } else {
        double reward = 0.0;
        bool done = false;
        if (has_def) {
            if (live & !oob) {
                uint64_t fresh[WT];
#pragma unroll
                for (int w = 0; w < WT; ++w) fresh[w] = 0ull;
                h1.y = ln.defender_tick(back);
                ln.defender_scan(step, episode, io, fresh);
#pragma unroll
                for (int w = 0; w < WT; ++w)       // the slot now holds the nodes re-imaged at this tick (released 16 ticks on)
                    if (fresh[w] != back[w]) S.ring[((ln.dclk & 15u) * WT + (uint32_t)w) * S.E + e] = fresh[w];
                ln.dclk = (ln.dclk + 1u) & 0xFFFFu;
            }
        }
        if (ere) {
            if (live & !oob) {                           // on_attacker_step_taken, then ExternalRandomEvents.step (env.py:1156-1158)
                const EreView V{body, C, T.base, S.N};
                h1.y = V.availability();
                uint32_t di = 0;
                random_events_step(V, [&]() { return ln.draw(di++, step, episode, io); });
            }
        }
        {
            // goals (env.py:1080-1116) on the state AFTER the defender acted, availability from BEFORE its scan
            const bool attacker_goal = (g_has != 0) & !(h1.x < __longlong_as_double(g_reward)) & !(ln.owned < g_own) &
                                       !(ln.owned < g_pctmin) &
                                       !(def_avail && h1.y >= __longlong_as_double(g_low));
            const bool sla_broken = def_avail && h1.y < __longlong_as_double(g_sla);
            const bool evicted = (g_evict != 0) & (ln.owned == 0);
            const bool win = attacker_goal | sla_broken;
            const bool play = live & !oob;
            done = play & (win | evicted);
            const double r_play = win ? __longlong_as_double(g_win) : (evicted ? __longlong_as_double(g_lose) : (ln.raw > 0.0 ? ln.raw : 0.0));   // max(0, reward), env.py:1169
            reward = play ? r_play : 0.0;
        }
        h1.x += reward;
        const bool trunc = live & !done & (g_max != 0) & (step >= g_max);
        {
            iok.reward[e] = (float)reward;
            iok.terminated[e] = live ? (done ? 1 : 0) : (uint8_t)((old_flags & F_DONE) ? 1 : 0);
            if (iok.truncated) iok.truncated[e] = live ? (trunc ? 1 : 0) : (uint8_t)((old_flags & F_TRUNC) ? 1 : 0);
            if (iok.availability) iok.availability[e] = h1.y;
            if (iok.step_count) iok.step_count[e] = (int32_t)step;
            if (iok.oob) iok.oob[e] = oob ? 1 : 0;
            if (iok.raw_reward) iok.raw_reward[e] = live ? (float)ln.raw : 0.0f;
            need_reset = (done | trunc) & (g_auto != 0);
            hk_reward = (float)reward;
            hk_done = live ? done : ((old_flags & F_DONE) != 0);
            flags |= (done ? F_DONE : 0u) | (trunc ? F_TRUNC : 0u);
            S.h0[e] = make_uint4(step, flags, ln.n_disc | (ln.n_creds << 16), ln.owned | (ln.dclk << 16));
            S.h1[e] = h1;
        }
    }
    {
        // sets the phase can have changed go back whole (WT == 1: one coalesced 8-byte store per set beats a compare and a
        // branch); an env about to be reset gets its columns rewritten below, after these stores in program order
        if (PK) {
            if (PHASE != 2 || has_def) {
                uint32_t f[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) f[q] = (uint32_t)ln.m[2 * q][0] | ((uint32_t)ln.m[2 * q + 1][0] << 16);
                reinterpret_cast<uint4*>(S.masks)[e] = make_uint4(f[0], f[1], f[2], f[3]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < M_COUNT; ++k) {
                const bool attacker_set = k != M_RUN;
                const bool defender_set = k == M_RUN || k == M_INST || k == M_PLO || k == M_PHI;
                if (!((PHASE != 2 && attacker_set) || (PHASE != 1 && has_def && defender_set))) continue;
#pragma unroll
                for (int w = 0; w < WT; ++w)
                    if (WT == 1 || ln.m[k][w] != m0[k][w]) S.masks[((uint32_t)k * WT + (uint32_t)w) * S.E + e] = ln.m[k][w];
            }
        }
    }
    }
    STAMP(5);              // all stores of the step retired
    if (PHASE != 1 && PK && C.init_image_ok) {
        // Packed batches: an env that just ended is re-initialised by its OWN lane with stores only — the body's reset image (<= 16 x 16
        // bytes), the sets and the header come from the config through the scalar cache, the episode counter was fetched at level 1.
        // (Round 2 let the whole wavefront copy the image from memory behind a fence, like the large layouts below: with ~1 % of the
        // envs ending per step that cost every other wavefront a write-acknowledgement round trip — 6.06 vs 5.05 us per step.)
        if (need_reset) {
            uint4* dst = reinterpret_cast<uint4*>(S.body + (size_t)e * S.body_stride);
            const uint32_t nv = S.body_stride >> 4;
#pragma unroll
            for (uint32_t i = 0; i < 16u; ++i)
                if (i < nv) dst[i] = make_uint4(C.init_image[4 * i], C.init_image[4 * i + 1], C.init_image[4 * i + 2], C.init_image[4 * i + 3]);
            reinterpret_cast<uint4*>(S.masks)[e] = make_uint4(C.init_packed[0], C.init_packed[1], C.init_packed[2], C.init_packed[3]);
            if (S.ring) for (uint32_t s = 0; s < 16u; ++s) S.ring[(size_t)s * S.E + e] = 0ull;
            S.h0[e] = make_uint4(0u, 0u, C.n_init, C.n_init);
            S.h1[e] = make_double2(0.0, 1.0);
            S.episode[e] = episode + 1u;
            S.pending[e] = 0.0;
        }
    } else if (PHASE != 1) {
        // Envs that just ended are re-initialised by the whole wavefront: ballot the lanes that need it, then all
        // 64 lanes copy the reset image of one env at a time with 16-byte accesses (coalesced), instead of one
        // lane writing N rows serially.
        const uint64_t rm = __ballot(need_reset);
        if (rm) {
            __threadfence_block();   // the owner lane's row stores must land before other lanes overwrite them
            const uint32_t lane = threadIdx.x & 63u;
            const uint32_t wave_base = e - lane;
            uint64_t mm = rm;
            while (mm) {
                const uint32_t l = (uint32_t)__builtin_ctzll(mm);
                mm &= mm - 1;
                uint8_t* dst = S.body + (size_t)(wave_base + l) * S.body_stride;
                for (uint32_t off = lane * 16u; off < S.body_stride; off += 64u * 16u)
                    *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(S.init_body + off);
            }
            if (need_reset) reset_header(S, T, e, S.episode[e] + 1u);
        }
    }
    if constexpr (Hook::kFinish) hook.finish(S, C, T, e, active, hk_reward, hk_done, episode);
    }   // steps of this launch
#ifdef MCBS_DIAG
    STAMP(6);
    if (io.stamps && (threadIdx.x & 63u) == 0) {
        unsigned long long* o = io.stamps + (size_t)(e >> 6) * 8;
        for (int i = 0; i < 7; ++i) o[i] = st_[i];
        o[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// One launch = one CyberBattleEnv.step of every env.  The argument list must stay below 256 bytes (mcbs_api.hip make_io): the random
// agent's parameters therefore travel only with the looping variant below, which is launched once per K steps.
template <int PHASE, int WTP, bool TOPO_LDS, int DEFK>
__global__ __launch_bounds__(256) void step_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, StepIO io) {
    NoHook nh;
    step_body<PHASE, WTP, TOPO_LDS, DEFK, false>(S, T, Cp, io, RollArgs{}, nh);
}

// mcbs_step_many / mcbs_rollout_random: io.n_steps consecutive steps in one launch; `roll` is a kernel ARGUMENT (nothing in device
// memory is patched per call, so launches on different streams or inside a stream capture cannot see each other's mode).
template <int WTP, bool TOPO_LDS, int DEFK>
__global__ __launch_bounds__(256) void step_many_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, StepIO io, RollArgs roll) {
    NoHook nh;
    step_body<0, WTP, TOPO_LDS, DEFK, true>(S, T, Cp, io, roll, nh);
}

} // namespace mcbs
