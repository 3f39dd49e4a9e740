// mcbs_step_coop.hip — the fused step (CyberBattleEnv.step, one launch) for LARGE topologies: G lanes per environment.
//
// mcbs_step.hip advances one env per LANE.  That is the right shape while an env's sets fit one 64-bit word (packed batches,
// <= 64 nodes): the transition is O(1) bit operations.  With more than 64 nodes every set is G = 2 or 4 words, the one-lane
// kernel holds 8 x G words per lane and pays ~30 instructions of word-select arithmetic per set access (1 255 VALU instructions
// per wavefront and step at 256 nodes, profiles/round2_notes.md section 2), and BASELINE.json's shards of those topologies
// (8 192 x Chain-100, 16 384 x Random-256) are only 128 / 256 wavefronts on 1 024 SIMDs: three quarters of the chip idle while
// each busy SIMD executes a long serial program.
//
// Here G consecutive lanes form one env's GROUP (G = words per set: 32 or 16 envs per wavefront, so the same shards launch
// 256 / 1 024 wavefronts) and lane w of the group OWNS WORD w of every set of the env:
//   * a membership test is one shift on the owner's word, and the answers of several tests are packed into one dword and
//     OR-reduced over the group with one or two DPP quad_perm moves (no LDS, no ballot: the groups are aligned pairs / quads);
//   * an update touches the owner's word only (a predicated OR / AND-NOT), and every lane stores just its own words that changed;
//   * the first 16 entries PER LANE of the discovery order and of the credential cache ride with the header (32 / 64 entries per
//     env instead of 16), so an action whose indices are below that needs no dependent list load;
//   * everything that is one value per env — header, action, checks, reward arithmetic, Philox draws, goals — is computed
//     redundantly by the G lanes (same instructions the one-lane kernel issues once per env; lanes are what this launch has
//     spare), loaded from one address per group, and stored by lane 0 of the group;
//   * the defender's availability popcount is a DPP add over the group; re-imaging clears bits in the owner's words.
// Control flow inside a group is uniform by construction (every branch condition is a per-env value), which is what makes the DPP
// exchanges legal under divergence BETWEEN groups.
//
// Same state layout, same rules, same memory-level discipline as mcbs_step.hip (all loads before the first store, one explicit
// wait, table loads fenced together); either kernel can advance the same batch (MCBS_NO_COOP=1 selects the one-lane kernel:
// tests/test_gpu_parity.py steps both side by side).  Scope: whole step (mcbs_step), no defender or ScanAndReimage, sets of at
// most 4 words; split phases, the looping variant, the learned / random-events defenders and "wide" credential sets stay on
// mcbs_step.hip.  Rules restated (citations as in mcbs_step.hip): actions.py:325-423 (__process_outcome), 524-606 (connect),
// 700-746 (reimage_node, on_attacker_step_taken), defender.py:42-55 (ScanAndReimage), env.py:707-751, 1080-1116, 1145-1185.
#pragma once
#include "mcbs_step.hip"

namespace mcbs {

// ------------------------------ exchanges inside a group of G = 2 or 4 aligned lanes (DPP quad_perm) ------------------------------
constexpr int DPP_XOR1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;   // quad_perm [2,3,0,1]
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int G>
__device__ __forceinline__ uint32_t g_or(uint32_t v) {
    v |= dpp_u32<DPP_XOR1>(v);
    if (G == 4) v |= dpp_u32<DPP_XOR2>(v);
    return v;
}
template <int G>
__device__ __forceinline__ uint32_t g_add(uint32_t v) {
    v += dpp_u32<DPP_XOR1>(v);
    if (G == 4) v += dpp_u32<DPP_XOR2>(v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = dpp_u32<CTRL>((uint32_t)b), hi = dpp_u32<CTRL>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)((uint64_t)lo | ((uint64_t)hi << 32)));
}
template <int G>   // (a + b) + (c + d) in every lane: the same bits whichever lane computes it (fp addition commutes)
__device__ __forceinline__ double g_add_f64(double v) {
    v = v + dpp_f64<DPP_XOR1>(v);
    if (G == 4) v = v + dpp_f64<DPP_XOR2>(v);
    return v;
}
// word J of a set, from its owner, to every lane of the group (J is a constant: quad_perm [J,J,J,J], pairs: [J,J,2+J,2+J])
template <int G, int J>
__device__ __forceinline__ uint64_t g_word(uint64_t mine) {
    constexpr int CTRL = G == 4 ? J * 0x55 : (J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6));
    const uint32_t lo = dpp_u32<CTRL>((uint32_t)mine), hi = dpp_u32<CTRL>((uint32_t)(mine >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// ------------------------------ per-lane working set: this lane's word of every set, the env's scalars replicated ------------------------------
template <int G, int DK>
struct CoopLane {
    const DevState& S;
    const StepCfg& C;
    const uint8_t* tb;      // hot image (Topo::hot, read through L1 / L2)
    uint32_t e, w;          // env, and the word of every set this lane owns (= its position in the group)
    uint8_t* body;
    uint32_t n_disc, n_creds, owned, dclk;
    uint64_t m[M_COUNT];
    uint64_t props;         // the target node's row, replicated
    uint32_t ever, since, tags;
    double raw;
    int okind, olevel, new_nodes, new_creds;

    __device__ __forceinline__ const HotNode* NS(uint32_t n) const { return reinterpret_cast<const HotNode*>(tb + C.hot_node) + n; }
    __device__ __forceinline__ Row* row(uint32_t n) const { return reinterpret_cast<Row*>(body + S.off_rows) + n; }
    __device__ __forceinline__ uint8_t* disc_list() const { return body + S.off_disc; }
    __device__ __forceinline__ uint16_t* cred_list() const { return reinterpret_cast<uint16_t*>(body + S.off_cred); }

    // bit n of set k as THIS lane sees it: the bit if this lane owns n's word, else 0 (OR over the group = the membership test)
    __device__ __forceinline__ uint32_t lbit(int k, uint32_t n) const {
        return (uint32_t)((m[k] >> (n & 63u)) & 1ull) & (uint32_t)((n >> 6) == w);
    }
    // element n as a one-word set in the owner lane, empty elsewhere or when !on
    __device__ __forceinline__ uint64_t obit(uint32_t n, bool on) const { return (on & ((n >> 6) == w)) ? (1ull << (n & 63u)) : 0ull; }
    __device__ __forceinline__ uint64_t valid_bits() const {
        if (w * 64u >= S.N) return 0ull;
        const uint32_t rem = S.N - w * 64u;
        return rem >= 64u ? ~0ull : ((1ull << rem) - 1ull);
    }
    __device__ __forceinline__ double avail_term(uint32_t n) const { return reinterpret_cast<const double*>(tb + C.hot_avail)[n]; }

    // One action of one env, by the G lanes of its group: mcbs_step.hip Lane::act with the set accesses distributed (same predicated
    // flow, same priorities; X, raw_nx, kind and the indices are per-env values, identical in the G lanes).
    __device__ __forceinline__ void act(bool X, double raw_nx, int kind, uint32_t src, uint32_t tgt, uint32_t col, uint32_t port, uint32_t triple) {
        const bool k2 = kind == 2;
        const uint4* tp = reinterpret_cast<const uint4*>(NS(tgt));
        const uint4 t0 = tp[0], t1 = tp[1];              // {props lo, hi, value, fw_in_allow} {fw_out_allow, listen, svc, flags}
        const uint64_t t_props = (uint64_t)t0.x | ((uint64_t)t0.y << 32);
        const int t_value = (int)t0.z;
        const uint32_t src_fw_out = NS(src)->fw_out_allow;
        const uint64_t auth = reinterpret_cast<const uint64_t*>(tb + C.hot_auth)[(tgt * C.P + port) * C.auth_words + (triple >> 6)];
        const uint4* dp = reinterpret_cast<const uint4*>(tb + C.hot_desc + (tgt * (C.L + C.R) + col) * (uint32_t)sizeof(HotDesc));
        const uint4 d0 = dp[0], d1 = dp[1];   // {cost lo,hi, probe lo,hi} {payload_off, cnt | tt << 16, kind | level << 8 | slot << 16, -}
        const uint4 d2 = dp[2], d3 = dp[3];   // the first four payload entries
        __builtin_amdgcn_sched_barrier(0);    // all eight table loads go out together (mcbs_step.hip)

        // six membership tests, one exchange
        uint32_t q = lbit(M_INST, src) | (lbit(M_RUN, tgt) << 1) | (lbit(M_INST, tgt) << 2) | (lbit(M_EVER, tgt) << 3) |
                     (lbit(M_PLO, tgt) << 4) | (lbit(M_PHI, tgt) << 5);
        q = g_or<G>(q);
        const bool src_owned = q & 1u, running = (q >> 1) & 1u, already = (q >> 2) & 1u, ever_owned = (q >> 3) & 1u;
        const uint32_t priv = (q >> 4) & 3u;

        // ---- connect_to_remote_machine checks, in the reference's order ----
        const bool out_ok = (src_fw_out >> port) & 1u, in_ok = (t0.w >> port) & 1u;
        const bool authorized = (auth >> (triple & 63u)) & 1ull;
        const bool reach = out_ok & in_ok & (bool)((t1.y >> port) & 1u);
        const bool c_proceed = reach & running & authorized;
        const double c_fail_raw = (reach & !running) ? 0.0 : -10.0;

        // ---- exploit checks: MACHINE_NOT_RUNNING 0 > SUPSPICIOUSNESS -5 > LOCAL_EXPLOIT_FAILED -20 / FAILED_REMOTE_EXPLOIT -50 > REPEAT -1 ----
        const uint32_t vk = d1.z & 0xFFu, level = (d1.z >> 8) & 0xFFu;
        const bool present = vk != 0xFFu;
        const bool pre_ok = ((d1.y >> 16) >> tags) & 1u;
        const bool esc = !k2 & (vk == MCBS_OUT_PRIVILEGE_ESCALATION);
        const bool repeat_esc = esc & (bool)((tags >> level) & 1u);
        const bool x_proceed = running & present & pre_ok & !repeat_esc;
        const double x_fail_raw = !running ? 0.0 : (!present ? -5.0 : (!pre_ok ? (kind == 0 ? -20.0 : -50.0) : -1.0));
        const int x_fail_kind = (running & present) ? (!pre_ok ? MCBS_OUT_EXPLOIT_FAILED : MCBS_OUT_PRIVILEGE_ESCALATION) : MCBS_OUT_NONE;
        const int x_lvl = (running & present & pre_ok & esc) ? (int)level : 0;

        const bool so = X & src_owned;
        const bool go = so & (k2 ? c_proceed : x_proceed);
        const bool xb = go & !k2;
        const uint32_t vk_ok = k2 ? (uint32_t)MCBS_OUT_LATERAL_MOVE : vk;
        const uint32_t own_level = esc ? level : 1u;

        // ---- __mark_node_as_owned (actions.py:251-275) ----
        const bool newly = go & !already & (k2 | esc | (vk == MCBS_OUT_LATERAL_MOVE));
        const bool first_time = newly & !ever_owned;
        const uint32_t np = priv > own_level ? priv : own_level;
        const bool chg = newly & (np != priv);
        const uint64_t bn = obit(tgt, newly), bc = obit(tgt, chg);
        m[M_EVER] |= bn;
        m[M_INST] |= bn;
        m[M_PLO] = (m[M_PLO] & ~bc) | ((np & 1u) ? bc : 0ull);
        m[M_PHI] = (m[M_PHI] & ~bc) | ((np & 2u) ? bc : 0ull);
        owned += (chg & (priv == 0u)) ? 1u : 0u;
        props |= newly ? t_props : 0ull;
        tags |= (xb & esc) ? (1u << own_level) : 0u;

        // ---- exploit bookkeeping (actions.py:386-423) ----
        const int r = first_time ? t_value : 0;
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
        constexpr uint32_t PF = 8;
        uint2 pre[PF] = {make_uint2(d2.x, d2.y), make_uint2(d2.z, d2.w), make_uint2(d3.x, d3.y), make_uint2(d3.z, d3.w),
                         make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u)};
#pragma unroll
        for (uint32_t i = 4; i < PF; ++i)
            if (__ballot(i < cnt)) pre[i] = pl[i < cnt ? i : 0u];
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): every load of the step has landed; stores only from here (mcbs_step.hip)
        auto leak = [&](const uint2 p) {                 // one LeakedCredentials / LeakedNodesId entry {node | cred << 16, triple | port << 16}
            const uint32_t pn = p.x & 0xFFFFu, pc = p.x >> 16, pt = p.y & 0xFFFFu;
            const uint32_t seen = g_or<G>(lbit(M_DISC, pn) | (lbit(M_GATH, pc) << 1) | (lbit(M_CACH, pt) << 2));
            const bool new_n = !(seen & 1u), new_g = creds & !(seen & 2u), new_c = creds & !(seen & 4u);
            // appended past the list's end whether or not the element is new (one slack slot; the count only advances for a new one);
            // the G lanes write the same value to the same address
            disc_list()[n_disc] = (uint8_t)pn;
            cred_list()[n_creds] = (uint16_t)pt;
            m[M_DISC] |= obit(pn, new_n);
            m[M_GATH] |= obit(pc, new_g);
            m[M_CACH] |= obit(pt, new_c);
            n_disc += new_n; nn += new_n; nc += new_g; n_creds += new_c; ncache += new_c;
        };
#pragma unroll
        for (uint32_t i = 0; i < PF; ++i) {
            if (!__ballot(i < cnt)) break;
            if (i < cnt) leak(pre[i]);
        }
        for (uint32_t i = PF; i < cnt; ++i) leak(pl[i]);
        rx += 5 * (int)nn + 3 * (int)nc;
        const double x_raw = (double)rx - __hiloint2double((int)d0.y, (int)d0.x);
        const double c_raw = already ? -1.0 : (double)r;

        raw = !X ? raw_nx : (!src_owned ? -1.0 : (!go ? (k2 ? c_fail_raw : x_fail_raw) : (k2 ? c_raw : x_raw)));
        okind = go ? (int)vk_ok : ((so & !k2) ? x_fail_kind : MCBS_OUT_NONE);
        olevel = (so & !k2) ? x_lvl : 0;
        new_nodes = (int)nn;
        new_creds = (int)ncache;
    }

    // ---- defender (every lane of the group draws the same numbers) ----
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

    // on_attacker_step_taken (actions.py:714-746); `back` = this lane's word of this tick's ring slot.  Returns the availability.
    __device__ __forceinline__ double defender_tick(uint64_t back) {
        m[M_RUN] |= back;
        const uint64_t im = ~m[M_RUN] & valid_bits();
        const uint32_t imaging = g_add<G>((uint32_t)__popcll(im));
        if (!imaging) return C.full_availability;
        double s;
        if (C.avail_uniform) {
            s = C.full_sum - (double)imaging * C.avail_term0;
        } else if (C.avail_any_order) {   // exact in any order: each lane sums the terms of its own word's imaging nodes, the group adds up
            double sub = 0.0;
            uint64_t bits = im;
            while (bits) { const uint32_t b = (uint32_t)__builtin_ctzll(bits); bits &= bits - 1; sub += avail_term(w * 64u + b); }
            s = C.full_sum - g_add_f64<G>(sub);
        } else {                          // the reference's node-order sum: every lane gathers the G words and walks all nodes
            uint64_t all[4] = {g_word<G, 0>(m[M_RUN]), g_word<G, 1>(m[M_RUN]), 0ull, 0ull};
            if (G == 4) { all[2] = g_word<G, 2 % G>(m[M_RUN]); all[3] = g_word<G, 3 % G>(m[M_RUN]); }
            s = 0.0;
            for (uint32_t n = 0; n < S.N; ++n) {
                const uint32_t j = n >> 6;
                const uint64_t wj = j == 0 ? all[0] : (j == 1 ? all[1] : (j == 2 ? all[2] : all[3]));
                if ((wj >> (n & 63u)) & 1ull) s += avail_term(n);
            }
        }
        return s / C.total_sla_weight;
    }

    // ScanAndReimageCompromisedMachines.step (defender.py:42-55) + reimage_node (actions.py:700-712); nodes re-imaged now go to `fresh`
    __device__ __forceinline__ void defender_scan(uint32_t step, uint32_t episode, const StepIO& io, uint64_t& fresh) {
        if (step % C.scan_frequency) return;
        uint32_t det = 0;
        const uint64_t r0 = C.reimagable[0], r1 = C.reimagable[1], r2 = C.reimagable[2], r3 = C.reimagable[3];   // scalar config words
        const uint64_t reim = w == 0 ? r0 : (w == 1 ? r1 : (w == 2 ? r2 : r3));
        for (uint32_t i = 0; i < C.scan_capacity; ++i) {
            int n = (int)floor(draw(i, step, episode, io) * (double)S.N);
            if (n >= (int)S.N) n = (int)S.N - 1;
            const uint32_t un = (uint32_t)n;
            const uint32_t q = g_or<G>(lbit(M_RUN, un) | (lbit(M_INST, un) << 1) | ((uint32_t)((reim >> (un & 63u)) & 1ull) & (uint32_t)((un >> 6) == w)) << 2 |
                                       (lbit(M_PLO, un) << 3) | (lbit(M_PHI, un) << 4));
            if ((q & 3u) != 3u) continue;                       // not Running, or not infected
            const double d = draw(C.scan_capacity + det, step, episode, io);
            det += 1;
            if (!(d <= C.scan_probability) || !(q & 4u)) continue;
            // reimage_node: agent removed, privilege NoAccess, Imaging; every earlier attack now predates last_reimaging
            if (w == 0) row(un)->since = 0;
            const uint64_t bit = obit(un, true);
            m[M_INST] &= ~bit; m[M_PLO] &= ~bit; m[M_PHI] &= ~bit; m[M_RUN] &= ~bit;
            fresh |= bit;
            owned -= (q >> 3) ? 1u : 0u;
        }
    }
};

// Header + set words of a freshly reset env, lane w writing word w of every set (reset_header of mcbs_device.h, distributed)
template <int G>
__device__ __forceinline__ void reset_words(const DevState& S, const Topo& T, uint32_t e, uint32_t w, uint32_t episode) {
    const mcbs_topo_header& H = T.H();
    const uint8_t* order = T.base + H.off_init_order;
    const mcbs_node_static* ns = reinterpret_cast<const mcbs_node_static*>(T.base + H.off_node);
    const uint32_t n_init = H.n_init_owned;
    uint64_t mk = 0, lo = 0, hi = 0;
    for (uint32_t i = 0; i < n_init; ++i) {
        const uint32_t n = order[i];
        if ((n >> 6) != w) continue;
        const uint64_t bit = 1ull << (n & 63u);
        mk |= bit;
        if (ns[n].priv0 & 1u) lo |= bit;
        if (ns[n].priv0 & 2u) hi |= bit;
    }
    uint64_t run = 0ull;
    if (w * 64u < S.N) { const uint32_t rem = S.N - w * 64u; run = rem >= 64u ? ~0ull : ((1ull << rem) - 1ull); }
    const uint64_t v[M_COUNT] = {mk, mk, mk, run, lo, hi, 0ull, 0ull};     // M_DISC, M_INST, M_EVER, M_RUN, M_PLO, M_PHI, M_GATH, M_CACH
#pragma unroll
    for (int k = 0; k < M_COUNT; ++k) S.masks[((size_t)k * G + w) * S.E + e] = v[k];
    if (S.ring) for (uint32_t s = 0; s < 16u; ++s) S.ring[((size_t)s * G + w) * S.E + e] = 0ull;
    if (w == 0) {
        S.h0[e] = make_uint4(0u, 0u, n_init, n_init);
        S.h1[e] = make_double2(0.0, 1.0);
        S.episode[e] = episode;
        S.pending[e] = 0.0;
    }
}

// One launch = one CyberBattleEnv.step of every env; 64-thread workgroups, 64 / G envs each.
template <int G, int DEFK>
__global__ __launch_bounds__(64) void step_coop_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, StepIO io) {
    static_assert(G == 2 || G == 4, "group size = words per set");
    static_assert(DEFK == MCBS_DEFENDER_NONE || DEFK == MCBS_DEFENDER_SCAN_AND_REIMAGE, "defender kinds of the cooperative kernel");
    constexpr bool has_def = DEFK == MCBS_DEFENDER_SCAN_AND_REIMAGE;
    constexpr uint32_t EPW = 64u / G, HEAD = 16u * G;     // envs per wavefront; list entries fetched with the header
    const StepCfg& C = *Cp;
    const uint32_t lane = threadIdx.x, w = lane % G;
    const uint32_t e = blockIdx.x * EPW + lane / G;
    const bool active = e < S.E;
    const uint32_t ec = active ? e : 0u;

    // ---------------- level 1: addresses depend on the env index (and the lane's word) only ----------------
    uint8_t* body = S.body + (size_t)ec * S.body_stride;
    const uint4 h0 = S.h0[ec];
    const uint32_t* ap = reinterpret_cast<const uint32_t*>(io.actions) + (size_t)ec * 5;
    const uint4 a03 = make_uint4(ap[0], ap[1], ap[2], ap[3]);
    const uint32_t a4 = ap[4];
    const uint4 dhead = *reinterpret_cast<const uint4*>(body + S.off_disc + 16u * w);            // entries 16w .. 16w+15 of the discovery order
    const uint4 chead0 = *reinterpret_cast<const uint4*>(body + S.off_cred + 32u * w);           // entries 16w .. 16w+15 of the credential cache
    const uint4 chead1 = *reinterpret_cast<const uint4*>(body + S.off_cred + 32u * w + 16u);
    uint64_t m0[M_COUNT];
#pragma unroll
    for (int k = 0; k < M_COUNT; ++k) m0[k] = S.masks[((uint32_t)k * G + w) * S.E + ec];
    double2 h1 = S.h1[ec];
    uint32_t episode = 0;
    if (has_def && C.rng_kind == MCBS_RNG_PHILOX) episode = S.episode[ec];

    uint32_t cL = C.L, cR = C.R, cP = C.P;
    unsigned long long g_reward = __double_as_longlong(C.goal_reward), g_low = __double_as_longlong(C.goal_low_availability),
                       g_sla = __double_as_longlong(C.maintain_sla),
                       g_win = __double_as_longlong(C.winning_reward), g_lose = __double_as_longlong(C.losing_reward);
    uint32_t g_has = C.has_attacker_goal, g_own = C.goal_own_atleast, g_evict = C.defender_goal_eviction, g_auto = C.auto_reset, g_max = C.max_episode_steps,
             g_pctmin = C.goal_own_pct_min;
    asm volatile("" : "+s"(cL), "+s"(cR), "+s"(cP), "+s"(g_reward), "+s"(g_low), "+s"(g_pctmin), "+s"(g_sla), "+s"(g_win), "+s"(g_lose), "+s"(g_has),
                      "+s"(g_own), "+s"(g_evict), "+s"(g_auto), "+s"(g_max));
    const uint8_t* tb = T.hot;

    bool need_reset = false;
    if (active) {
        const uint32_t old_flags = h0.y;
        const bool ended = (old_flags & (F_DONE | F_TRUNC)) != 0;
        const bool skip_env = (int)a03.x == MCBS_ACTION_SKIP;
        const bool live = !ended & !skip_env;

        CoopLane<G, DEFK> ln{S, C, tb, ec, w, body, h0.z & 0xFFFFu, h0.z >> 16, h0.w & 0xFFFFu, h0.w >> 16, {}, 0ull, 0u, 0u, 0u, 0.0, MCBS_OUT_NONE, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < M_COUNT; ++k) ln.m[k] = m0[k];
        // level 2 (needs the header): this lane's word of this defender tick's ring slot
        const uint64_t back = has_def ? S.ring[((ln.dclk & 15u) * G + w) * S.E + ec] : 0ull;

        uint32_t step = h0.x, flags = old_flags;
        // ---------------- __execute_action (cyberbattle_env.py:707-751), index checks as booleans ----------------
        const int kind = (int)a03.x;
        const uint32_t a1 = a03.y, a2 = a03.z, a3 = a03.w;
        const bool k0 = kind == 0, k1 = kind == 1, k2 = kind == 2;
        const bool skip = k2 & (a4 >= ln.n_creds);
        const bool bad = (a1 >= ln.n_disc) | (a2 >= (k0 ? cL : ln.n_disc)) | (!k0 & (a3 >= (k1 ? cR : cP)));
        const bool oob = live & (!(k0 | k1 | k2) | (!skip & bad));
        const bool X = live & !skip & !oob & (k0 | k1 | k2);
        const uint32_t i1 = X ? a1 : 0u, i2 = (X & !k0) ? a2 : i1, i4 = (X & k2) ? a4 : 0u;
        // entry i of a list: the lane that fetched its 16-entry slice extracts it, the group ORs (entries >= HEAD: one more level)
        const uint32_t mine = (((i1 >> 4) == w) ? byte_of(dhead, i1 & 15u) : 0u) | ((((i2 >> 4) == w) ? byte_of(dhead, i2 & 15u) : 0u) << 8) |
                              ((((i4 >> 4) == w) ? half_of(chead0, chead1, i4 & 15u) : 0u) << 16);
        const uint32_t got = g_or<G>(mine);
        uint32_t src = got & 0xFFu, tgt = (got >> 8) & 0xFFu, triple = got >> 16;
        if ((i1 | i2 | i4) >= HEAD) {
            const uint32_t s2 = ln.disc_list()[i1], t2 = ln.disc_list()[i2], c2 = ln.cred_list()[i4];
            src = i1 >= HEAD ? s2 : src; tgt = i2 >= HEAD ? t2 : tgt; triple = i4 >= HEAD ? c2 : triple;
        }
        // ---------------- level 2: the target's row and the table look-ups ----------------
        const uint4 r0 = *reinterpret_cast<const uint4*>(ln.row(tgt));
        const uint64_t pt = (uint64_t)r0.x | ((uint64_t)r0.y << 32);
        ln.props = pt & ROW_PROPS_MASK; ln.tags = (uint32_t)(pt >> 60);
        ln.ever = r0.z; ln.since = r0.w;
        ln.act(X, skip ? -1.0 : 0.0, kind, src, tgt, X ? (k0 ? a2 : (k1 ? cL + a3 : 0u)) : 0u, (X & k2) ? a3 : 0u, triple);
        const uint64_t wpt = ln.props | ((uint64_t)ln.tags << 60);
        if (w == 0) *reinterpret_cast<uint4*>(ln.row(tgt)) = make_uint4((uint32_t)wpt, (uint32_t)(wpt >> 32), ln.ever, ln.since);   // unchanged rows go back as they were
        const uint32_t nf = (oob ? F_OOB : 0u) | ((uint32_t)ln.okind << F_KIND_SHIFT) | ((uint32_t)ln.olevel << F_LEVEL_SHIFT) |
                            ((uint32_t)ln.new_nodes << F_NEWNODES_SHIFT) | ((uint32_t)ln.new_creds << F_NEWCREDS_SHIFT);
        flags = live ? nf : old_flags;
        step += live ? 1u : 0u;

        double reward = 0.0;
        bool done = false;
        uint64_t fresh = 0ull;
        if (has_def) {
            if (live & !oob) {
                h1.y = ln.defender_tick(back);
                ln.defender_scan(step, episode, io, fresh);
                if (fresh != back) S.ring[((ln.dclk & 15u) * G + w) * S.E + e] = fresh;   // released 16 ticks from now (same slot)
                ln.dclk = (ln.dclk + 1u) & 0xFFFFu;
            }
        }
        {
            // goals (env.py:1080-1116) on the state AFTER the defender acted, availability from BEFORE its scan
            const bool attacker_goal = (g_has != 0) & !(h1.x < __longlong_as_double(g_reward)) & !(ln.owned < g_own) &
                                       !(ln.owned < g_pctmin) &
                                       !(has_def && h1.y >= __longlong_as_double(g_low));
            const bool sla_broken = has_def && h1.y < __longlong_as_double(g_sla);
            const bool evicted = (g_evict != 0) & (ln.owned == 0);
            const bool win = attacker_goal | sla_broken;
            const bool play = live & !oob;
            done = play & (win | evicted);
            const double r_play = win ? __longlong_as_double(g_win) : (evicted ? __longlong_as_double(g_lose) : (ln.raw > 0.0 ? ln.raw : 0.0));
            reward = play ? r_play : 0.0;
        }
        h1.x += reward;
        const bool trunc = live & !done & (g_max != 0) & (step >= g_max);
        need_reset = (done | trunc) & (g_auto != 0);
        flags |= (done ? F_DONE : 0u) | (trunc ? F_TRUNC : 0u);
        if (w == 0) {
            io.reward[e] = (float)reward;
            io.terminated[e] = live ? (done ? 1 : 0) : (uint8_t)((old_flags & F_DONE) ? 1 : 0);
            if (io.truncated) io.truncated[e] = live ? (trunc ? 1 : 0) : (uint8_t)((old_flags & F_TRUNC) ? 1 : 0);
            if (io.availability) io.availability[e] = h1.y;
            if (io.step_count) io.step_count[e] = (int32_t)step;
            if (io.oob) io.oob[e] = oob ? 1 : 0;
            if (io.raw_reward) io.raw_reward[e] = live ? (float)ln.raw : 0.0f;
            S.h0[e] = make_uint4(step, flags, ln.n_disc | (ln.n_creds << 16), ln.owned | (ln.dclk << 16));
            S.h1[e] = h1;
        }
        // every lane stores the words of its own that changed
#pragma unroll
        for (int k = 0; k < M_COUNT; ++k)
            if (ln.m[k] != m0[k]) S.masks[((uint32_t)k * G + w) * S.E + e] = ln.m[k];
    }
    // Envs that just ended are re-initialised by the whole wavefront (mcbs_step.hip): ballot the groups that need it, all 64 lanes copy
    // the reset image of one env at a time, then each lane of such a group resets its own word of every set.
    const uint64_t rm = __ballot(need_reset && w == 0u);
    if (rm) {
        __threadfence_block();
        uint64_t mm = rm;
        while (mm) {
            const uint32_t l = (uint32_t)__builtin_ctzll(mm);
            mm &= mm - 1;
            uint8_t* dst = S.body + (size_t)(blockIdx.x * EPW + l / G) * S.body_stride;
            for (uint32_t off = lane * 16u; off < S.body_stride; off += 64u * 16u)
                *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(S.init_body + off);
        }
        if (need_reset) reset_words<G>(S, T, e, w, S.episode[e] + 1u);
    }
}

} // namespace mcbs
