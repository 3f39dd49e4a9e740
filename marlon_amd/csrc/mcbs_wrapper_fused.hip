// mcbs_wrapper_fused.hip — one step of marlon's attacker wrappers for a batch of SMALL topologies in ONE launch (gfx950).
//
// What a trainer consumes is AttackerEnvWrapper.step (attack_wrapper.py:255-372) under MaskedDiscreteAttackerWrapper
// (action_masking.py:112-142) under SB3's DummyVecEnv (baseline_marlon_agent.py:100-167): decode the policy's action, step the
// environment, observe (after the attacker's action, BEFORE the defender's turn: env.py:1153 vs 1156-1158), book-keep, auto-reset.
// Round 2 ran that as three launches — decode + attacker half, observation (sixteen lanes per env), defender half + finish — whose
// kernels took 6.9 + 15.3 + 5.2 us at 65 536 Chain-10 envs; the observation kernel was bound by instruction ISSUE (104 instructions
// per env for 924 bytes), and every launch re-read the state the previous one had stored.
//
// Here ONE wavefront advances 64 envs through the whole wrapper step:
//   loads   : everything the step loads (mcbs_step.hip levels 1 and 2), the policy's action, the wrapper's counters — once;
//   compute : lane = env for the decode, the attacker's action, the defender / goals and the wrapper's bookkeeping (mcbs_step.hip's
//             step_body with three hook points, the same code the headline kernel runs);
//   observe : between the attacker's action and the defender's turn each lane hands what its env's observation is made of — list heads
//             (kept up to date in registers while leaked entries are appended), node rows, three sets, flags and counts — to the
//             workgroup through 132 bytes of LDS, straight from the registers the step holds; after the step all four wavefronts turn
//             the 64 hand-overs into records by discovery index (known properties, privilege, (node index, port) per cached credential);
//   stores  : state, outputs, the re-initialisation of the envs that ended (their own lane, stores only), and then the observation of
//             the wavefront's 64 envs STREAMED OUT COOPERATIVELY: the 64 envs' rows of each observation array are one contiguous
//             block of memory (43 KB of property flags, 6 KB of cache rows ...), so lane L of store k writes 16 bytes at
//             (64 k + L) x 16 of the block — every store instruction covers one whole kilobyte of consecutive addresses — and builds
//             its four dwords from the LDS record of the env they belong to.  An env that ended gets its last observation written
//             to the TERMINAL arrays and the reset observation (kept in LDS) to the live ones; an env whose action was intercepted
//             (undiscovered node index: attack_wrapper.py:286-308) keeps the observation it had.
// No load is issued behind a store anywhere in the launch, nothing is read back, and the observation costs ~25 instructions per 16
// output bytes of a whole wavefront instead of ~100 per env-quarter.  Values are those of obs_tiny_kernel + wrapper_finish_body
// field by field (tests/test_gpu_vecenv.py, tests/test_gpu_logits.py run both paths).
//
// Scope (mcbs_api.hip falls back to the three launches otherwise): packed batches (<= 16 nodes, < 16 credentials), no mask field
// requested (the policy masks its logits with mcbs_mask_logits), observation rows whose dword counts are multiples of four
// (Chain-10 @12/12: 168 / 12 / 24 / 20; ToyCtf @12/10: 120 / 12 / 20 / 20), not the random-events defender.
#pragma once
#include "mcbs_aux.hip"
#include "mcbs_obs.hip"
#include "mcbs_step.hip"

namespace mcbs {

constexpr uint32_t FUSED_FIELDS = 5;          // 0 scalars [7], 1 leaked_credentials [K, 4], 2 credential_cache_matrix [Cmax, 2], 3 properties [Nmax, NP], 4 privilege [Nmax]
constexpr uint32_t FUSED_FRESH_DWORDS = 1024; // LDS room for one reset observation
constexpr uint32_t FUSED_THREADS = 256;        // one workgroup = 64 envs: the first wavefront steps them (lane = env), all four stream the observation
constexpr uint32_t FUSED_STAGE_DWORDS = 37;   // per env; odd, so that lane-per-env accesses fall into 37 i mod 32: all banks
constexpr uint32_t FUSED_RAW_DWORDS = 33;     // per env: what the stepping lane hands over (list heads, rows, sets); odd as well

struct FusedArgs {                             // kernel argument, by value (the per-step output pointers change from step to step)
    mcbs_wrapper_buffers w;
    float modifier;
    int32_t max_timesteps, auto_reset;
    uint32_t Nmax, Cmax, K, NP;
    const int64_t* md;                         // exactly one of md [E, 10] / discrete [E]
    const int64_t* discrete;
    int32_t* decoded;                          // [E, 5] engine rows (output)
    int32_t* obs[FUSED_FIELDS];                // observation arrays (NULL: field not requested)
    int32_t* term[FUSED_FIELDS];               // terminal-observation arrays (auto_reset)
    const int32_t* fresh[FUSED_FIELDS];        // one row each: the observation of a freshly reset env (auto_reset)
    uint32_t dwords[FUSED_FIELDS];             // per env
    uint32_t fresh_off[FUSED_FIELDS];          // dword offset of the field's reset row in LDS
    FastDiv d_u4[FUSED_FIELDS];                // division by dwords / 4 (fields 1..4), by dwords (field 0)
    FastDiv dNP;
    ObsDigest* digest;
    const ObsDigest* reset_digest;
    const mcbs_triple* triples;                // the topology's (node, credential, port) triples (< 16 of them)
    uint32_t n_triples, pad;
};

struct FusedStage {                            // view of one env's 37 dwords of LDS: what the streaming loops read
    uint32_t* p;
    // [0..16] properties of the node at discovery index j (17th: zero), [17..20] privilege of the node at discovery index j, one byte each,
    // [21..28] (node index | port << 8) of cached credential r as u16, [29] flags word of the step, [30] n_disc | n_creds << 8 | live << 16 |
    // blank << 17 | ended << 18, [31] owned-source bits by discovery index, [32..35] discovery index of node n, one byte each
    __device__ __forceinline__ uint32_t& props(uint32_t j) const { return p[j]; }
    __device__ __forceinline__ uint8_t* privb() const { return reinterpret_cast<uint8_t*>(p + 17); }
    __device__ __forceinline__ uint32_t priv4(uint32_t r) const { return p[17 + r]; }
    __device__ __forceinline__ uint16_t* cred() const { return reinterpret_cast<uint16_t*>(p + 21); }
    __device__ __forceinline__ uint32_t& flags() const { return p[29]; }
    __device__ __forceinline__ uint32_t& meta() const { return p[30]; }
    __device__ __forceinline__ uint32_t& own() const { return p[31]; }
    __device__ __forceinline__ uint8_t* ext_of() const { return reinterpret_cast<uint8_t*>(p + 32); }
};
struct FusedRaw {                              // view of one env's 33 dwords of LDS: written by the stepping lane between the attacker's action and the defender's turn
    uint32_t* p;
    // [0..3] discovery order (16 node ids), [4..11] credential cache (16 triple ids, u16), [12..27] the sixteen 4-byte node rows (target's as the
    // attacker left it), [28] agent-installed | privilege bit 0 << 16, [29] privilege bit 1
    __device__ __forceinline__ const uint8_t* disc() const { return reinterpret_cast<const uint8_t*>(p); }
    __device__ __forceinline__ const uint16_t* cache() const { return reinterpret_cast<const uint16_t*>(p + 4); }
    __device__ __forceinline__ uint32_t row(uint32_t n) const { return p[12 + n]; }
};
constexpr uint32_t FM_LIVE = 1u << 16, FM_BLANK = 1u << 17, FM_ENDED = 1u << 18;

// The hooks of step_body (mcbs_step.hip NoHook lists them) for the fused wrapper step.  One object per lane.
struct FusedHook {
    static constexpr bool kAction = true, kObs = true, kFinish = true;
    const FusedArgs& A;
    uint32_t* lds;                 // [64 x 37] records | [64 x 33] hand-overs | [1024] reset observation | [32] triple table (node | port << 16) | [16] reset digest
    uint32_t lane;                 // thread index in the workgroup; the stepping wavefront's lanes: 0..63 = env in the workgroup
    // level-1 loads
    // the policy's action: ten named scalars, not an array — a select between ELEMENTS of an array that lives in an object whose address
    // escapes is compiled into a dynamically indexed stack array (DESIGN.md "toolchain note"), and drags the object and the kernel's
    // 600-byte argument block into scratch memory with it
    int64_t p0, p1, p2, p3, p4, p5, p6, p7, p8, p9;   // p0: the Discrete index, or MultiDiscrete component 0
    int32_t ts;
    int64_t n_valid, n_invalid;
    double ret;
    // decoded
    int32_t row[5];
    bool invalid;

    __device__ __forceinline__ FusedStage stage(uint32_t env_in_wg) const { return FusedStage{lds + env_in_wg * FUSED_STAGE_DWORDS}; }
    __device__ __forceinline__ FusedRaw raw(uint32_t env_in_wg) const { return FusedRaw{lds + 64u * FUSED_STAGE_DWORDS + env_in_wg * FUSED_RAW_DWORDS}; }
    __device__ __forceinline__ uint32_t* fresh_lds() const { return lds + 64u * (FUSED_STAGE_DWORDS + FUSED_RAW_DWORDS); }
    __device__ __forceinline__ uint32_t* triple_lds() const { return fresh_lds() + FUSED_FRESH_DWORDS; }
    __device__ __forceinline__ uint32_t* rdig_lds() const { return fresh_lds() + FUSED_FRESH_DWORDS + 32u; }

    __device__ __forceinline__ void load_action(uint32_t ec) {
        const int64_t* v = A.md ? A.md + (size_t)ec * 10 : A.discrete + ec;      // (one load of p0 for both encodings: two stores of loaded
        p0 = v[0];                                                               // values in the arms of a branch are merged into a store through a
        if (A.md) {                                                              // selected stack address)
            p1 = v[1]; p2 = v[2]; p3 = v[3]; p4 = v[4]; p5 = v[5]; p6 = v[6]; p7 = v[7]; p8 = v[8]; p9 = v[9];
        }
    }
    // (every index into the argument arrays is a compile-time constant — template parameters, not loop counters: an index the compiler
    // cannot fold makes it copy the 600-byte argument block to scratch memory at kernel entry)
    template <uint32_t F>
    __device__ __forceinline__ void fill_fresh(uint32_t t, uint32_t nt) {
        if (A.obs[F]) for (uint32_t i = t; i < A.dwords[F]; i += nt) fresh_lds()[A.fresh_off[F] + i] = (uint32_t)A.fresh[F][i];
    }
    // the reset observation and the reset digest into LDS, by the wavefronts that do not step (threads t of nt), while the first one does
    __device__ __forceinline__ void fill_reset_rows(uint32_t t, uint32_t nt) {
        if (!A.auto_reset) return;
        fill_fresh<0>(t, nt); fill_fresh<1>(t, nt); fill_fresh<2>(t, nt); fill_fresh<3>(t, nt); fill_fresh<4>(t, nt);
        if (t < 16u) rdig_lds()[t] = reinterpret_cast<const uint32_t*>(A.reset_digest)[t];
    }
    // issued behind the step's own level-1 loads (same wait): the wrapper's counters and the triple table (into LDS)
    __device__ __forceinline__ void level1(uint32_t ec) {
        ts = A.w.timesteps[ec];
        n_valid = A.w.valid_action_count[ec];
        n_invalid = A.w.invalid_action_count[ec];
        ret = A.w.episode_returns[ec];
        if (lane < 16u) {
            const mcbs_triple t = A.triples[lane < A.n_triples ? lane : 0u];
            triple_lds()[lane] = (uint32_t)t.node | ((uint32_t)t.port << 16);
        }
    }

    // AttackerEnvWrapper.step's decode + interception of undiscovered node indices (attack_wrapper.py:255-308, 236-253) and
    // MaskedDiscreteAttackerWrapper._decode (action_masking.py:112-142): decode_body of mcbs_aux.hip on registers
    __device__ __forceinline__ void decode(const DevState& S, const StepCfg& C, const uint4& h0, uint4& a03, uint32_t& a4) {
        const int64_t nd = (int64_t)(h0.z & 0xFFFFu);
        int64_t kind, a = 0, b = 0, c = 0, d = 0;
        if (A.md) {
            kind = p0;
            // (mask arithmetic, not selects: a select between two LOADED members is rewritten into a load from a selected address —
            // a dynamically indexed stack object, see above)
            const int64_t m0 = -(int64_t)(kind == 0), m1 = -(int64_t)(kind == 1), m2 = ~(m0 | m1);
            a = (p1 & m0) | (p3 & m1) | (p6 & m2);
            b = (p2 & m0) | (p4 & m1) | (p7 & m2);
            c = (p5 & m1) | (p8 & m2);
            d = p9 & m2;
        } else {
            const int64_t N = A.Nmax, P = C.P, Cm = A.Cmax, L = C.L, R = C.R;
            const int64_t connect_size = N * N * P * Cm, local_size = N * L;
            const int64_t idx = p0;
            const bool is_c = idx < connect_size, is_l = !is_c && idx < connect_size + local_size;
            const int64_t rel = is_c ? idx : (is_l ? idx - connect_size : idx - connect_size - local_size);
            const int64_t inner = is_c ? Cm : (is_l ? L : R);
            int64_t x0, q, qp, qn, qm, qr;
            if (idx >= 0 && idx < (1ll << 31)) {
                const uint32_t r32 = (uint32_t)rel, i32 = (uint32_t)inner, q32 = r32 / i32, p32 = (uint32_t)P, n32 = (uint32_t)N;
                const uint32_t qp32 = q32 / p32;
                x0 = r32 - q32 * i32; q = q32; qp = qp32; qr = q32 - qp32 * p32;
                qn = is_c ? qp32 / n32 : q32 / n32; qm = is_c ? qp32 - (uint32_t)qn * n32 : q32 - (uint32_t)qn * n32;
            } else {
                x0 = rel % inner; q = rel / inner; qp = q / P; qr = q % P;
                qn = is_c ? qp / N : q / N; qm = is_c ? qp % N : q % N;
            }
            (void)qp;
            kind = is_c ? 2 : (is_l ? 0 : 1);
            a = is_l ? q : qn;
            b = is_l ? x0 : qm;
            c = is_c ? qr : (is_l ? 0 : x0);
            d = is_c ? x0 : 0;
        }
        const bool ok = kind == 0 ? a < nd : ((kind == 1 || kind == 2) ? (a < nd && b < nd) : false);    // _action_in_discovered_range
        row[0] = ok ? (int32_t)kind : MCBS_ACTION_SKIP;
        row[1] = (int32_t)a; row[2] = (int32_t)b; row[3] = (int32_t)c; row[4] = (int32_t)d;
        invalid = !ok;
        a03 = make_uint4((uint32_t)row[0], (uint32_t)row[1], (uint32_t)row[2], (uint32_t)row[3]);
        a4 = (uint32_t)row[4];
    }

    // Between the attacker's action and the defender's turn: the stepping lane hands what its env's observation is made of to the workgroup
    // — list heads, node rows, the three sets — as it holds them in registers (nine 16-byte LDS writes), plus flags and counts.
    template <class LaneT>
    __device__ __forceinline__ void stage_obs(const DevState& S, const StepCfg& C, const LaneT& ln, uint32_t flags, bool not_skipped) {
        const FusedRaw rw = raw(lane);
        rw.p[0] = ln.rec_dh.x; rw.p[1] = ln.rec_dh.y; rw.p[2] = ln.rec_dh.z; rw.p[3] = ln.rec_dh.w;
        rw.p[4] = ln.rec_c0.x; rw.p[5] = ln.rec_c0.y; rw.p[6] = ln.rec_c0.z; rw.p[7] = ln.rec_c0.w;
        rw.p[8] = ln.rec_c1.x; rw.p[9] = ln.rec_c1.y; rw.p[10] = ln.rec_c1.z; rw.p[11] = ln.rec_c1.w;
        rw.p[12] = ln.prw0.x; rw.p[13] = ln.prw0.y; rw.p[14] = ln.prw0.z; rw.p[15] = ln.prw0.w;
        rw.p[16] = ln.prw1.x; rw.p[17] = ln.prw1.y; rw.p[18] = ln.prw1.z; rw.p[19] = ln.prw1.w;
        rw.p[20] = ln.prw2.x; rw.p[21] = ln.prw2.y; rw.p[22] = ln.prw2.z; rw.p[23] = ln.prw2.w;
        rw.p[24] = ln.prw3.x; rw.p[25] = ln.prw3.y; rw.p[26] = ln.prw3.z; rw.p[27] = ln.prw3.w;
        rw.p[12 + (ln.ptgt & 15u)] = ln.pword;              // the target's row as the attacker left it (same lane, program order)
        rw.p[28] = ((uint32_t)ln.m[M_INST][0] & 0xFFFFu) | ((uint32_t)ln.m[M_PLO][0] << 16);
        rw.p[29] = (uint32_t)ln.m[M_PHI][0];
        const FusedStage st = stage(lane);
        const bool blank = (flags & F_OOB) != 0;
        st.flags() = flags;
        st.meta() = ln.n_disc | (ln.n_creds << 8) | (not_skipped ? FM_LIVE : 0u) | (blank ? FM_BLANK : 0u);
    }

    // ... and the whole workgroup (thread `lane` of FUSED_THREADS) turns the 64 hand-overs into the records the streaming loops read:
    // pass A, one (env, discovery index) per thread and iteration; pass B, one (env, cached credential) — obs_tiny_kernel's staging, spread
    // over four wavefronts instead of serialised behind the stepping lane (2.7 of its 9 us).
    __device__ __forceinline__ void convert_nodes(const DevState& S) {
        const uint32_t pmask = (1u << S.tiny_p) - 1u;
        for (uint32_t item = lane; item < 64u * 16u; item += FUSED_THREADS) {
            const uint32_t env = item >> 4, j = item & 15u;
            const FusedRaw rw = raw(env);
            const FusedStage st = stage(env);
            const uint32_t meta = st.meta(), n_disc = meta & 0xFFu;
            const uint32_t n = rw.disc()[j] & 15u;
            const bool on = j < n_disc;
            const uint32_t sets = rw.p[28], phi = rw.p[29];
            st.props(j) = on ? (rw.row(n) & pmask) : 0u;
            st.privb()[j] = (on && !(meta & FM_BLANK)) ? (uint8_t)((((sets >> 16) >> n) & 1u) | (((phi >> n) & 1u) << 1)) : (uint8_t)0;
            if (on) st.ext_of()[n] = (uint8_t)j;
            const uint64_t owned = __ballot(on && ((sets >> n) & 1u));          // four envs per wavefront and iteration, 16 bits each
            if (j == 0u) {
                st.props(16) = 0u;
                st.own() = (meta & FM_BLANK) ? 0u : (uint32_t)(owned >> ((lane & 48u))) & 0xFFFFu;
            }
        }
    }
    __device__ __forceinline__ void convert_creds() {
        const uint32_t* tr = triple_lds();
        for (uint32_t item = lane; item < 64u * 16u; item += FUSED_THREADS) {
            const uint32_t env = item >> 4, r = item & 15u;
            const FusedRaw rw = raw(env);
            const FusedStage st = stage(env);
            const uint32_t t = tr[rw.cache()[r] & 15u];                          // node | port << 16 (entries beyond n_creds are never read)
            st.cred()[r] = (uint16_t)((uint32_t)st.ext_of()[t & 15u] | ((t >> 16) << 8));
        }
    }

    // After the step's own stores: the wrapper's bookkeeping (wrapper_finish_body, word for word) and the re-initialisation of an env that
    // ended (own lane, stores only).  The workgroup streams the observations afterwards (wrapper_fused_kernel).
    __device__ __forceinline__ void finish(const DevState& S, const StepCfg& C, const Topo& T, uint32_t e, bool active, float reward, bool terminated,
                                           uint32_t episode) {
        const mcbs_wrapper_buffers& w = A.w;
        bool done = false;
        if (active) {
            int32_t* o = A.decoded + (size_t)e * 5;
            o[0] = row[0]; o[1] = row[1]; o[2] = row[2]; o[3] = row[3]; o[4] = row[4];
            const_cast<uint8_t*>(w.invalid)[e] = invalid ? 1 : 0;
            const int32_t t = ts + 1;
            const float shaped = reward + (invalid ? A.modifier : 0.0f);
            const double r2 = ret + (double)shaped;
            const bool trunc = t >= A.max_timesteps;
            done = terminated || trunc;
            const bool clear = done && A.auto_reset;
            w.timesteps[e] = clear ? 0 : t;
            w.valid_action_count[e] = clear ? 0 : n_valid + (invalid ? 0 : 1);
            w.invalid_action_count[e] = clear ? 0 : n_invalid + (invalid ? 1 : 0);
            w.episode_returns[e] = clear ? 0.0 : r2;
            w.last_cyber_reward[e] = reward;
            w.has_cyber_reward[e] = clear ? 0 : 1;
            w.rewards[e] = shaped;
            w.truncated[e] = trunc ? 1 : 0;
            w.dones[e] = done ? 1 : 0;
            w.episode_return_out[e] = r2;
            w.episode_length_out[e] = t;
            if (w.executed) w.executed[e] = invalid ? 0 : 1;
            if (clear) {                                  // mcbs_reset for this env (packed batch: mcbs_step.hip's reset tail)
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
            stage(lane).meta() |= (done && A.auto_reset) ? FM_ENDED : 0u;
        }
    }

    // dword d of field f of the env whose LDS record is st (obs_tiny_kernel's values)
    __device__ __forceinline__ uint4 props4(const FusedStage& st, uint32_t r, uint32_t n_disc, bool blank) const {
        if (blank) return make_uint4(2u, 2u, 2u, 2u);
        const uint32_t d = 4u * r, i = fdiv(d, A.dNP), p = d - i * A.NP;
        const uint64_t bits = ((uint64_t)st.props(i) | ((uint64_t)st.props(i + 1u) << A.NP)) >> p;      // rows beyond n_disc are zero
        (void)n_disc;
        return make_uint4((uint32_t)bits & 1u, (uint32_t)(bits >> 1) & 1u, (uint32_t)(bits >> 2) & 1u, (uint32_t)(bits >> 3) & 1u);
    }

    // dword idx of field f (1..4) of the env whose LDS record is st — the same values one at a time, for rows that are not whole vectors
    template <uint32_t f>
    __device__ __forceinline__ uint32_t dword_value(const FusedStage& st, uint32_t idx) const {
        const uint32_t meta = st.meta(), n_creds = (meta >> 8) & 0xFFu;
        const bool blank = meta & FM_BLANK;
        if (f == 3) {
            if (blank) return 2u;
            const uint32_t i = fdiv(idx, A.dNP), p = idx - i * A.NP;
            return (st.props(i) >> p) & 1u;
        }
        if (f == 4) return (uint32_t)st.privb()[idx & 15u];                       // (zero beyond n_disc and for a blank observation: convert_nodes)
        if (f == 2) {
            const uint32_t r = idx >> 1, c = st.cred()[r & 15u];
            return (!blank && r < n_creds) ? ((idx & 1u) ? c >> 8 : c & 0xFFu) : 0u;
        }
        const uint32_t flags = st.flags(), kind = (flags >> F_KIND_SHIFT) & 0xFu, new_creds = (flags >> F_NEWCREDS_SHIFT) & 0x3FFu;
        const uint32_t r = idx >> 2, col = idx & 3u, ci = n_creds - new_creds + r, c = st.cred()[ci & 15u];
        const bool have = !blank && kind == MCBS_OUT_LEAKED_CREDENTIALS && r < new_creds;
        return have ? (col == 0u ? 1u : (col == 1u ? ci : (col == 2u ? c & 0xFFu : c >> 8))) : 0u;
    }

    // ---- fields 1..4: whole 16-byte vectors, each inside one env's row; thread `lane` of `NT` ----
    template <uint32_t f>
    __device__ __forceinline__ void stream_field(uint32_t e0, uint32_t n_env) {
        const uint32_t* fr = fresh_lds();
        {
            if (!A.obs[f]) return;
            if ((A.dwords[f] & 3u) || (f == 3 && A.NP < 4u)) {          // rows that are not whole 16-byte vectors: dword by dword (still coalesced)
                const uint32_t dw = A.dwords[f], total = n_env * dw;
                int32_t* out = A.obs[f] + (size_t)e0 * dw;
                int32_t* tout = A.term[f] ? A.term[f] + (size_t)e0 * dw : nullptr;
                for (uint32_t q = lane; q < total; q += FUSED_THREADS) {
                    const uint32_t env = fdiv(q, A.d_u4[f]), idx = q - env * dw;
                    const FusedStage st = stage(env);
                    const uint32_t meta = st.meta();
                    const uint32_t v = dword_value<f>(st, idx);
                    if (meta & FM_ENDED) { if (meta & FM_LIVE) tout[q] = (int32_t)v; else tout[q] = out[q]; out[q] = (int32_t)fr[A.fresh_off[f] + idx]; }
                    else if (meta & FM_LIVE) out[q] = (int32_t)v;
                }
                return;
            }
            const uint32_t u4 = A.dwords[f] >> 2, total = n_env * u4;
            uint4* out = reinterpret_cast<uint4*>(A.obs[f] + (size_t)e0 * A.dwords[f]);
            uint4* tout = A.term[f] ? reinterpret_cast<uint4*>(A.term[f] + (size_t)e0 * A.dwords[f]) : nullptr;
            const uint4* frow = reinterpret_cast<const uint4*>(fr + A.fresh_off[f]);
            for (uint32_t q = lane; q < total; q += FUSED_THREADS) {
                const uint32_t env = fdiv(q, A.d_u4[f]), r = q - env * u4;
                const FusedStage st = stage(env);
                const uint32_t meta = st.meta(), flags = st.flags();
                const uint32_t n_disc = meta & 0xFFu, n_creds = (meta >> 8) & 0xFFu;
                const bool blank = meta & FM_BLANK;
                uint4 v;
                if (f == 3) v = props4(st, r, n_disc, blank);
                else if (f == 4) {                                    // privilege of the nodes at discovery indices 4r .. 4r + 3: one byte each
                    const uint32_t pv = st.priv4(r & 3u);
                    v = make_uint4(pv & 0xFFu, (pv >> 8) & 0xFFu, (pv >> 16) & 0xFFu, pv >> 24);
                } else if (f == 2) {                                  // cache entries 2r, 2r + 1: (node index, port)
                    const uint16_t* cr = st.cred();
                    const uint32_t c0 = cr[(2u * r) & 15u], c1 = cr[(2u * r + 1u) & 15u];
                    const bool on0 = !blank && 2u * r < n_creds, on1 = !blank && 2u * r + 1u < n_creds;
                    v = make_uint4(on0 ? c0 & 0xFFu : 0u, on0 ? c0 >> 8 : 0u, on1 ? c1 & 0xFFu : 0u, on1 ? c1 >> 8 : 0u);
                } else {                                              // leaked credential r: (1, cache index, node index, port) or zeros (env.py:857-869)
                    const uint32_t kind = (flags >> F_KIND_SHIFT) & 0xFu, new_creds = (flags >> F_NEWCREDS_SHIFT) & 0x3FFu;
                    const bool have = !blank && kind == MCBS_OUT_LEAKED_CREDENTIALS && r < new_creds;
                    const uint32_t ci = n_creds - new_creds + r, c = st.cred()[ci & 15u];
                    v = have ? make_uint4(1u, ci, c & 0xFFu, c >> 8) : make_uint4(0u, 0u, 0u, 0u);
                }
                if (meta & FM_ENDED) { if (meta & FM_LIVE) tout[q] = v; else tout[q] = out[q]; out[q] = frow[r]; }      // the episode's last observation (an intercepted
                                                                                                          // action: the one that stands); the env's next one is the reset observation
                else if (meta & FM_LIVE) out[q] = v;                         // (an intercepted action leaves the env's observation as it was)
            }
        }
    }

    __device__ __forceinline__ void stream(const DevState& S, uint32_t e0) {
        const uint32_t n_env = S.E - e0 < 64u ? S.E - e0 : 64u;       // envs of this wavefront
        const uint32_t* fr = fresh_lds();
        stream_field<1>(e0, n_env); stream_field<2>(e0, n_env); stream_field<3>(e0, n_env); stream_field<4>(e0, n_env);
        // ---- field 0: seven scalars per env, dword by dword ----
        if (A.obs[0]) {
            const uint32_t total = n_env * 7u;
            int32_t* out = A.obs[0] + (size_t)e0 * 7u;
            int32_t* tout = A.term[0] ? A.term[0] + (size_t)e0 * 7u : nullptr;
            for (uint32_t q = lane; q < total; q += FUSED_THREADS) {
                const uint32_t env = fdiv(q, A.d_u4[0]), j = q - env * 7u;
                const FusedStage st = stage(env);
                const uint32_t meta = st.meta(), flags = st.flags();
                const uint32_t n_disc = meta & 0xFFu, n_creds = (meta >> 8) & 0xFFu;
                const bool blank = meta & FM_BLANK;
                const uint32_t kind = (flags >> F_KIND_SHIFT) & 0xFu, level = (flags >> F_LEVEL_SHIFT) & 3u, new_nodes = (flags >> F_NEWNODES_SHIFT) & 0x3FFu;
                int32_t v = 0;
                if (j == 6u) v = (int32_t)n_disc;
                else if (!blank) {
                    if (j == 0u) v = (kind == MCBS_OUT_LEAKED_NODES || kind == MCBS_OUT_LEAKED_CREDENTIALS) ? (int32_t)new_nodes : 0;
                    else if (j == 1u) v = kind == MCBS_OUT_LATERAL_MOVE;
                    else if (j == 2u) v = kind == MCBS_OUT_CUSTOMER_DATA;
                    else if (j == 3u) v = kind == MCBS_OUT_PROBE_SUCCEEDED ? 2 : (kind == MCBS_OUT_PROBE_FAILED ? 1 : 0);
                    else if (j == 4u) v = kind == MCBS_OUT_PRIVILEGE_ESCALATION ? (int32_t)level : 0;
                    else v = (int32_t)n_creds;
                }
                if (meta & FM_ENDED) { if (meta & FM_LIVE) tout[q] = v; else tout[q] = out[q]; out[q] = (int32_t)fr[A.fresh_off[0] + j]; }
                else if (meta & FM_LIVE) out[q] = v;
            }
        }
        // ---- the digests (mcbs_mask_logits): four 16-byte vectors per env ----
        {
            uint4* out = reinterpret_cast<uint4*>(A.digest + e0);
            const uint4* rd = reinterpret_cast<const uint4*>(rdig_lds());
            for (uint32_t q = lane; q < n_env * 4u; q += FUSED_THREADS) {
                const uint32_t env = q >> 2, part = q & 3u;
                const FusedStage st = stage(env);
                const uint32_t meta = st.meta();
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (part == 0u) v.x = st.own();
                else if (part == 2u) v = make_uint4(meta & 0xFFu, (meta >> 8) & 0xFFu, (meta & FM_BLANK) ? 1u : 0u, 0u);
                if (meta & FM_ENDED) out[q] = rd[part];
                else if (meta & FM_LIVE) out[q] = v;
            }
        }
    }
};

template <int DEFK>
__global__ __launch_bounds__(FUSED_THREADS) void wrapper_fused_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, StepIO io, FusedArgs A) {
    __shared__ uint32_t lds[64u * (FUSED_STAGE_DWORDS + FUSED_RAW_DWORDS) + FUSED_FRESH_DWORDS + 32u + 16u];
    const uint32_t tid = threadIdx.x;
    FusedHook hook{A, lds, tid, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, {0, 0, 0, 0, 0}, false};
    // The first wavefront advances the workgroup's 64 envs (lane = env: the headline kernel's code with the hooks above) and leaves their
    // observation records in LDS; the other three meanwhile put the reset observation there.  Then all four stream the observation out:
    // with one wavefront per SIMD every LDS round trip of the streaming loops was exposed (30 us per launch); four hide each other's.
    if (tid < 64u) step_body<0, 0, false, DEFK, false>(S, T, Cp, io, RollArgs{}, hook);
    else hook.fill_reset_rows(tid - 64u, FUSED_THREADS - 64u);
    // LDS-only barriers: the records are complete (lgkmcnt(0)), nobody waits for the global stores in flight (vmcnt untouched)
#define MCBS_LDS_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xC07F); \
                                __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
    MCBS_LDS_BARRIER();
    hook.convert_nodes(S);
    MCBS_LDS_BARRIER();
    hook.convert_creds();
    MCBS_LDS_BARRIER();
#undef MCBS_LDS_BARRIER
    hook.stream(S, blockIdx.x * 64u);
}

} // namespace mcbs
