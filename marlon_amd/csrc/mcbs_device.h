// mcbs_device.h — device-side state layout and topology accessors of the MI355X step engine.
//
// Data layout in HBM (DESIGN.md "State layout"):
//   * header columns, structure-of-arrays with the ENV index fastest, so that lane i of a wavefront
//     (= env base+i) reads consecutive addresses: one 16-byte uint4 {step_count, flags, counts, aux}
//     and one double2 {cum_reward, availability} per env, plus u64 bit-mask columns [word][env] for the
//     node sets (discovered / agent installed / ever owned / running / privilege>=LocalUser) and for the
//     gathered-credential and cached-triple sets;
//   * one "body" record per env (array-of-structures, stride body_stride): the discovery order (u8 node
//     ids) and the credential cache (u16 triple ids) first — their first 16 entries each are fetched with
//     the header, before the action is decoded — then 16-byte node rows {discovered-property mask + privilege_k
//     tags, attacked-ever and attacked-since-reimage slot masks} (packed batches: 4 bytes per row; privilege levels, the
//     running flags and the re-imaging countdown live in the set columns and the re-imaging ring).  Rows are
//     gathered by node id, which differs per env, so they sit next to each other per env rather than
//     along the env axis.
//   * the topology blob (include/mcbs.h "MCBT") is shared by every env and read-only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mcbs.h"

namespace mcbs {

// flags word (h0.y)
constexpr uint32_t F_DONE = 1u, F_TRUNC = 2u, F_OOB = 4u;
constexpr uint32_t F_SKIP = 8u;       // set by the attacker phase of a split step for a skip action, consumed by phase 2
constexpr int F_KIND_SHIFT = 4;      // 4 bits  last outcome kind (MCBS_OUT_*)
constexpr int F_LEVEL_SHIFT = 8;     // 2 bits  last escalation level
constexpr int F_NEWNODES_SHIFT = 12; // 10 bits newly discovered nodes of the last action
constexpr int F_NEWCREDS_SHIFT = 22; // 10 bits credentials newly added to the cache by the last action

struct Row {            // 16 bytes
    uint64_t props_tags; // bits 0..59 discovered properties, bits 60..63 privilege_k tags appended to node.properties (actions.py:378)
    uint32_t ever;      // slot s exploited at least once            (actions.py:396-407)
    uint32_t since;     // ... and not re-imaged since
};
static_assert(sizeof(Row) == 16, "row");
constexpr uint64_t ROW_PROPS_MASK = (1ull << 60) - 1ull;

// "Hot image": the part of the topology the step kernel reads, re-packed for it on the host at topology creation (read through
// L1 / L2; optionally staged in LDS by every workgroup).  One 32-byte record per node and one 64-byte descriptor per (node,
// vulnerability column): the slot_of -> slot indirection of the interchange blob is flattened away, the head of the leak payload
// is inline, and the authorisation table is indexed by the cached credential's TRIPLE id, so that every table address depends on
// the decoded action only (one level of loads).
struct HotNode {        // 32 bytes
    uint64_t props;     // static properties that become known when the node is owned
    int32_t  value;
    uint32_t fw_in_allow, fw_out_allow, listen;
    uint16_t svc_off, svc_cnt;
    uint8_t  flags, pad[3];
};
struct HotDesc {        // 64 bytes; kind == 0xFF: the node does not have this vulnerability
    double   cost;
    uint64_t probe_mask;
    uint32_t payload_off;
    uint16_t payload_cnt, precond_tt;
    uint8_t  kind, level, slot, pad;
    uint32_t pad2;
    mcbs_payload inline_payload[4];   // the first four LeakedCredentials / LeakedNodesId entries, so that a leak needs no load that
                                      // depends on the descriptor (longer lists continue in the payload section)
};
static_assert(sizeof(HotNode) == 32 && sizeof(HotDesc) == 64 && sizeof(mcbs_payload) == 8, "hot records");

// node-set and credential-set columns, [word][env] u64 each
enum { M_DISC = 0, M_INST, M_EVER, M_RUN, M_PLO, M_PHI, M_GATH, M_CACH, M_COUNT };

struct DevState {
    uint4*    h0;       // [E] {step_count, flags, n_discovered | n_creds << 16, owned_count | defender_clock << 16}
    double2*  h1;       // [E] {cum_reward, availability}
    uint32_t* episode;  // [E]
    double*   pending;  // [E] raw reward carried between the split phases of mcbs_step_observe
    uint64_t* masks;    // [M_COUNT][WT][E]: discovered / agent installed / ever owned / running / privilege bit 0 / bit 1 /
                        // gathered credential strings / cached credential triples; every set padded to WT words.
                        // `packed` batches (<= 16 nodes, credential strings and triples: Chain-10, ToyCtf): [E][M_COUNT] u16
                        // instead, i.e. ONE 16-byte word per env holds all eight sets (a quarter of the bytes the step moves)
    // word w of set k of env e, whichever layout the batch uses (kernels off the hot path; the step kernel is specialised)
    __host__ __device__ __forceinline__ uint64_t set_word(const void* base, int k, uint32_t w, uint32_t e) const {
        return packed ? (uint64_t)static_cast<const uint16_t*>(base)[(size_t)e * M_COUNT + (uint32_t)k]
                      : static_cast<const uint64_t*>(base)[((size_t)k * WT + w) * E + e];
    }
    __host__ __device__ __forceinline__ void put_word(void* base, int k, uint32_t w, uint32_t e, uint64_t v) const {
        if (packed) static_cast<uint16_t*>(base)[(size_t)e * M_COUNT + (uint32_t)k] = (uint16_t)v;
        else static_cast<uint64_t*>(base)[((size_t)k * WT + w) * E + e] = v;
    }
    __device__ __forceinline__ uint64_t get(int k, uint32_t w, uint32_t e) const {
        return (wide && k == M_CACH) ? cach[(size_t)w * E + e] : set_word(masks, k, w, e);
    }
    __device__ __forceinline__ void put(int k, uint32_t w, uint32_t e, uint64_t v) const {
        if (wide && k == M_CACH) cach[(size_t)w * E + e] = v;
        else put_word(masks, k, w, e, v);
    }
    __device__ __forceinline__ bool has(int k, uint32_t n, uint32_t e) const { return (get(k, n >> 6, e) >> (n & 63u)) & 1ull; }
    uint64_t* ring;     // [16][NW][E] nodes being re-imaged, by the defender tick (mod 16) that releases them; null without defender
    uint8_t*  body;     // [E][body_stride]
    const uint8_t* init_body; // [body_stride] image of a freshly reset env
    uint32_t E, N, NW, SW, TW, WT;  // WT = words per set = max(NW, SW, TW) rounded to 1, 2 or 4
    uint64_t* cach;     // [TW][E]: the cached-triple set when it needs more than 4 words (`wide`: ActiveDirectory networks cache up to
                        // ~800 (node, port, credential) triples); it then lives here instead of in masks / registers, and the step
                        // kernel stages a lane's words in LDS only while a credential-leak payload is processed
    uint32_t wide;
    uint32_t packed;    // 1: the sets of an env are 16-bit fields of one uint4 (see masks) and its node rows are 4 bytes each:
    uint32_t tiny_p, tiny_v;   // properties (tiny_p bits) | tags (4) | attacked-ever (tiny_v bits) | attacked-since (tiny_v bits) <= 32 bits,
                        // so that the whole body (discovery order 16 B, credential cache 32 B, <= 16 rows 64 B) is fetched with the header
                        // and the step has no load that depends on the action (Chain-10: 14 + 4 + 7 + 7 = 32 bits)
    __host__ __device__ __forceinline__ uint32_t tiny_pack(uint64_t props, uint32_t tags, uint32_t ever, uint32_t since) const {
        return (uint32_t)props | (tags << tiny_p) | (ever << (tiny_p + 4u)) | (tiny_v ? since << (tiny_p + 4u + tiny_v) : 0u);
    }
    // row n of the env whose body starts at body_e, in the canonical 16-byte form, whichever layout the batch uses
    __host__ __device__ __forceinline__ Row row_get(const uint8_t* body_e, uint32_t n) const {
        if (!packed) return reinterpret_cast<const Row*>(body_e + off_rows)[n];
        const uint32_t w = reinterpret_cast<const uint32_t*>(body_e + off_rows)[n], vm = (1u << tiny_v) - 1u;
        Row r;
        r.props_tags = (uint64_t)(w & ((1u << tiny_p) - 1u)) | ((uint64_t)((w >> tiny_p) & 0xFu) << 60);
        r.ever = (w >> (tiny_p + 4u)) & vm;
        r.since = tiny_v ? (w >> (tiny_p + 4u + tiny_v)) & vm : 0u;
        return r;
    }
    __host__ __device__ __forceinline__ void row_put(uint8_t* body_e, uint32_t n, const Row& r) const {
        if (!packed) { reinterpret_cast<Row*>(body_e + off_rows)[n] = r; return; }
        reinterpret_cast<uint32_t*>(body_e + off_rows)[n] =
            tiny_pack(r.props_tags & ((1ull << tiny_p) - 1ull), (uint32_t)(r.props_tags >> 60), r.ever & ((1u << tiny_v) - 1u),
                      r.since & ((1u << tiny_v) - 1u));
    }
    uint32_t body_stride, off_disc, off_cred, off_rows, Cmax;
    uint32_t off_fw;    // body offset of uint16 fw[n_fw_lists]: per-env state of the six manageable rule names in every firewall
                        // rule list (bit r: a rule named r exists, bit 6+r: the first one is ALLOW); MCBS_DEFENDER_EXTERNAL only
};

struct Topo {           // device view of the MCBT blob and of the hot image built from it
    const uint8_t* base;
    const uint8_t* hot;
    __device__ __forceinline__ const mcbs_topo_header& H() const { return *reinterpret_cast<const mcbs_topo_header*>(base); }
};

struct StepCfg {        // the parts of mcbs_batch_cfg the kernels read
    double goal_reward, goal_low_availability, goal_own_atleast_percent;
    double maintain_sla, winning_reward, losing_reward, scan_probability;
    double total_sla_weight, full_availability, full_sum;
    uint64_t seed, env_id_base;
    uint32_t has_attacker_goal, goal_own_atleast, defender_goal_eviction, defender_kind;
    uint32_t scan_capacity, scan_frequency, auto_reset, max_episode_steps, rng_kind, avail_any_order;
    uint32_t L, R, P, V, n_props, K;
    // section offsets into the blob, hoisted so kernels do not chase the header
    uint32_t off_node, off_slot_of, off_slot, off_payload, off_service, off_allowed, off_triple;
    // hot image (Topo::hot) section offsets and size
    uint32_t hot_node, hot_desc, hot_payload, hot_auth, auth_words, hot_triple, hot_avail, hot_bytes;
    uint8_t  rule_port[8];   // identifier-port index of RDP, SSH, HTTPS, HTTP, su, sudo (0xFF: not an identifier port)
    uint32_t n_services, n_fw_lists;
    uint32_t hot_fwlist;     // uint32[N]: incoming list id | outgoing list id << 16
    // ExternalRandomEvents (MCBS_DEFENDER_RANDOM_EVENTS): per-env overlay in the body (offsets from the env's body start) and the
    // static tables it is edited against.  Every node's OWN vulnerability keys, in dictionary order, as identifier columns;
    // a presence mask of those keys; the running bits of its services; every firewall rule LIST as {count, entries[cap]} of
    // u16 (name | allow << 8) at ere_lists[l] = offset | cap << 16 (the lists are objects of their own: DESIGN.md "rule lists").
    uint32_t ere_off_keys, ere_off_kcnt, ere_off_present, ere_off_svc, ere_off_fw, ere_key_cap, ere_n_library;
    uint32_t off_ere;        // blob offset of mcbs_ere_tables
    uint64_t ere_lib_cols;   // columns that are library (global) vulnerabilities: present on every node, always
    const uint32_t* ere_lists;
    uint32_t off_service_cold, off_allowed_cold;   // (aliases of off_service / off_allowed, kept next to their only hot-path user)
    // defender look-ups that would otherwise be table reads AFTER the step's stores (a load behind a store waits for its write
    // acknowledgement): which nodes may be re-imaged, as set words read through the scalar cache, and the availability term when every
    // node has the same one (all reference samples: 1.0) and sums are exact in any order
    uint64_t reimagable[4];
    double   avail_term0;
    uint32_t avail_uniform, pad_u;
    // packed batches: everything an env's re-initialisation writes, as constants the step kernel reads through the scalar cache — the
    // reset image of the body (<= 256 bytes: discovery order 16, credential cache 32, <= 16 four-byte rows, the learned defender's rule-list words), the eight sets as the one
    // uint4, and the number of initially owned nodes — so that the auto-reset of an env that just ended is a handful of STORES by its own
    // lane: no load behind the step's stores, no wave-level copy, no fence (bench.py `headline_with_resets`)
    uint32_t init_image[64];
    uint32_t init_packed[4];
    uint32_t n_init;
    uint32_t goal_own_pct_min;   // AttackerGoal.own_atleast_percent as a count: the smallest k with !(k / N < percent) in the reference's own
                                 // fp64 division (env.py:1093-1095), found on the host (N + 1: never) — the kernels compare integers instead
                                 // of dividing doubles per env and step
    uint32_t init_image_ok;  // 0: the body does not fit init_image (learned-defender / random-events state behind the rows): wave-level copy instead
    uint32_t pad_v[1];
};

// mcbs_rollout_random: the looping step kernel samples each step's action itself; passed as a kernel argument of that variant only
struct RollArgs {
    uint32_t mode = 0;       // 0 off (actions are read), 1 uniform in the action space, 2 the sample_valid_action distribution
    uint32_t nmax = 0, cmax = 0, pad = 0;
    uint64_t seed = 0, step0 = 0;
};

struct StepIO {
    const int32_t* actions;
    float* reward;
    uint8_t* terminated;
    double* availability;
    int32_t* step_count;
    uint8_t* truncated;
    uint8_t* oob;
    float* raw_reward;
    const double* tape;
    uint32_t tape_dps;
    uint32_t n_steps;            // mcbs_step_many: steps in this launch (else unused)
#ifdef MCBS_DIAG
    unsigned long long* stamps;  // diagnostic builds only: [waves][8] s_memtime stamps.  Kept out of the product build: the step
                                 // kernel's arguments must end below byte 256 (see mcbs_api.hip make_io)
#endif
};

struct FastDiv { uint32_t mul, sh1, sh2; };   // n / d for 32-bit n (Granlund-Montgomery round-up form), set up on the host
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv d) {
    const uint32_t t = __umulhi(n, d.mul);
    return (t + ((n - t) >> d.sh1)) >> d.sh2;
}

struct ObsIO {
    int32_t* scalars; int32_t* leaked; int32_t* cache_matrix; int32_t* props; int32_t* priv;
    int8_t* mask_local; int8_t* mask_remote; int8_t* mask_connect; int8_t* mask_discrete;
    uint32_t Nmax, Cmax, K;
    const uint8_t* env_mask; // optional [E]: only envs with a non-zero byte are written
    uint32_t masks_only;   // compute_action_mask: ignore the out-of-bound flag, write only mask fields
    // small action spaces (<= 256 (source, target) pairs): the per-env wavefront of obs_small_kernel also streams the two big
    // masks, so that one launch writes the whole observation and the digest never makes a round trip through memory
    uint32_t fuse_remote;  // mask_remote written by obs_small_kernel: 1 = dwords; 2 = 16-byte chunks of per-source blocks (mcbs_obs.hip stream_blocks)
    uint32_t fuse_connect; // mask_connect written by obs_small_kernel: 1 = 16-byte chunks, row length P*C a multiple of 16; 2 = dwords, any row
                           // length; 3 = 16-byte chunks for row lengths that are NOT a multiple of 16 (ToyCtf: 70), pattern of one lcm period;
                           // 4 = 16-byte chunks of per-source blocks in four LDS variants (stream_blocks)
    uint32_t disc_stride;  // bytes between two envs' rows of mask_discrete (mcbs_set_mask_discrete_stride; the dense length N*N*P*C + N*L + N*N*R by default)
    uint32_t nt_discrete;  // 1: env rows start on 128-byte lines of their own: the connect region's whole chunks use non-temporal stores
    uint32_t disc_remote_blocks;  // 1: ... and so does its remote region
    uint32_t disc_blocks;  // 1: the connect region of mask_discrete (row lengths that are not a multiple of 16) goes through per-source blocks too
    uint32_t blk_region;   // dwords per block variant in LDS (the larger of the connect / remote blocks + 16 bytes, rounded up to 16 bytes); 0: unused
    uint32_t conn_pc;      // fuse_connect == 3: chunks per pattern period, lcm(P*C, 16) / 16
    uint32_t nt_connect;   // 1: the fused connect stream uses non-temporal stores (every env's mask is whole 128-byte lines)
    uint32_t fuse_discrete; // 1: mask_discrete (connect | local | remote per env, 4-byte granularity) written by obs_small_kernel
    // divisors of the observation's index arithmetic (a generic 32-bit division costs ~25 instructions per lane, and the per-env
    // routine had a score of them): properties per node, local / remote ids, Nmax, row length P*C, Cmax, pattern period, chunks per row
    FastDiv dNP, dL, dR, dNm, dRL, dC, dPC, dCPR;
    FastDiv dBLc, dBLr;    // per-source block lengths: Nmax * P * Cmax (connect), Nmax * R (remote)
};

// Local-vulnerability mask of node n as the action mask sees it (env.py:653-659): static, except under ExternalRandomEvents where
// the node's own keys change (library vulnerabilities stay visible on every node)
__device__ __forceinline__ uint32_t local_mask_of(const StepCfg& C, const mcbs_node_static* NS, const uint8_t* body_e, uint32_t n) {
    if (C.defender_kind != MCBS_DEFENDER_RANDOM_EVENTS) return NS[n].local_mask;
    const uint64_t present = reinterpret_cast<const uint64_t*>(body_e + C.ere_off_present)[n] | C.ere_lib_cols;
    return (uint32_t)present & (C.L >= 32u ? ~0u : ((1u << C.L) - 1u));
}

// ------------------------------ Philox4x32-10 (Random123) ------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double to_double53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// Header + mask columns of a freshly reset env (cyberbattle_env.py:375-394, actions.py:149-152): the nodes with
// agent_installed are owned at max(initial privilege, LocalUser), discovered in network order, every node Running,
// no credential gathered, availability 1.0.  The body (rows, discovery order) is copied from init_body by the caller.
__device__ __forceinline__ void reset_header(const DevState& S, const Topo& T, uint32_t e, uint32_t episode) {
    const mcbs_topo_header& H = T.H();
    const uint8_t* order = T.base + H.off_init_order;
    const mcbs_node_static* ns = reinterpret_cast<const mcbs_node_static*>(T.base + H.off_node);
    const uint32_t n_init = H.n_init_owned;
    for (uint32_t w = 0; w < S.NW; ++w) {
        uint64_t m = 0, lo = 0, hi = 0;
        for (uint32_t i = 0; i < n_init; ++i) {
            const uint32_t n = order[i];
            if ((n >> 6) != w) continue;
            const uint64_t bit = 1ull << (n & 63u);
            m |= bit;
            if (ns[n].priv0 & 1u) lo |= bit;
            if (ns[n].priv0 & 2u) hi |= bit;
        }
        const uint32_t rem = S.N - w * 64u;
        S.put(M_DISC, w, e, m); S.put(M_INST, w, e, m); S.put(M_EVER, w, e, m); S.put(M_PLO, w, e, lo); S.put(M_PHI, w, e, hi);
        S.put(M_RUN, w, e, rem >= 64u ? ~0ull : ((1ull << rem) - 1ull));
        if (S.ring) for (uint32_t s = 0; s < 16u; ++s) S.ring[((size_t)s * S.WT + w) * S.E + e] = 0;
    }
    for (uint32_t w = 0; w < S.WT; ++w) { S.put(M_GATH, w, e, 0ull); S.put_word(S.masks, M_CACH, w, e, 0ull); }
    if (S.wide) for (uint32_t w = 0; w < S.TW; ++w) S.cach[(size_t)w * S.E + e] = 0ull;
    for (uint32_t w = S.NW; w < S.WT; ++w)
        for (int k = 0; k < M_GATH; ++k) S.put(k, w, e, 0ull);
    S.h0[e] = make_uint4(0u, 0u, n_init, n_init);
    S.h1[e] = make_double2(0.0, 1.0);
    S.episode[e] = episode;
    S.pending[e] = 0.0;
}

} // namespace mcbs
