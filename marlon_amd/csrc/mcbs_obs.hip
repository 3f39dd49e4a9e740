// mcbs_obs.hip — observation and action-mask kernels (gfx950).
//
// Writes what CyberBattleEnv.__observation_reward_from_action_result / __get_blank_observation /
// __update_action_mask assemble (cyberbattle_env.py:859-933, 753-773, 643-677) directly in the flat layout
// AttackerEnvWrapper.transform_observation hands to Stable-Baselines3 (attack_wrapper.py:474-522) and, for
// mask_discrete, in the order of MaskedDiscreteAttackerWrapper.action_masks (action_masking.py:96-110:
// connect, local, remote).  Pure streaming stores: the state read per env is a few hundred bytes, the
// output up to N*N*P*C bytes, so this tier is bound by HBM write bandwidth.
//
//   obs_small_kernel : ONE WAVEFRONT PER ENV.  The 64 lanes load the env's discovery order together, build the
//                      node -> external-index map in LDS, ballot the "source is owned" bits, and stream out the
//                      small fields with lane-strided (coalesced) stores.  It also leaves a 64-byte digest per
//                      env (owned-source bits, counts, blank flag) for the mask kernel.
//                      For action spaces of up to 256 (source, target) pairs the same wavefront streams the big masks too.
//   obs_scan_kernel  : the same per-env routine for mcbs_observe_masked (the envs a VecEnv just reset): a wavefront scans 64
//                      envs' mask bytes and writes the flagged ones, so a sparse mask costs next to nothing.
//   obs_tiny_kernel  : mask-less observations of topologies with <= 16 nodes: SIXTEEN LANES per env, four envs per wavefront.
//   mask_connect_rows_kernel / mask_fast_kernel / mask_kernel : the big masks of larger action spaces, from the digest alone
//                      (rows of 16-byte chunks with the pattern in LDS; generic 16-byte chunk builder; byte walker).
#pragma once
#include "mcbs_device.h"

namespace mcbs {

// The fused connect-mask stream (obs_env) uses NON-TEMPORAL 16-byte stores when every env's mask is a whole number of 128-byte lines
// (Chain-10: 13 824 B): the gigabyte an observation writes is read back by nobody on this GPU before it falls out of every cache, and
// not allocating it in L2 / Infinity Cache was worth 6-12 % (192 -> 171-186 us for 989 MB; profiles/round2_notes.md).  Where envs share
// cache lines at their boundaries (mask_discrete: 14 172 B per env) non-temporal stores were SLOWER (213 -> 225 us, also when only the
// interior lines used them), as they were for Chain-100's row kernel (35-70 GB per call, -4 %): those keep plain stores.
__device__ __forceinline__ void stream_store16(uint4* p, const uint4 v) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p));
}

struct ObsDigest {          // 64 bytes per env
    uint64_t own_ext[4];    // bit i: the node at external index i has the agent installed
    uint32_t n_disc, n_creds, blank, pad;
    uint32_t pad2[4];
};
static_assert(sizeof(ObsDigest) == 64, "digest");

// Per-wavefront staging area of obs_small_kernel: everything the observation of one env needs, by EXTERNAL index (position in
// the discovery order) for nodes and by cache position for credentials.
struct ObsStage {            // carved out of dynamic shared memory, sized for the topology at launch (obs_stage_bytes)
    uint64_t* props;         // [N]  discovered properties of the node at external index i
    uint32_t* lmask;         // [N]  its local-vulnerability mask (static)
    uint4*    pat;           // [65] fused connect mask: the bytes of one "on" row (+ the first 4 again), <= 1 040
    uint8_t*  ext_of;        // [N]  node id -> external index
    uint8_t*  priv;          // [N]  privilege level of the node at external index i
    uint8_t*  cred_node;     // [T]  node id of cached credential r
    uint8_t*  cred_port;     // [T]  port index of cached credential r
    uint8_t*  onb;           // [260] fused masks: row q = (source, target) is on
    uint32_t* blk;           // [4 x blk_region dwords] per-source block of a fused mask in its four on / off variants (see stream_blocks)
};
__host__ __device__ inline uint32_t obs_stage_bytes(uint32_t n_nodes, uint32_t n_triples, uint32_t blk_region = 0) {   // per wavefront, multiple of 16
    const uint32_t N = (n_nodes + 15u) & ~15u, T = (n_triples + 16u) & ~15u;
    return 8u * N + 4u * N + 1040u + N + N + T + T + 272u + 16u * blk_region;
}
__device__ __forceinline__ ObsStage obs_stage_at(uint8_t* base, uint32_t n_nodes, uint32_t n_triples) {
    const uint32_t N = (n_nodes + 15u) & ~15u, T = (n_triples + 16u) & ~15u;
    ObsStage s;
    s.props = reinterpret_cast<uint64_t*>(base); base += 8u * N;
    s.lmask = reinterpret_cast<uint32_t*>(base); base += 4u * N;
    s.pat = reinterpret_cast<uint4*>(base); base += 1040u;
    s.ext_of = base; base += N;
    s.priv = base; base += N;
    s.cred_node = base; base += T;
    s.cred_port = base; base += T;
    s.onb = base; base += 272u;
    s.blk = reinterpret_cast<uint32_t*>(base);
    return s;
}

__device__ __forceinline__ void obs_env_masks(const StepCfg& C, const ObsIO& O, const uint32_t e, const uint32_t lane, const ObsStage& st,
                                              const uint64_t (&own_ext)[4], const uint32_t n_disc, const uint32_t n_creds, const bool blank);

// Structure: ALL loads first, then ALL stores.  On gfx9 loads and stores share the vmcnt counter and retire in order, so a
// load issued after a store waits for that store's write acknowledgement; a first version that interleaved "load what this
// field needs, store the field" per field spent most of a wavefront's ~25 us lifetime in such waits.  Now lane i fetches
// everything about discovered node i and lane r everything about cached credential r in one burst (three dependent levels:
// header; the two lists; rows / static tables), parks it in LDS by external index, and the rest of the kernel only computes
// from LDS and streams stores.
// The observation of ONE env by one wavefront (e wave-uniform, st = the wavefront's staging area in LDS).
__device__ __forceinline__ void obs_env(const DevState& S, const Topo& T, const StepCfg& C, const ObsIO& O, ObsDigest* digest, const uint32_t e,
                                        const uint32_t lane, const ObsStage& st) {
    // level 1: everything whose address depends on the env index only goes out with the header — entry `lane` of both lists (clamped to
    // the list; meaningful only below the counts the header brings) and, for up to 64 nodes, the one word of the three node sets the
    // observation reads.  The rows / static tables those entries point at are level 2: two dependent round trips per wavefront, not three.
    const uint8_t* body_l1 = S.body + (size_t)e * S.body_stride;
    const uint32_t n_trip_cap = T.H().n_triples;                       // the credential list holds up to n_triples entries (+ one slack slot)
    const uint4 h0 = S.h0[e];
    const uint32_t dl_head = body_l1[S.off_disc + (lane < S.N ? lane : S.N - 1u)];
    const uint32_t cl_head = reinterpret_cast<const uint16_t*>(body_l1 + S.off_cred)[lane < n_trip_cap ? lane : n_trip_cap];
    const bool one_word = S.NW == 1u;
    uint64_t w_inst = 0, w_plo = 0, w_phi = 0;
    if (one_word) { w_inst = S.get(M_INST, 0, e); w_plo = S.get(M_PLO, 0, e); w_phi = S.get(M_PHI, 0, e); }
    const uint32_t flags = h0.y, n_disc = h0.z & 0xFFFFu, n_creds = h0.z >> 16;
    if (!O.masks_only && (flags & F_SKIP)) return;   // split step, skip action: the env's previous observation stands
    const bool blank = !O.masks_only && (flags & F_OOB) != 0;
    const uint32_t kind = (flags >> F_KIND_SHIFT) & 0xFu, level = (flags >> F_LEVEL_SHIFT) & 3u;
    const uint32_t new_nodes = (flags >> F_NEWNODES_SHIFT) & 0x3FFu, new_creds = (flags >> F_NEWCREDS_SHIFT) & 0x3FFu;
    const uint8_t* body = S.body + (size_t)e * S.body_stride;
    const uint8_t* dl = body + S.off_disc;
    const uint16_t* cl = reinterpret_cast<const uint16_t*>(body + S.off_cred);
    const mcbs_node_static* NS = reinterpret_cast<const mcbs_node_static*>(T.base + C.off_node);
    const mcbs_triple* TR = reinterpret_cast<const mcbs_triple*>(T.base + C.off_triple);
    const uint32_t Nm = O.Nmax, NP = C.n_props;

    // ---------------- loads ----------------
    // (the list heads were fetched with the header, before the counts were known: see the top of this function)
    uint64_t own_ext[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
        const uint32_t i = c * 64u + lane;
        if (c * 64u < n_disc) {               // wave-uniform
            bool own = false;
            if (i < n_disc) {
                const uint32_t n = c == 0 ? dl_head : dl[i];
                own = one_word ? (bool)((w_inst >> n) & 1ull) : S.has(M_INST, n, e);
                st.ext_of[n] = (uint8_t)i;
                st.props[i] = S.row_get(body, n).props_tags & ROW_PROPS_MASK;
                st.lmask[i] = local_mask_of(C, NS, body, n);
                st.priv[i] = one_word ? (uint8_t)(((w_plo >> n) & 1ull) | (((w_phi >> n) & 1ull) << 1))
                                      : (uint8_t)((uint32_t)S.has(M_PLO, n, e) | ((uint32_t)S.has(M_PHI, n, e) << 1));
            }
            own_ext[c] = __ballot(own);
        }
    }
    for (uint32_t i = lane; i < n_creds; i += 64u) {
        const mcbs_triple t = TR[i < 64u ? cl_head : cl[i]];
        st.cred_node[i] = (uint8_t)t.node;
        st.cred_port[i] = (uint8_t)t.port;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();

    // ---------------- stores only from here on ----------------
    if (lane == 0) {
        ObsDigest d;
        for (int c = 0; c < 4; ++c) d.own_ext[c] = blank ? 0ull : own_ext[c];
        d.n_disc = n_disc; d.n_creds = n_creds; d.blank = blank ? 1u : 0u; d.pad = 0;
        d.pad2[0] = d.pad2[1] = d.pad2[2] = d.pad2[3] = 0;
        digest[e] = d;
    }

    if (!O.masks_only) {
        if (O.scalars && lane < 7) {
            int32_t v = 0;
            if (lane == 6) v = (int32_t)n_disc;
            else if (!blank) {
                if (lane == 0) v = (kind == MCBS_OUT_LEAKED_NODES || kind == MCBS_OUT_LEAKED_CREDENTIALS) ? (int32_t)new_nodes : 0;
                else if (lane == 1) v = kind == MCBS_OUT_LATERAL_MOVE;
                else if (lane == 2) v = kind == MCBS_OUT_CUSTOMER_DATA;
                else if (lane == 3) v = kind == MCBS_OUT_PROBE_SUCCEEDED ? 2 : (kind == MCBS_OUT_PROBE_FAILED ? 1 : 0);
                else if (lane == 4) v = kind == MCBS_OUT_PRIVILEGE_ESCALATION ? (int32_t)level : 0;
                else v = (int32_t)n_creds;
            }
            O.scalars[(size_t)e * 7 + lane] = v;
        }
        if (O.leaked) {
            int32_t* out = O.leaked + (size_t)e * O.K * 4;
            const bool have = !blank && kind == MCBS_OUT_LEAKED_CREDENTIALS;
            for (uint32_t idx = lane; idx < O.K * 4u; idx += 64u) {
                const uint32_t r = idx >> 2, c = idx & 3u;
                int32_t v = 0;
                if (have && r < new_creds) {
                    const uint32_t ci = n_creds - new_creds + r;
                    v = c == 0 ? 1 : c == 1 ? (int32_t)ci : c == 2 ? (int32_t)st.ext_of[st.cred_node[ci]] : (int32_t)st.cred_port[ci];
                }
                out[idx] = v;
            }
        }
        if (O.cache_matrix) {
            int32_t* out = O.cache_matrix + (size_t)e * O.Cmax * 2;
            for (uint32_t idx = lane; idx < O.Cmax * 2u; idx += 64u) {
                const uint32_t r = idx >> 1;
                int32_t v = 0;
                if (!blank && r < n_creds) v = (idx & 1u) ? (int32_t)st.cred_port[r] : (int32_t)st.ext_of[st.cred_node[r]];
                out[idx] = v;
            }
        }
        if (O.props && NP) {
            int32_t* out = O.props + (size_t)e * Nm * NP;
            uint32_t i = fdiv(lane, O.dNP), p = lane - i * NP;      // (node, property) of flat index `lane`, then advanced by 64
            const uint32_t di = fdiv(64u, O.dNP), dp = 64u - di * NP;
            for (uint32_t idx = lane; idx < Nm * NP; idx += 64u) {
                int32_t v = blank ? 2 : 0;
                if (!blank && i < n_disc) v = (int32_t)((st.props[i] >> p) & 1ull);
                out[idx] = v;
                p += dp; i += di;
                if (p >= NP) { p -= NP; i += 1u; }
            }
        }
        if (O.priv) {
            int32_t* out = O.priv + (size_t)e * Nm;
            for (uint32_t i = lane; i < Nm; i += 64u) out[i] = (!blank && i < n_disc) ? (int32_t)st.priv[i] : 0;
        }
    }
    obs_env_masks(C, O, e, lane, st, own_ext, n_disc, n_creds, blank);
}

// The mask fields of ONE env by one wavefront (e wave-uniform): mask_local and, for small action spaces, the fused big masks.  Reads
// st.lmask (local-vulnerability mask by external index) and uses st.onb / st.pat / st.blk as scratch; own_ext bit i: the node at
// external index i is an owned source (meaningful below n_disc).  Called by obs_env (a wavefront per env) and by obs_quad_kernel
// (four envs staged by 16-lane groups, then streamed one after the other by the whole wavefront).
__device__ __forceinline__ void obs_env_masks(const StepCfg& C, const ObsIO& O, const uint32_t e, const uint32_t lane, const ObsStage& st,
                                              const uint64_t (&own_ext)[4], const uint32_t n_disc, const uint32_t n_creds, const bool blank) {
    const uint32_t Nm = O.Nmax, L = C.L;
    if (O.mask_local) {
        int8_t* out = O.mask_local + (size_t)e * Nm * L;
        for (uint32_t idx = lane; idx < Nm * L; idx += 64u) {
            const uint32_t i = fdiv(idx, O.dL), l = idx - i * L;
            int8_t v = 0;
            if (!blank && i < n_disc && ((own_ext[i >> 6] >> (i & 63u)) & 1ull)) v = (int8_t)((st.lmask[i] >> l) & 1u);
            out[idx] = v;
        }
    }
    if (!(O.fuse_remote | O.fuse_connect | O.fuse_discrete)) return;
    // ---- fused big masks (small action spaces: Nm*Nm <= 256 rows).  Row q = (source s, target t) is "on" when s is an owned
    // discovered node and t a discovered one; a ballot per 64 rows leaves the row bits in scalar registers ----
    const uint32_t rows = Nm * Nm;
    uint64_t on[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        if (k * 64u < rows) {                 // wave-uniform
            const uint32_t q = k * 64u + lane, s = fdiv(q, O.dNm), t = q - s * Nm;
            const bool v = !blank && q < rows && ((own_ext[(s >> 6) & 3u] >> (s & 63u)) & 1ull) && t < n_disc;
            on[k] = __ballot(v);
            st.onb[q] = v ? 1 : 0;
        }
    }
    if (lane < 4u) st.onb[256u + lane] = 0;   // (row `rows` is read, never used, when the last dword ends exactly at the end)
    auto row_on = [&](uint32_t q) -> bool {
        const uint32_t k = q >> 6;
        const uint64_t w = k == 0 ? on[0] : (k == 1 ? on[1] : (k == 2 ? on[2] : on[3]));
        return (w >> (q & 63u)) & 1ull;
    };
    // DEAD SPANS.  What one store instruction of the wavefront covers (a span: 64 lanes x 4 or 16 bytes) touches a handful of
    // consecutive rows, and most rows are off for most of an episode (a few owned sources x the discovered targets): when no row of the
    // span is on — a scalar test of the ballot words — the span is zeros and none of the per-lane pattern work is done.  The SQ
    // counters had these kernels bound by instruction issue, not by HBM (1 200–1 800 VALU instructions per 12–15 KB wavefront;
    // profiles/round2_notes.md section 3c).
    auto any_on = [&](uint32_t qlo, uint32_t qhi) -> bool {       // uniform arguments: is a row in qlo .. qhi (inclusive) on?
        bool any = false;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t lo = k * 64u;
            if (qhi >= lo && qlo <= lo + 63u) {
                const uint32_t a = qlo > lo ? qlo - lo : 0u, b = qhi < lo + 63u ? qhi - lo : 63u;
                const uint64_t m = (b == 63u ? ~0ull : ((1ull << (b + 1u)) - 1ull)) & ~((1ull << a) - 1ull);
                any |= (on[k] & m) != 0ull;
            }
        }
        return any;
    };
    // remote[s][t][r] = on(s, t): 4 bytes per lane and iteration (MR is a multiple of 4), (row, offset in row) advanced incrementally
    auto stream_remote_dwords = [&](uint32_t* out) {
        const uint32_t R = C.R, MR = rows * R;
        if (R >= 4u) {
            const uint32_t dq = fdiv(256u, O.dR), dr = 256u - dq * R, span_rows = fdiv(256u + R - 2u, O.dR) + 1u;
            uint32_t q = fdiv(lane * 4u, O.dR), r = lane * 4u - q * R;
            for (uint32_t i0 = lane * 4u; i0 < MR; i0 += 256u) {
                const uint32_t qf = __builtin_amdgcn_readfirstlane(q), ql = qf + span_rows < 259u ? qf + span_rows : 259u;
                uint32_t v = 0;
                if (any_on(qf, ql)) {
                    const uint32_t nb = R - r;                                // bytes of this dword inside row q (>= 4: all)
                    const uint32_t mq = nb >= 4u ? 0x01010101u : ((1u << (8u * nb)) - 1u) & 0x01010101u;
                    v = (row_on(q) ? mq : 0u) | (row_on(q + 1u) ? (0x01010101u & ~mq) : 0u);
                }
                out[i0 >> 2] = v;
                r += dr; q += dq;
                if (r >= R) { r -= R; q += 1u; }
            }
        } else {
            for (uint32_t i0 = lane * 4u; i0 < MR; i0 += 256u) {
                uint32_t v = 0;
#pragma unroll
                for (uint32_t b = 0; b < 4u; ++b) v |= (uint32_t)row_on(fdiv(i0 + b, O.dR)) << (8u * b);
                out[i0 >> 2] = v;
            }
        }
    };
    // PER-SOURCE BLOCKS.  connect[s][t][p][c] = owned(s) && t < n_disc && c < n_creds and remote[s][t][r] = owned(s) && t < n_disc: the
    // Nm x RL (Nm x R) bytes of one source are the SAME block for every owned source and zeros for the others.  The block is built once per
    // env in LDS in four variants — what a 16-byte chunk starting in source s's block reads depends only on (owned(s), owned(s + 1)):
    // [0] zeros, [1] block ++ 16 zeros, [2] zeros ++ the block's first 16 bytes, [3] block ++ its first 16 bytes — so that a chunk is
    // ONE address computation and one 16-byte LDS read, live or dead, straddling two sources or not (round 2 switched the bytes of the
    // (at most two) ROWS a chunk touches on and off per chunk: ~50 instructions, and ToyCtf's observation was bound by instruction issue).
    auto build_blocks = [&](uint32_t BL, uint32_t RG, auto&& dword_at) {     // dword_at(b): the block's four bytes at offset b (< BL, multiple of 4)
        __builtin_amdgcn_wave_barrier();                   // (an earlier env's / field's blocks may still be in use by other lanes)
        const uint32_t bw = BL >> 2;
        for (uint32_t wd = lane; wd < bw + 4u; wd += 64u) {
            const bool tail = wd >= bw;
            const uint32_t w = dword_at(tail ? (wd - bw) * 4u : wd * 4u);
            st.blk[wd] = 0u;
            st.blk[RG + wd] = tail ? 0u : w;
            st.blk[2u * RG + wd] = tail ? w : 0u;
            st.blk[3u * RG + wd] = w;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    };
    auto stream_blocks = [&](uint4* out, uint32_t total, uint32_t BL, uint32_t RG, const FastDiv& dBL, uint32_t B0) {   // chunk c = mask bytes B0 + 16 c ..
        const uint32_t srcs = blank ? 0u : (uint32_t)own_ext[0];          // bit s: the node at discovery index s is an owned source (Nm <= 16)
        const uint32_t ds = fdiv(1024u, dBL), doff = 1024u - ds * BL;
        uint32_t sidx = fdiv(B0 + lane * 16u, dBL), off = B0 + lane * 16u - sidx * BL;
        for (uint32_t c = lane; c < total; c += 64u) {
            const uint32_t* p = st.blk + ((srcs >> sidx) & 3u) * RG + (off >> 2);
            out[c] = make_uint4(p[0], p[1], p[2], p[3]);
            off += doff; sidx += ds;
            if (off >= BL) { off -= BL; sidx += 1u; }
        }
    };
    if (O.fuse_remote == 2u) {
        const uint32_t R = C.R, BL = Nm * R;
        build_blocks(BL, O.blk_region, [&](uint32_t b) -> uint32_t {
            uint32_t w = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) w |= (uint32_t)(fdiv(b + i, O.dR) < n_disc) << (8u * i);
            return blank ? 0u : w;
        });
        stream_blocks(reinterpret_cast<uint4*>(O.mask_remote + (size_t)e * rows * R), (rows * R) >> 4, BL, O.blk_region, O.dBLr, 0u);
    }
    if (O.fuse_remote == 1u) stream_remote_dwords(reinterpret_cast<uint32_t*>(O.mask_remote + (size_t)e * rows * C.R));
    // General form of the connect region, any row length RL (ToyCtf: 70 bytes), as dwords: the bytes of one "on" row sit in LDS
    // followed by its first four bytes again, so the dword at row offset r is two aligned LDS words shifted by r & 3; the bytes
    // that spill into the next row take that row's on/off.  (row, offset) advance incrementally by 256 bytes per iteration.
    auto stream_connect_dwords = [&](uint32_t* out) {
        const uint32_t Cc = O.Cmax, RL = C.P * Cc, nw = (RL + 4u + 3u) >> 2;
        uint32_t* pat = reinterpret_cast<uint32_t*>(st.pat);
        __builtin_amdgcn_wave_barrier();               // (an earlier pattern may still be in use by other lanes)
        for (uint32_t wd = lane; wd < nw; wd += 64u) {
            uint32_t r = wd * 4u, w = 0;
            if (r >= RL) r -= RL;
            uint32_t c = r - fdiv(r, O.dC) * Cc;
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                w |= (uint32_t)(c < n_creds) << (8u * i);
                r += 1u; c += 1u;
                if (c == Cc) c = 0u;
                if (r == RL) { r = 0u; c = 0u; }
            }
            pat[wd] = w;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        const uint32_t total = (rows * RL) >> 2, dq = fdiv(256u, O.dRL), dr = 256u - dq * RL, span_rows = fdiv(256u + RL - 2u, O.dRL) + 1u;
        uint32_t b0 = lane * 4u, q = fdiv(b0, O.dRL), r = b0 - q * RL;
        for (uint32_t k = lane; k < total; k += 64u) {
            const uint32_t qf = __builtin_amdgcn_readfirstlane(q), ql = qf + span_rows < 259u ? qf + span_rows : 259u;
            uint32_t v = 0;
            if (any_on(qf, ql)) {                                         // (uniform)
                const uint32_t lo = pat[r >> 2], hi = pat[(r >> 2) + 1u];
                const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, r & 3u);
                const uint32_t nb = RL - r;                               // bytes of this dword inside row q (>= 4: all)
                const uint32_t mq = nb >= 4u ? 0xFFFFFFFFu : ((1u << (8u * nb)) - 1u);
                v = w & ((st.onb[q] ? mq : 0u) | (st.onb[q + 1u] ? ~mq : 0u));
            }
            out[k] = v;
            r += dr; q += dq;
            if (r >= RL) { r -= RL; q += 1u; }
        }
    };
    if (O.fuse_connect == 3u) {
        // Row lengths that are not a multiple of 16 (ToyCtf: 7 ports x 10 credentials = 70 bytes), still 16 bytes per lane and store: the
        // byte stream of an all-on env repeats every lcm(RL, 16) bytes = conn_pc chunks (35 for RL = 70), lane j builds chunk j of that period
        // once per env, and chunk c of the env is pattern[c mod conn_pc] with the bytes of the (at most two) rows it touches switched on or
        // off.  (row, offset in row, pattern index) advance incrementally by 64 chunks per iteration.
        const uint32_t Cc = O.Cmax, RL = C.P * Cc, PC = O.conn_pc;
        __builtin_amdgcn_wave_barrier();               // (an earlier pattern may still be in use by other lanes)
        if (lane < PC) {
            uint32_t r = lane * 16u - fdiv(lane * 16u, O.dRL) * RL, c = r - fdiv(r, O.dC) * Cc, w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t i = 0; i < 16u; ++i) {
                w[i >> 2] |= (uint32_t)(c < n_creds) << (8u * (i & 3u));
                r += 1u; c += 1u;
                if (c == Cc) c = 0u;
                if (r == RL) { r = 0u; c = 0u; }
            }
            st.pat[lane] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        uint4* out = reinterpret_cast<uint4*>(O.mask_connect + (size_t)e * rows * RL);
        const uint32_t total = (rows * RL) >> 4, dq = fdiv(1024u, O.dRL), dr = 1024u - dq * RL, dj = 64u - fdiv(64u, O.dPC) * PC,
                       span_rows = fdiv(1024u + RL - 2u, O.dRL) + 1u;
        uint32_t q = fdiv(lane * 16u, O.dRL), r0 = lane * 16u - q * RL, pj = lane - fdiv(lane, O.dPC) * PC;
        for (uint32_t c = lane; c < total; c += 64u) {
            const uint32_t qf = __builtin_amdgcn_readfirstlane(q), ql = qf + span_rows < 259u ? qf + span_rows : 259u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (any_on(qf, ql)) {                                        // (uniform)
                const uint4 p = st.pat[pj];
                const uint32_t b = RL - r0;                              // bytes of this chunk inside row q (>= 16: all)
                const uint32_t a0 = st.onb[q] ? 0xFFFFFFFFu : 0u, a1 = st.onb[q + 1u] ? 0xFFFFFFFFu : 0u;
                uint32_t sel[4];
#pragma unroll
                for (uint32_t d4 = 0; d4 < 4u; ++d4) {
                    const uint32_t lo = 4u * d4;                         // dword d4 holds bytes lo .. lo + 3 of the chunk
                    const uint32_t m = b >= lo + 4u ? 0xFFFFFFFFu : (b <= lo ? 0u : ((1u << (8u * (b - lo))) - 1u));
                    sel[d4] = (a0 & m) | (a1 & ~m);
                }
                v = make_uint4(p.x & sel[0], p.y & sel[1], p.z & sel[2], p.w & sel[3]);
            }
            out[c] = v;
            r0 += dr; q += dq; pj += dj;
            if (r0 >= RL) { r0 -= RL; q += 1u; }
            if (pj >= PC) pj -= PC;
        }
    }
    auto connect_block_dword = [&](uint32_t b) -> uint32_t {      // four bytes of one source's connect block [t][p][c] at offset b
        const uint32_t Cc = O.Cmax, RL = C.P * Cc;
        uint32_t t = fdiv(b, O.dRL), r = b - t * RL, c = r - fdiv(r, O.dC) * Cc, w = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) {
            w |= (uint32_t)(t < n_disc && c < n_creds) << (8u * i);
            r += 1u; c += 1u;
            if (c == Cc) c = 0u;
            if (r == RL) { r = 0u; c = 0u; t += 1u; }
        }
        return blank ? 0u : w;
    };
    if (O.fuse_connect == 4u) {
        const uint32_t RL = C.P * O.Cmax, BL = Nm * RL;
        build_blocks(BL, O.blk_region, connect_block_dword);
        stream_blocks(reinterpret_cast<uint4*>(O.mask_connect + (size_t)e * rows * RL), (rows * RL) >> 4, BL, O.blk_region, O.dBLc, 0u);
    }
    if (O.fuse_connect == 2u) stream_connect_dwords(reinterpret_cast<uint32_t*>(O.mask_connect + (size_t)e * rows * C.P * O.Cmax));
    if (O.fuse_connect == 1u) {
        // connect[s][t][p][c] = on(s, t) && c < n_creds.  An "on" row is RL = P*C bytes of the pattern "n_creds ones, C - n_creds
        // zeros" repeated P times; RL is a multiple of 16 here, so the row is cpr (<= 64) fixed 16-byte chunks: lane j builds
        // chunk j once per env (LDS), then the wavefront streams rows * cpr chunks as `on(q) ? pattern[j] : 0`, one coalesced
        // kilobyte per store instruction, with (row, chunk-in-row) advanced incrementally instead of divided out per chunk.
        const uint32_t Cc = O.Cmax, RL = C.P * Cc, cpr = RL >> 4;
        if (lane < cpr) {
            uint32_t c = lane * 16u - fdiv(lane * 16u, O.dC) * Cc, w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t i = 0; i < 16u; ++i) {
                w[i >> 2] |= (uint32_t)(c < n_creds) << (8u * (i & 3u));
                c = c + 1u == Cc ? 0u : c + 1u;
            }
            st.pat[lane] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        uint4* out = reinterpret_cast<uint4*>(O.mask_connect + (size_t)e * rows * RL);
        const uint32_t total = rows * cpr, dq = fdiv(64u, O.dCPR), dj = 64u - dq * cpr, span_rows = dq + 1u;
        uint32_t q = fdiv(lane, O.dCPR), j = lane - q * cpr;
        for (uint32_t c = lane; c < total; c += 64u) {
            const uint32_t qf = __builtin_amdgcn_readfirstlane(q), ql = qf + span_rows < 259u ? qf + span_rows : 259u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (any_on(qf, ql)) {                                               // (uniform)
                const uint4 p = st.pat[j];
                v = row_on(q) ? p : make_uint4(0, 0, 0, 0);
            }
            if (O.nt_connect) stream_store16(out + c, v); else out[c] = v;      // (uniform)
            j += dj; q += dq;
            if (j >= cpr) { j -= cpr; q += 1u; }
        }
    }
    if (O.fuse_discrete) {
        // MaskedDiscreteAttackerWrapper.action_masks (action_masking.py:96-110): connect | local | remote, one flat int8 vector per
        // env.  Its length is not a multiple of 16 in general (Chain-10: 14 172), so env bases are only 4-byte aligned and the
        // three regions are streamed as dwords.
        const uint32_t RL = C.P * O.Cmax, R = C.R;
        const uint32_t M = rows * RL, ML = Nm * L, MR = rows * R;
        int8_t* base = O.mask_discrete + (size_t)e * O.disc_stride;
        const uint32_t cpr = RL >> 4;
        // value of byte b of the env's flat mask (connect | local | remote), for the few bytes not covered by whole 16-byte chunks
        auto byte_at = [&](uint32_t b) -> uint32_t {
            if (b < M) { const uint32_t q = fdiv(b, O.dRL), r = b - q * RL; return (uint32_t)(row_on(q) && (r - fdiv(r, O.dC) * O.Cmax) < n_creds); }
            b -= M;
            if (b < ML) {
                const uint32_t i = fdiv(b, O.dL), l = b - i * L;
                return (uint32_t)(!blank && i < n_disc && ((own_ext[(i >> 6) & 3u] >> (i & 63u)) & 1ull) && ((st.lmask[i & 255u] >> l) & 1u));
            }
            return (uint32_t)row_on(fdiv(b - ML, O.dR));
        };
        // the remote region of the flat mask ([M + ML, D): 4-byte aligned) as per-source blocks: whole 16-byte chunks from LDS, the few
        // bytes before / after them as dwords
        auto remote_region_blocks = [&]() {
            int8_t* rb = base + M + ML;
            const uint32_t h2 = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(rb) & 15u)) & 15u, BLr = Nm * R;
            build_blocks(BLr, O.blk_region, [&](uint32_t b) -> uint32_t {
                uint32_t w = 0;
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) w |= (uint32_t)(fdiv(b + i, O.dR) < n_disc) << (8u * i);
                return blank ? 0u : w;
            });
            const uint32_t nch = (MR - h2) >> 4, t2 = h2 + (nch << 4);
            if (lane * 4u < h2 || (lane >= 4u && lane < 8u && t2 + (lane - 4u) * 4u < MR)) {
                const uint32_t b0 = lane < 4u ? lane * 4u : t2 + (lane - 4u) * 4u;
                uint32_t v = 0;
#pragma unroll
                for (uint32_t b = 0; b < 4u; ++b) v |= byte_at(M + ML + b0 + b) << (8u * b);
                *reinterpret_cast<uint32_t*>(rb + b0) = v;
            }
            stream_blocks(reinterpret_cast<uint4*>(rb + h2), nch, BLr, O.blk_region, O.dBLr, h2);
        };
        const bool remote_blocks = O.disc_remote_blocks && MR >= 64u;
        if ((RL & 15u) == 0u && cpr <= 64u && M >= 64u) {
            // Row length a multiple of 16 (Chain-10: 96): env bases are only 4-byte aligned (the flat length, 14 172, is not a multiple of
            // 16), so the first 16-byte boundary sits h = 0 / 4 / 8 / 12 bytes into the env.  From there every chunk starts h bytes into a
            // 16-byte column of its row: lane j builds the SHIFTED pattern chunk j (bytes h + 16 j ... of the row followed by itself), and
            // only the last chunk of a row straddles into the next one (its upper h bytes take that row's on / off).
            const uint32_t h = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(base) & 15u)) & 15u, Cc = O.Cmax;
            __builtin_amdgcn_wave_barrier();               // (an earlier pattern may still be in use by other lanes)
            if (lane < cpr) {
                uint32_t r = h + lane * 16u, w[4] = {0, 0, 0, 0};
                if (r >= RL) r -= RL;
                uint32_t c = r - fdiv(r, O.dC) * Cc;
#pragma unroll
                for (uint32_t i = 0; i < 16u; ++i) {
                    w[i >> 2] |= (uint32_t)(c < n_creds) << (8u * (i & 3u));
                    r += 1u; c += 1u;
                    if (c == Cc) c = 0u;
                    if (r == RL) { r = 0u; c = 0u; }
                }
                st.pat[lane] = make_uint4(w[0], w[1], w[2], w[3]);
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            if (lane * 4u < h) {                             // head: the h bytes before the first aligned chunk (row 0)
                uint32_t v = 0;
#pragma unroll
                for (uint32_t b = 0; b < 4u; ++b) v |= byte_at(lane * 4u + b) << (8u * b);
                reinterpret_cast<uint32_t*>(base)[lane] = v;
            }
            const uint32_t nchunks = (M - h) >> 4, dq = fdiv(64u, O.dCPR), dj = 64u - dq * cpr;
            uint4* out16 = reinterpret_cast<uint4*>(base + h);

            uint32_t q = fdiv(h + lane * 16u, O.dRL), j = ((h + lane * 16u) - q * RL - h) >> 4;
            // bytes [0, 16 - h) of a straddling chunk belong to row q, the rest to row q + 1
            const uint32_t nb = 16u - h;
            uint32_t keep[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) keep[k] = nb >= 4u * (k + 1u) ? 0xFFFFFFFFu : (nb <= 4u * k ? 0u : ((1u << (8u * (nb - 4u * k))) - 1u));
            const uint32_t span_rows = dq + 2u;
            for (uint32_t c = lane; c < nchunks; c += 64u) {
                const uint32_t qf = __builtin_amdgcn_readfirstlane(q), ql = qf + span_rows < 259u ? qf + span_rows : 259u;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (any_on(qf, ql)) {                                    // (uniform)
                    const uint4 p = st.pat[j];
                    const uint32_t a = row_on(q) ? 0xFFFFFFFFu : 0u;
                    v = make_uint4(p.x & a, p.y & a, p.z & a, p.w & a);
                    if (h && j == cpr - 1u) {
                        const uint32_t nx = row_on(q + 1u) ? 0xFFFFFFFFu : 0u;
                        v = make_uint4(p.x & ((a & keep[0]) | (nx & ~keep[0])), p.y & ((a & keep[1]) | (nx & ~keep[1])),
                                       p.z & ((a & keep[2]) | (nx & ~keep[2])), p.w & ((a & keep[3]) | (nx & ~keep[3])));
                    }
                }
                if (O.nt_discrete) stream_store16(out16 + c, v); else out16[c] = v;      // (uniform)
                j += dj; q += dq;
                if (j >= cpr) { j -= cpr; q += 1u; }
            }
            const uint32_t t0 = h + (nchunks << 4), D = remote_blocks ? M + ML : M + ML + MR;    // tail: the rest of connect (< 16 bytes), local, remote
            if (remote_blocks) remote_region_blocks();
            for (uint32_t b0 = t0 + lane * 4u; b0 < D; b0 += 256u) {
                uint32_t v = 0;
#pragma unroll
                for (uint32_t b = 0; b < 4u; ++b) v |= byte_at(b0 + b) << (8u * b);
                *reinterpret_cast<uint32_t*>(base + b0) = v;
            }
        } else {
        uint32_t* out = reinterpret_cast<uint32_t*>(base);
        if (O.disc_blocks && M >= 64u) {
            // connect rows that are not a multiple of 16 bytes (ToyCtf: 70): per-source blocks (above) for every whole 16-byte chunk of the
            // region — env bases are only 4-byte aligned, the first chunk starts h bytes in —, dwords for the few bytes around them
            const uint32_t h = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(base) & 15u)) & 15u, BL = Nm * RL;
            build_blocks(BL, O.blk_region, connect_block_dword);
            const uint32_t nch = (M - h) >> 4, t0 = h + (nch << 4);
            if (lane * 4u < h || (t0 + (lane - 4u) * 4u < M && lane >= 4u && lane < 8u)) {      // lanes 0..3: head dwords, lanes 4..7: tail dwords
                const uint32_t b0 = lane < 4u ? lane * 4u : t0 + (lane - 4u) * 4u;
                uint32_t v = 0;
#pragma unroll
                for (uint32_t b = 0; b < 4u; ++b) v |= byte_at(b0 + b) << (8u * b);
                *reinterpret_cast<uint32_t*>(base + b0) = v;
            }
            stream_blocks(reinterpret_cast<uint4*>(base + h), nch, BL, O.blk_region, O.dBLc, h);
        } else stream_connect_dwords(out);
        out += M >> 2;
        for (uint32_t i0 = lane * 4u; i0 < ML; i0 += 256u) {      // local[i][l], same rule as mask_local
            uint32_t v = 0;
#pragma unroll
            for (uint32_t b = 0; b < 4u; ++b) v |= byte_at(M + i0 + b) << (8u * b);
            out[i0 >> 2] = v;
        }
        out += ML >> 2;
        if (remote_blocks) remote_region_blocks(); else stream_remote_dwords(out);
        }
    }
}

// Small topologies, no mask field requested (a policy that applies the action mask to its logits with mcbs_mask_logits): the observation
// is ~1 KB per env and a wavefront per env spends its life waiting on two dependent loads for 64 B-sized jobs — the launch is bound by
// wavefront turnover (45 us for 65 536 Chain-10 envs).  Here SIXTEEN LANES serve an env (four envs per wavefront): lane j fetches
// everything about discovered node j and cached credential j, the 16-lane group stages it in its own 192 bytes of LDS, and streams the
// fields out.  Same values as obs_env, field by field (tests/test_gpu_vecenv.py, test_gpu_logits.py).  Needs at most 16 nodes / cached
// credentials per env and a single set word (NW == 1).
struct TinyStage { uint64_t props[16]; uint8_t priv[16], ext_of[16], cred_ext[16], cred_port[16]; };   // 192 bytes per env
static_assert(sizeof(TinyStage) == 192, "tiny stage");

struct TinyOut { uint32_t own16, n_disc, n_creds; bool blank, live; };   // per lane: what its env's mask fields need (obs_quad_kernel)

// The small fields of the env of one 16-lane group (j = lane in the group, st = the group's staging area).  LMASK: also leave the
// local-vulnerability mask of the node at external index j in lmask[j] (the mask fields' input).  Every lane of the wavefront runs to
// the end (wave-level operations inside; obs_quad_kernel goes on with the masks).
template <bool LMASK>
__device__ __forceinline__ TinyOut obs_tiny_group(const DevState& S, const Topo& T, const StepCfg& C, const ObsIO& O, ObsDigest* digest, const uint32_t e,
                                                  const uint32_t j, const uint32_t lane, TinyStage& st, uint32_t* lmask) {
    const bool valid = e < S.E;
    const uint32_t ec = valid ? e : 0u;
    const uint8_t* body = S.body + (size_t)ec * S.body_stride;
    const uint32_t n_trip_cap = T.H().n_triples;
    // level 1
    const uint4 h0 = S.h0[ec];
    const uint32_t n = body[S.off_disc + (j < S.N ? j : S.N - 1u)];
    const uint32_t tid = reinterpret_cast<const uint16_t*>(body + S.off_cred)[j < n_trip_cap ? j : n_trip_cap];
    const uint64_t w_inst = S.get(M_INST, 0, ec), w_plo = S.get(M_PLO, 0, ec), w_phi = S.get(M_PHI, 0, ec);
    const uint32_t flags = h0.y, n_disc = h0.z & 0xFFFFu, n_creds = h0.z >> 16;
    const bool live = valid && !(flags & F_SKIP);             // split step, skip action: the env's previous observation stands
    const bool blank = (flags & F_OOB) != 0;
    const uint32_t kind = (flags >> F_KIND_SHIFT) & 0xFu, level = (flags >> F_LEVEL_SHIFT) & 3u;
    const uint32_t new_nodes = (flags >> F_NEWNODES_SHIFT) & 0x3FFu, new_creds = (flags >> F_NEWCREDS_SHIFT) & 0x3FFu;
    // level 2
    const mcbs_triple* TR = reinterpret_cast<const mcbs_triple*>(T.base + C.off_triple);
    const bool is_node = j < n_disc, is_cred = j < n_creds;
    const uint64_t props = S.row_get(body, n).props_tags & ROW_PROPS_MASK;
    if constexpr (LMASK) lmask[j] = is_node ? local_mask_of(C, reinterpret_cast<const mcbs_node_static*>(T.base + C.off_node), body, n) : 0u;
    const mcbs_triple tr = TR[is_cred ? tid : 0u];
    const bool own = is_node && ((w_inst >> n) & 1ull);
    const uint32_t own16 = (uint32_t)(__ballot(own) >> (lane & 48u)) & 0xFFFFu;      // this env's owned-source bits by external index
    if (is_node) {
        st.ext_of[n & 15u] = (uint8_t)j;
        st.props[j] = props;
        st.priv[j] = (uint8_t)(((w_plo >> n) & 1ull) | (((w_phi >> n) & 1ull) << 1));
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (is_cred) { st.cred_ext[j] = st.ext_of[tr.node & 15u]; st.cred_port[j] = (uint8_t)tr.port; }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    TinyOut res;
    res.own16 = own16; res.n_disc = n_disc; res.n_creds = n_creds; res.blank = blank; res.live = live;
    if (!live) return res;                                     // (after the wave-level operations: the four envs of a wavefront differ)
    // ---------------- stores ----------------
    if (j == 0) {
        ObsDigest d;
        d.own_ext[0] = blank ? 0ull : (uint64_t)own16; d.own_ext[1] = d.own_ext[2] = d.own_ext[3] = 0ull;
        d.n_disc = n_disc; d.n_creds = n_creds; d.blank = blank ? 1u : 0u; d.pad = 0;
        d.pad2[0] = d.pad2[1] = d.pad2[2] = d.pad2[3] = 0;
        digest[e] = d;
    }
    const uint32_t Nm = O.Nmax, NP = C.n_props;
    if (O.scalars && j < 7u) {
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
        O.scalars[(size_t)e * 7 + j] = v;
    }
    // One ROW per lane from here on (a leaked-credential record, a cache entry, a node's properties): what a row needs is in the lane's
    // own registers, a dword-per-lane walk over the flat arrays cost ~12 instructions per dword (LDS read, 64-bit shift, index carry) and
    // made this kernel issue-bound (16.5 us for 924 B per env; rows: see profiles/round2_notes.md).
    if (O.leaked) {
        const bool have = !blank && kind == MCBS_OUT_LEAKED_CREDENTIALS;
        for (uint32_t r = j; r < O.K; r += 16u) {                // (1, cache index, external node index, port) or zeros (env.py:857-869)
            int32_t* out = O.leaked + ((size_t)e * O.K + r) * 4;
            int4 v = make_int4(0, 0, 0, 0);
            if (have && r < new_creds) {
                const uint32_t ci = n_creds - new_creds + r;
                v = make_int4(1, (int32_t)ci, (int32_t)st.cred_ext[ci & 15u], (int32_t)st.cred_port[ci & 15u]);
            }
            if ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) *reinterpret_cast<int4*>(out) = v;
            else { out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w; }
        }
    }
    if (O.cache_matrix) {
        for (uint32_t r = j; r < O.Cmax; r += 16u) {             // (external node index, port) of cache entry r
            int32_t* out = O.cache_matrix + ((size_t)e * O.Cmax + r) * 2;
            const bool on = !blank && r < n_creds;
            const int2 v = make_int2(on ? (int32_t)st.cred_ext[r & 15u] : 0, on ? (int32_t)st.cred_port[r & 15u] : 0);
            if ((reinterpret_cast<uintptr_t>(out) & 7u) == 0) *reinterpret_cast<int2*>(out) = v;
            else { out[0] = v.x; out[1] = v.y; }
        }
    }
    if (O.props && NP) {
        // [Nm][NP] dwords, 0 / 1 per property bit of a discovered node, 0 for the others, all 2 for a blank observation.  Sixteen bytes
        // per lane and step, the env's 16 lanes side by side (a lane per node ROW was tried: 56-byte lane strides, twice as slow): the
        // four dwords of a vector belong to node i or i + 1, whose property words are joined into one bit string.
        int32_t* out = O.props + (size_t)e * Nm * NP;
        const uint32_t total = Nm * NP;
        if (NP >= 4u && (total & 3u) == 0u && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
            const uint32_t di = fdiv(64u, O.dNP), dp = 64u - di * NP;    // one step = 16 lanes x 4 dwords
            uint32_t i = fdiv(4u * j, O.dNP), p = 4u * j - i * NP;
            const uint64_t row_mask = (1ull << NP) - 1ull;       // NP <= 60
            for (uint32_t q = j; q < total / 4u; q += 16u) {
                int4 v = make_int4(2, 2, 2, 2);
                if (!blank) {
                    const uint64_t b0 = i < n_disc ? (st.props[i & 15u] & row_mask) : 0ull, b1 = i + 1u < n_disc ? st.props[(i + 1u) & 15u] : 0ull;
                    const uint32_t nib = (uint32_t)(((b0 | (b1 << NP)) >> p) & 0xFull);       // bits p .. p + 3 <= NP + 2 <= 62
                    v = make_int4((int32_t)(nib & 1u), (int32_t)((nib >> 1) & 1u), (int32_t)((nib >> 2) & 1u), (int32_t)(nib >> 3));
                }
                reinterpret_cast<int4*>(out)[q] = v;
                p += dp; i += di;
                if (p >= NP) { p -= NP; i += 1u; }
            }
        } else {
            uint32_t i = fdiv(j, O.dNP), p = j - i * NP;
            const uint32_t di = fdiv(16u, O.dNP), dp = 16u - di * NP;
            for (uint32_t idx = j; idx < total; idx += 16u) {
                int32_t v = blank ? 2 : 0;
                if (!blank && i < n_disc) v = (int32_t)((st.props[i & 15u] >> p) & 1ull);
                out[idx] = v;
                p += dp; i += di;
                while (p >= NP) { p -= NP; i += 1u; }
            }
        }
    }
    if (O.priv && j < Nm) O.priv[(size_t)e * Nm + j] = (!blank && j < n_disc) ? (int32_t)st.priv[j] : 0;
    return res;
}

__global__ __launch_bounds__(256) void obs_tiny_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, ObsIO O, ObsDigest* digest) {
    const StepCfg& C = *Cp;
    __shared__ TinyStage stage[16];
    const uint32_t grp = threadIdx.x >> 4, j = threadIdx.x & 15u, lane = threadIdx.x & 63u;
    obs_tiny_group<false>(S, T, C, O, digest, blockIdx.x * 16u + grp, j, lane, stage[grp], nullptr);
}

// Small topologies WITH mask fields: the small fields as in obs_tiny_kernel — four envs per wavefront, sixteen lanes each, so the loads
// and the dozen small set-up jobs of an env are paid once per FOUR envs —, then the whole wavefront streams the mask fields of its four
// envs one after the other (obs_env_masks, the same writers as a wavefront per env).  obs_small_kernel spent ~1 250 vector instructions
// per 12-15 KB env, half of them in per-env set-up that kept 16 of 64 lanes busy, and was bound by instruction issue, not by HBM
// (profiles/round3_notes.md section 7).
__global__ __launch_bounds__(256) void obs_quad_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, ObsIO O, ObsDigest* digest) {
    const StepCfg& C = *Cp;
    extern __shared__ uint4 obs_lds[];                         // per wavefront: onb [272] | pat [1 040] | blk [16 x blk_region]
    __shared__ TinyStage stage[16];
    __shared__ uint32_t lmask[16][16];
    const uint32_t grp = threadIdx.x >> 4, j = threadIdx.x & 15u, lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const TinyOut r = obs_tiny_group<true>(S, T, C, O, digest, blockIdx.x * 16u + grp, j, lane, stage[grp], lmask[grp]);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    uint8_t* scratch = reinterpret_cast<uint8_t*>(obs_lds) + wave * (272u + 1040u + 16u * O.blk_region);
    ObsStage st{};
    st.onb = scratch; st.pat = reinterpret_cast<uint4*>(scratch + 272u); st.blk = reinterpret_cast<uint32_t*>(scratch + 272u + 1040u);
    const uint32_t packed = r.own16 | (r.n_disc << 16) | (r.n_creds << 21) | ((uint32_t)r.blank << 26) | ((uint32_t)r.live << 27);   // counts <= 16
#pragma unroll 1
    for (uint32_t g = 0; g < 4u; ++g) {
        const uint32_t d = __builtin_amdgcn_readlane(packed, g * 16u);      // env g of this wavefront, as scalars
        if (!((d >> 27) & 1u)) continue;                                    // past the batch, or a skip action: the previous observation stands
        const uint64_t own_ext[4] = {(uint64_t)(d & 0xFFFFu), 0ull, 0ull, 0ull};
        st.lmask = lmask[wave * 4u + g];
        // (the writers' per-lane set-up is recomputed per env on purpose: hoisted out of this loop it held 147 VGPRs, three wavefronts per SIMD)
        uint32_t lane_g = lane;
        asm volatile("" : "+v"(lane_g));
        obs_env_masks(C, O, blockIdx.x * 16u + wave * 4u + g, lane_g, st, own_ext, (d >> 16) & 31u, (d >> 21) & 31u, ((d >> 26) & 1u) != 0u);
        __builtin_amdgcn_wave_barrier();                                    // the next env reuses the scratch area
    }
}


// One wavefront per env (every env, or the env's byte in env_mask decides: callers with sparse masks use obs_scan_kernel instead).
__global__ __launch_bounds__(256) void obs_small_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, ObsIO O, ObsDigest* digest) {
    const StepCfg& C = *Cp;   // device copy: by value it would push the arguments past 256 bytes (profiles/round1_notes.md)
    extern __shared__ uint4 obs_lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;   // wave index as a scalar
    const uint32_t e = blockIdx.x * 4u + wave;
    if (e >= S.E) return;                     // whole wavefront leaves together
    if (O.env_mask && !O.env_mask[e]) return; // wave-uniform: one wavefront per env
    const uint32_t n_triples_all = T.H().n_triples;
    const ObsStage st = obs_stage_at(reinterpret_cast<uint8_t*>(obs_lds) + wave * obs_stage_bytes(S.N, n_triples_all, O.blk_region), S.N, n_triples_all);
    obs_env(S, T, C, O, digest, e, lane, st);
}

// mcbs_observe_masked (the reset observation of the envs a VecEnv just reset): a wavefront scans the mask bytes of 64 envs (one
// coalesced load + ballot) and writes the flagged envs' observations one after the other — E / 64 wavefronts in all, so the launch
// costs a few microseconds when no env was reset, instead of E wavefronts that each look at one byte and leave.
__global__ __launch_bounds__(256) void obs_scan_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, ObsIO O, ObsDigest* digest) {
    const StepCfg& C = *Cp;
    extern __shared__ uint4 obs_lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t e0 = (blockIdx.x * 4u + wave) * 64u;
    if (e0 >= S.E) return;
    const uint32_t n_triples_all = T.H().n_triples;
    const ObsStage st = obs_stage_at(reinterpret_cast<uint8_t*>(obs_lds) + wave * obs_stage_bytes(S.N, n_triples_all, O.blk_region), S.N, n_triples_all);
    uint64_t m = __ballot(e0 + lane < S.E && O.env_mask[e0 + lane] != 0);
    while (m) {
        const uint32_t e = e0 + (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        obs_env(S, T, C, O, digest, e, lane, st);
        __builtin_amdgcn_wave_barrier();      // the next env reuses the staging area
        __threadfence_block();
    }
}

// REGION 0: connect [N,N,P,C]   1: remote [N,N,R]   2: local [N,L]
template <int W, int REGION>
__global__ __launch_bounds__(256) void mask_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, const ObsDigest* digest, int8_t* dst,
                                                   size_t env_stride, size_t region_off, uint32_t Nm, uint32_t Cm, const uint8_t* env_mask,
                                                   uint32_t skip_flagged) {
    const StepCfg& C = *Cp;
    const uint32_t e = blockIdx.x;
    if (env_mask && !env_mask[e]) return;      // uniform per workgroup
    if (skip_flagged && (S.h0[e].y & F_SKIP)) return;
    const uint32_t k = blockIdx.y * blockDim.x + threadIdx.x;
    const uint32_t inner = REGION == 0 ? C.P * Cm : (REGION == 1 ? C.R : C.L);
    const uint32_t len = REGION == 2 ? Nm * inner : Nm * Nm * inner;
    const uint32_t idx0 = k * W;
    if (idx0 >= len) return;
    const ObsDigest d = digest[e];
    uint32_t q = idx0 / inner, in = idx0 - q * inner;     // q = s*N + t (or s for local), in = position inside the row
    uint32_t s, t;
    if (REGION == 2) { s = q; t = 0; } else { s = q / Nm; t = q - s * Nm; }
    const uint8_t* dl = S.body + (size_t)e * S.body_stride + S.off_disc;
    const mcbs_node_static* NS = reinterpret_cast<const mcbs_node_static*>(T.base + C.off_node);
    alignas(16) int8_t v[W];
    uint32_t lmask = 0;
    bool row_on = false;
    auto load_row = [&]() {
        row_on = s < Nm && ((d.own_ext[(s >> 6) & 3u] >> (s & 63u)) & 1ull) && (REGION == 2 || t < d.n_disc);
        if (REGION == 2 && row_on) lmask = local_mask_of(C, NS, S.body + (size_t)e * S.body_stride, dl[s]);
    };
    load_row();
    uint32_t c = REGION == 0 ? in % Cm : 0u;              // credential index inside the (source, target, port) row
#pragma unroll
    for (int j = 0; j < W; ++j) {
        int8_t b = 0;
        if (idx0 + j < len && row_on) {
            if (REGION == 0) b = c < d.n_creds;
            else if (REGION == 1) b = 1;
            else b = (int8_t)((lmask >> in) & 1u);
        }
        v[j] = b;
        if (REGION == 0 && ++c == Cm) c = 0;
        if (++in == inner) {
            in = 0;
            if (REGION == 2) s += 1; else if (++t == Nm) { t = 0; s += 1; }
            load_row();
        }
    }
    int8_t* out = dst + (size_t)e * env_stride + region_off + idx0;
    if (W == 16) *reinterpret_cast<uint4*>(out) = *reinterpret_cast<const uint4*>(v);
    else if (W == 4) *reinterpret_cast<uint32_t*>(out) = *reinterpret_cast<const uint32_t*>(v);
    else out[0] = v[0];
}

// ---- fast path of the two big masks (16-byte stores) -------------------------------------------------------------
// Both masks are periodic: connect[s][t][p][c] = own(s) && t < n_disc && c < n_creds is, per (s,t) row of RL = P*C
// bytes, the pattern "n_creds ones, C - n_creds zeros" repeated; because every row length is a multiple of C the
// pattern depends only on (byte index mod C) across the whole env, and rows only switch it on or off.
// remote[s][t][r] = own(s) && t < n_disc is, per source row of RL = N*R bytes, "n_disc*R ones then zeros" (C = RL).
// A thread therefore builds its 16 bytes from two byte-range masks instead of walking them one by one.
__device__ __forceinline__ void ones_upto(uint32_t k, uint64_t& lo, uint64_t& hi) {   // bytes [0,k) = 0xFF, k in 0..16
    lo = k >= 8u ? ~0ull : (k ? (~0ull >> (64u - 8u * k)) : 0ull);
    hi = k <= 8u ? 0ull : (k >= 16u ? ~0ull : (~0ull >> (64u - 8u * (k - 8u))));
}

// 16 bytes of one env's periodic mask starting at byte idx0 (multiple of 16)
template <int REGION>
__device__ __forceinline__ uint4 mask_chunk(const ObsDigest& d, uint32_t ones, uint32_t idx0, uint32_t RL, uint32_t Cc, uint32_t Nm,
                                            FastDiv dRL, FastDiv dC, FastDiv dN) {
    const uint32_t q0 = fdiv(idx0, dRL), r0 = idx0 - q0 * RL;
    uint32_t c = REGION == 0 ? r0 - fdiv(r0, dC) * Cc : r0;
    auto row_on = [&](uint32_t q) -> bool {
        uint32_t s = q, t = 0;
        if (REGION == 0) { s = fdiv(q, dN); t = q - s * Nm; }
        return s < Nm && ((d.own_ext[(s >> 6) & 3u] >> (s & 63u)) & 1ull) && (REGION == 1 || t < d.n_disc);
    };
    uint64_t plo = 0, phi = 0;                 // which of the 16 bytes are 1 inside an "on" row
    for (uint32_t pos = 0; pos < 16u;) {
        const uint32_t seg = min(16u - pos, Cc - c);
        const uint32_t k = c < ones ? min(ones - c, seg) : 0u;
        uint64_t alo, ahi, blo, bhi;
        ones_upto(pos + k, alo, ahi);
        ones_upto(pos, blo, bhi);
        plo |= alo & ~blo; phi |= ahi & ~bhi;
        pos += seg; c = 0;
    }
    const uint32_t b = min(16u, RL - r0);      // bytes of row q0 in this chunk; the rest belongs to row q0 + 1 (RL >= 16)
    uint64_t flo, fhi;
    ones_upto(b, flo, fhi);
    const bool on0 = row_on(q0), on1 = b < 16u && row_on(q0 + 1u);
    const uint64_t mlo = (on0 ? flo : 0ull) | (on1 ? ~flo : 0ull), mhi = (on0 ? fhi : 0ull) | (on1 ? ~fhi : 0ull);
    const uint64_t lo = plo & mlo & 0x0101010101010101ull, hi = phi & mhi & 0x0101010101010101ull;
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}

// FLAT = false: blockIdx.y strides over envs (digest and flags read once per env), threads stride over the env's chunks.
// FLAT = true : regions shorter than a workgroup's reach (remote: N*N*R bytes): one flat (env, chunk) index.
// Either way a few thousand workgroups stream the whole [E, len] array.
template <int REGION, bool FLAT>   // REGION 0 connect, 1 remote
__global__ __launch_bounds__(256) void mask_fast_kernel(DevState S, const ObsDigest* digest, int8_t* dst, size_t env_stride,
                                                        size_t region_off, uint32_t len, uint32_t RL, uint32_t Cc, uint32_t Nm,
                                                        uint32_t Rr, FastDiv dRL, FastDiv dC, FastDiv dN, const uint8_t* env_mask,
                                                        uint32_t skip_flagged, FastDiv dCPE) {
    const uint32_t cpe = len / 16u;          // 16-byte chunks per env
    if (FLAT) {
        const uint32_t nthreads = gridDim.x * gridDim.y * blockDim.x, total = S.E * cpe;
        for (uint32_t g = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; g < total; g += nthreads) {
            const uint32_t e = fdiv(g, dCPE), idx0 = (g - e * cpe) * 16u;
            if (env_mask && !env_mask[e]) continue;
            if (skip_flagged && (S.h0[e].y & F_SKIP)) continue;
            const ObsDigest d = digest[e];
            const uint32_t ones = REGION == 0 ? d.n_creds : d.n_disc * Rr;
            *reinterpret_cast<uint4*>(dst + (size_t)e * env_stride + region_off + idx0) = mask_chunk<REGION>(d, ones, idx0, RL, Cc, Nm, dRL, dC, dN);
        }
    } else {
        for (uint32_t e = blockIdx.y; e < S.E; e += gridDim.y) {
            if (env_mask && !env_mask[e]) continue;
            if (skip_flagged && (S.h0[e].y & F_SKIP)) continue;
            const ObsDigest d = digest[e];
            const uint32_t ones = REGION == 0 ? d.n_creds : d.n_disc * Rr;     // leading ones of one period
            int8_t* out = dst + (size_t)e * env_stride + region_off;
            for (uint32_t idx0 = (blockIdx.x * blockDim.x + threadIdx.x) * 16u; idx0 < len; idx0 += gridDim.x * blockDim.x * 16u)
                *reinterpret_cast<uint4*>(out + idx0) = mask_chunk<REGION>(d, ones, idx0, RL, Cc, Nm, dRL, dC, dN);
        }
    }
}

// Connect mask for large action spaces (row length a multiple of 16: Chain-100's 816 bytes).  Same idea as the fused wavefront,
// one workgroup per (env, range of rows): the 16-byte chunks of an "on" row and the on / off byte of every row of the range are
// staged in LDS once, then every chunk is `on[row] ? pattern[chunk in row] : 0` with the (row, chunk) pair advanced
// incrementally — about ten instructions per 16 bytes where the generic kernel spends eighty on divisions and byte masks.
__global__ __launch_bounds__(256) void mask_connect_rows_kernel(DevState S, const ObsDigest* digest, int8_t* dst, size_t env_stride, size_t region_off,
                                                                uint32_t Nm, uint32_t Cc, uint32_t RL, uint32_t rows_per_block,
                                                                const uint8_t* env_mask, uint32_t skip_flagged) {
    extern __shared__ uint4 rows_lds[];
    const uint32_t cpr = RL >> 4, rows = Nm * Nm, bd = blockDim.x;
    uint4* pat = rows_lds;                                           // [cpr]
    uint8_t* on = reinterpret_cast<uint8_t*>(rows_lds + cpr);        // [rows_per_block]
    const uint32_t r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (uint32_t e = blockIdx.y; e < S.E; e += gridDim.y) {
        if (env_mask && !env_mask[e]) continue;                      // uniform per workgroup
        if (skip_flagged && (S.h0[e].y & F_SKIP)) continue;
        const ObsDigest d = digest[e];
        for (uint32_t j = threadIdx.x; j < cpr; j += bd) {
            uint32_t c = (j * 16u) % Cc, w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t i = 0; i < 16u; ++i) {
                w[i >> 2] |= (uint32_t)(c < d.n_creds) << (8u * (i & 3u));
                c = c + 1u == Cc ? 0u : c + 1u;
            }
            pat[j] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        for (uint32_t q = r0 + threadIdx.x; q < r1; q += bd) {
            const uint32_t s = q / Nm, t = q - s * Nm;
            on[q - r0] = (uint8_t)(((d.own_ext[(s >> 6) & 3u] >> (s & 63u)) & 1ull) && t < d.n_disc);
        }
        __syncthreads();
        uint4* out = reinterpret_cast<uint4*>(dst + (size_t)e * env_stride + region_off) + (size_t)r0 * cpr;
        const uint32_t total = (r1 - r0) * cpr, dq = bd / cpr, dj = bd - dq * cpr;
        uint32_t q = threadIdx.x / cpr, j = threadIdx.x - q * cpr;
        for (uint32_t c = threadIdx.x; c < total; c += bd) {
            const uint4 p = pat[j];
            out[c] = on[q] ? p : make_uint4(0, 0, 0, 0);
            j += dj; q += dq;
            if (j >= cpr) { j -= cpr; q += 1u; }
        }
        __syncthreads();                                             // the next env overwrites the staging area
    }
}

#define MCBS_INST(W) \
    template __global__ void mask_kernel<W, 0>(DevState, Topo, const StepCfg*, const ObsDigest*, int8_t*, size_t, size_t, uint32_t, uint32_t, const uint8_t*, uint32_t); \
    template __global__ void mask_kernel<W, 1>(DevState, Topo, const StepCfg*, const ObsDigest*, int8_t*, size_t, size_t, uint32_t, uint32_t, const uint8_t*, uint32_t); \
    template __global__ void mask_kernel<W, 2>(DevState, Topo, const StepCfg*, const ObsDigest*, int8_t*, size_t, size_t, uint32_t, uint32_t, const uint8_t*, uint32_t);
MCBS_INST(16)
MCBS_INST(4)
MCBS_INST(1)

} // namespace mcbs
