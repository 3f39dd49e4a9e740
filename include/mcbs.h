/* mcbs.h — C ABI of the MI355X-native batched CyberBattleSim step engine (libmcbs.so).
 *
 * Drop-in boundary for ONE hot path of zsh239040/MARLon: the attacker/defender environment
 * step behind `cyberbattle._env.cyberbattle_env.CyberBattleEnv.step` as driven by
 * `marlon.simulate` and marlon's env wrappers.  The reference is pure Python (no FFI of its
 * own); every entry point below names the reference interface it replaces.  Conventions:
 *   - plain pointers and sizes only, no torch / C++ types;
 *   - return 0 on success, a negative MCBS_E* code otherwise; mcbs_last_error() gives the
 *     thread-local message;
 *   - the library owns the environment state (HBM), the caller owns every I/O buffer
 *     (device pointers, e.g. torch-ROCm `tensor.data_ptr()`);
 *   - every call is asynchronous on the caller's stream (`void* stream` is a hipStream_t,
 *     NULL = the null stream) and performs no hidden synchronisation unless stated;
 *   - one batch handle is not re-entrant; different handles are independent (one per GPU).
 *
 * The same topology blob is read by the CPU oracle (oracle/cbs_oracle.c), which is test
 * infrastructure and never linked into this library.
 */
#ifndef MCBS_H
#define MCBS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCBS_ABI_VERSION 1u

/* ---- error codes ---- */
#define MCBS_OK           0
#define MCBS_EINVAL      -1   /* bad argument / malformed blob */
#define MCBS_ELIMIT      -2   /* topology exceeds an engine limit (see MCBS_MAX_*) */
#define MCBS_EHIP        -3   /* HIP runtime error (message has hipGetErrorString) */
#define MCBS_ENOMEM      -4
#define MCBS_ESTATE      -5   /* call not valid in the current state */

/* ---- engine limits ---- */
#define MCBS_MAX_NODES        256   /* node ids are u8; masks are <= 4 x u64 */
#define MCBS_MAX_PORTS         32   /* firewall / listen tables are u32 port masks */
#define MCBS_MAX_PROPS         60   /* property sets share a u64 with the 4 privilege tags of a node row */
#define MCBS_MAX_SLOTS         32   /* vulnerabilities applicable to one node (library + own) */
#define MCBS_MAX_LOCAL_VULNS   32   /* local-vulnerability mask per node is u32 */
#define MCBS_MAX_CRED_STRINGS 256    /* every set is held in <= 4 x u64 registers per env */
#define MCBS_MAX_TRIPLES      1024  /* distinct (node, port, credential) triples; more than 256 are kept as a wide set in memory */

/* ================================================================================
 * Topology blob ("MCBT", little endian, every section 16-byte aligned).
 * Built on the host by marlon_amd.flatten from a model.Environment
 * (reference records: simulation/model.py:63-77,226-247,263-345,347-362,377-396).
 * ================================================================================ */
#define MCBS_TOPO_MAGIC 0x5442434Du /* "MCBT" */

/* outcome kinds (reference classes, simulation/model.py:118-198) */
enum {
    MCBS_OUT_NONE = 0,
    MCBS_OUT_LEAKED_CREDENTIALS = 1,
    MCBS_OUT_LEAKED_NODES = 2,
    MCBS_OUT_PRIVILEGE_ESCALATION = 3,
    MCBS_OUT_LATERAL_MOVE = 4,
    MCBS_OUT_CUSTOMER_DATA = 5,
    MCBS_OUT_PROBE_SUCCEEDED = 6,
    MCBS_OUT_PROBE_FAILED = 7,
    MCBS_OUT_EXPLOIT_FAILED = 8,
    MCBS_OUT_OTHER = 9
};

/* precondition byte code (oracle only; the GPU uses mcbs_vuln_slot.precond_tt) */
enum {
    MCBS_OP_PROP_BASE = 0x00, /* 0x00..0x3F push static property bit i */
    MCBS_OP_TAG_BASE = 0x40,  /* 0x40..0x43 push tag privilege_k */
    MCBS_OP_TRUE = 0x80,
    MCBS_OP_FALSE = 0x81,
    MCBS_OP_NOT = 0x82,
    MCBS_OP_AND = 0x83,
    MCBS_OP_OR = 0x84
};

#define MCBS_NODE_INSTALLED0 0x01u /* agent_installed in the initial environment */
#define MCBS_NODE_REIMAGABLE 0x02u

typedef struct mcbs_topo_header { /* 192 bytes */
    uint32_t magic, abi_version, total_bytes, header_bytes;
    uint32_t n_nodes, n_ports, n_props, n_local, n_remote;
    uint32_t n_cred_strings, n_triples, max_slots;
    uint32_t n_slots_total, n_payload, n_services, n_allowed, n_code;
    uint32_t max_leak_per_action; /* longest LeakedCredentials list (env.py:421-428) */
    uint32_t avail_any_order;     /* 1: every availability term is an exact multiple of one power of two,
                                     so the node-order sum of actions.py:728-745 is order independent */
    uint32_t reserved0;
    double   total_sla_weight;    /* sum of node sla weights, node order */
    double   full_availability;   /* availability with every node Running (node-order sum / total) */
    uint32_t off_node;            /* mcbs_node_static[n_nodes] */
    uint32_t off_slot_of;         /* uint8[n_nodes * (n_local + n_remote)], 0xFF = not present */
    uint32_t off_slot;            /* mcbs_vuln_slot[n_nodes * max_slots] */
    uint32_t off_payload;         /* mcbs_payload[n_payload] */
    uint32_t off_service;         /* mcbs_service[n_services] */
    uint32_t off_allowed;         /* uint16[n_allowed] credential-string ids */
    uint32_t off_triple;          /* mcbs_triple[n_triples] */
    uint32_t off_code;            /* uint8[n_code] precondition byte code */
    uint32_t off_init_order;      /* uint8[n_nodes]: nodes owned at reset, network order, then 0xFF */
    uint32_t n_init_owned;
    double   full_sum;            /* node-order sum of every avail_term (numerator of full_availability) */
    /* learned-defender tier (marlon/defender_agents/defender.py): firewall rule lists by NAME */
    /* Rule LISTS are first-class: topologies routinely hand the same Python list object to several nodes / directions
     * (toy_ctf.py:14-19,25,73; chainpattern.py:49-54,103), copy.deepcopy keeps that aliasing inside each env, so an edit
     * through one node is seen through every alias.  A node refers to its two lists by id. */
    uint32_t off_fw_rule;         /* mcbs_fw_rule[n_fw_rules]: the rules of list 0, then list 1, ... in list order (oracle) */
    uint32_t n_fw_rules;
    uint32_t off_fw_range;        /* uint16[n_fw_lists * 2]: {offset, count} of each list in the rule array */
    uint32_t n_names;             /* firewall port names: ids 0..n_ports-1 are the identifier ports, then other names */
    uint8_t  rule_name[8];        /* name ids of LearningDefender.firewall_rule_list = RDP, SSH, HTTPS, HTTP, su, sudo (6 used) */
    uint8_t  rule_port[8];        /* identifier-port index of each of those names, 0xFF if it is not an identifier port */
    uint32_t n_fw_lists;          /* distinct rule list objects */
    uint32_t off_fw_list0;        /* uint16[n_fw_lists]: initial state of the six manageable names in each list: bit r = a rule named r
                                     exists, bit 6+r = the first one is ALLOW */
    uint32_t off_ere;             /* mcbs_ere_tables + arrays: what the ExternalRandomEvents defender needs (defender.py:58-148) */
    uint32_t reserved;
} mcbs_topo_header;

/* Tables of the ExternalRandomEvents defender; array offsets are relative to the start of this structure.  A vulnerability
 * "column" is its index in the identifiers: local l -> l, remote r -> n_local + r (at most 64 columns). */
typedef struct mcbs_ere_tables { /* 40 bytes */
    uint32_t n_library;       /* global library vulnerabilities = the first n_library slots of every node */
    uint32_t key_cap;         /* capacity of a node's key list: the longest own list + n_library */
    uint32_t off_own_keys;    /* uint8[n_nodes][key_cap]: the node's OWN vulnerability dictionary keys in order, as columns (0xFF pad) */
    uint32_t off_own_cnt;     /* uint8[n_nodes] */
    uint32_t off_lib_sorted;  /* uint8[n_library]: library columns in name order (numpy.setdiff1d returns sorted names) */
    uint32_t pad;
    uint64_t lib_cols;        /* bit c: column c is a library vulnerability */
    uint8_t  sample_name[8];  /* firewall name ids of model.SAMPLE_IDENTIFIERS.ports (7 used): the ports firewall_change_add opens */
} mcbs_ere_tables;

typedef struct mcbs_node_static { /* 64 bytes */
    uint64_t props;        /* static properties that are declared identifiers (bit = index) */
    double   sla_weight;
    double   avail_term;   /* sla_weight * (1 + running service weights) / (1 + all service weights) */
    int32_t  value;
    uint32_t fw_in_allow;  /* port bit set iff the FIRST incoming rule for the port is ALLOW */
    uint32_t fw_out_allow;
    uint32_t listen;       /* port bit set iff some service has that name (running or not) */
    uint32_t local_mask;   /* bit l set iff local vuln id l is in the library or in the node's dict (env.py:658-663) */
    uint16_t svc_off, svc_cnt;
    uint8_t  flags;        /* MCBS_NODE_* */
    uint8_t  priv0;        /* initial privilege level */
    uint8_t  tags0;        /* privilege_k tags literally present in the initial property list */
    uint8_t  n_slots;
    uint32_t fw_lists;     /* rule list ids: incoming | outgoing << 16 (learned-defender tier) */
    uint32_t pad[2];
} mcbs_node_static;

typedef struct mcbs_fw_rule { /* 2 bytes */
    uint8_t name;          /* firewall port name id */
    uint8_t allow;         /* RulePermission.ALLOW */
} mcbs_fw_rule;

typedef struct mcbs_vuln_slot { /* 32 bytes; slot s of node n applies to target n */
    double   cost;
    uint64_t probe_mask;   /* ProbeSucceeded: declared, non-tag properties it reveals */
    uint32_t payload_off;
    uint16_t payload_cnt;
    uint16_t precond_tt;   /* bit t = precondition value on this node when its tag set is t (4 bits) */
    uint32_t code_off;     /* oracle byte code */
    uint16_t code_len;
    uint8_t  kind;         /* MCBS_OUT_* */
    uint8_t  level;        /* PrivilegeEscalation level */
} mcbs_vuln_slot;

typedef struct mcbs_payload { /* 8 bytes: one LeakedCredentials / LeakedNodesId list entry */
    uint16_t node;
    uint16_t cred;         /* credential-string id (LeakedCredentials) */
    uint16_t triple;       /* (node, port, credential) triple id (LeakedCredentials) */
    uint16_t port;
} mcbs_payload;

typedef struct mcbs_service { /* 16 bytes */
    double   sla_weight;
    uint16_t allowed_off, allowed_cnt;
    uint8_t  port, running;
    uint16_t pad;
} mcbs_service;

typedef struct mcbs_triple { /* 8 bytes */
    uint16_t node, cred;
    uint16_t port, pad;
} mcbs_triple;

/* ================================================================================
 * Batch configuration = CyberBattleEnv constructor arguments (env.py:470-485) plus the
 * in-env defender (defender.py:27-55) and batching knobs.
 * ================================================================================ */
#define MCBS_DEFENDER_NONE 0
#define MCBS_DEFENDER_SCAN_AND_REIMAGE 1 /* ScanAndReimageCompromisedMachines */
#define MCBS_FW_GROWTH 120             /* ExternalRandomEvents: rules a list may gain beyond its initial length before the overflow flag */
#define MCBS_DEFENDER_RANDOM_EVENTS 3    /* ExternalRandomEvents (defender.py:58-148): random patching / planting of vulnerabilities,
                                           service stops and firewall edits, every step, on every node */
#define MCBS_DEFENDER_EXTERNAL 2         /* no in-env defender; a learned defender acts through mcbs_defender_step
                                           (marlon: DefenderEnvWrapper + LearningDefender) */

#define MCBS_RNG_PHILOX 0 /* draw i of a step = half (i&1) of Philox4x32-10(key = (seed lo, seed hi ^ env id hi),
                             ctr = (global env id lo, episode, step_count, i>>1)), 53-bit doubles (hi>>5, lo>>6) */
#define MCBS_ACTION_SKIP 3

#define MCBS_RNG_TAPE   1 /* draws read from a caller tape (parity against the reference's global RNGs) */

typedef struct mcbs_batch_cfg {
    uint32_t abi_version;
    uint32_t n_envs;
    int32_t  device;                 /* HIP device ordinal */
    uint32_t maximum_node_count;     /* bounds (env.py:172-224) */
    uint32_t maximum_total_credentials;
    uint32_t maximum_discoverable_credentials_per_action;
    /* AttackerGoal (env.py:227-241); has_attacker_goal = 0 means attacker_goal=None */
    uint32_t has_attacker_goal;
    uint32_t goal_own_atleast;
    double   goal_reward;
    double   goal_low_availability;
    double   goal_own_atleast_percent;
    /* DefenderGoal / DefenderConstraint (env.py:244-254) */
    uint32_t defender_goal_eviction;
    uint32_t defender_kind;
    double   maintain_sla;
    double   winning_reward, losing_reward;
    /* ScanAndReimageCompromisedMachines(probability, scan_capacity, scan_frequency) */
    double   scan_probability;
    uint32_t scan_capacity, scan_frequency;
    /* batching */
    uint32_t auto_reset;             /* 1: an env that ends is re-initialised inside the same step (VecEnv semantics) */
    uint32_t max_episode_steps;      /* 0 = unlimited; else `truncated` is raised at this step count */
    uint32_t rng_kind;
    uint32_t reserved0;
    uint64_t seed;
    uint64_t env_id_base;            /* global id of env 0 of this shard (multi-GPU: rank * n_envs) */
} mcbs_batch_cfg;

typedef struct mcbs_topology mcbs_topology;
typedef struct mcbs_batch mcbs_batch;

/* Observation buffers, marlon-flat layout = what AttackerEnvWrapper.transform_observation
 * returns (marlon/baseline_models/env_wrappers/attack_wrapper.py:474-522; fields built at
 * env.py:753-773,859-933).  Any pointer may be NULL to skip that field.  E = n_envs,
 * N = maximum_node_count, C = maximum_total_credentials, K = max discoverable per action. */
typedef struct mcbs_obs_buffers {
    int32_t* scalars;                     /* [E,7]: newly_discovered_nodes_count, lateral_move, customer_data_found,
                                             probe_result, escalation, credential_cache_length, discovered_node_count */
    int32_t* leaked_credentials;          /* [E,K,4] rows (used, cache_idx, target ext idx, port idx) */
    int32_t* credential_cache_matrix;     /* [E,C,2] rows (target ext idx, port idx) */
    int32_t* discovered_nodes_properties; /* [E,N,n_props] */
    int32_t* nodes_privilegelevel;        /* [E,N] */
    int8_t*  mask_local;                  /* [E,N,L] */
    int8_t*  mask_remote;                 /* [E,N,N,R] */
    int8_t*  mask_connect;                /* [E,N,N,P,C] */
    int8_t*  mask_discrete;               /* [E, N*N*P*C + N*L + N*N*R]: MaskedDiscreteAttackerWrapper.action_masks()
                                             order (action_masking.py:96-110): connect, local, remote */
} mcbs_obs_buffers;

/* Per-step outputs of mcbs_step beyond reward/terminated (StepInfo, env.py:1176-1182). */
typedef struct mcbs_info_buffers {
    double*  network_availability; /* [E] */
    int32_t* step_count;           /* [E] */
    uint8_t* truncated;            /* [E] */
    uint8_t* out_of_bound;         /* [E] 1 = OutOfBoundIndexError path was taken (env.py:1171-1174) */
    float*   raw_reward;           /* [E] ActionResult.reward before the clamp / terminal override of env.py:1162-1169
                                      (0 on the out-of-bound path); parity and reward-shaping aid */
} mcbs_info_buffers;

const char* mcbs_last_error(void);
uint32_t    mcbs_abi_version(void);

/* model.Environment (+ identifiers) -> device-resident tables.  Replaces the deep-copied
 * networkx graph of env.py:375-376. */
int  mcbs_topology_create(const void* blob, size_t nbytes, int32_t device, mcbs_topology** out);
void mcbs_topology_destroy(mcbs_topology*);

/* CyberBattleEnv.__init__ (env.py:470-566) for n_envs environments sharing one topology. */
int  mcbs_batch_create(const mcbs_topology*, const mcbs_batch_cfg*, mcbs_batch** out);
void mcbs_batch_destroy(mcbs_batch*);

/* CyberBattleEnv.reset (env.py:1187-1209) for every env, or for the envs whose byte in the
 * device array env_mask[E] is non-zero (NULL = all). */
int  mcbs_reset(mcbs_batch*, const uint8_t* env_mask, void* stream);

/* Every env back to the state mcbs_batch_create left it in: mcbs_reset for the whole batch with the episode counters at 0 again
 * (mcbs_reset advances them, so that an env's next episode draws fresh defender randomness: Philox is keyed by (seed, global env id,
 * episode, step)).  The reference's counterpart is constructing the environment anew (env.py:470-566); used to replay a recorded
 * trajectory from its start (bench.py, tools/). */
int  mcbs_rewind(mcbs_batch*, void* stream);

/* CyberBattleEnv.step (env.py:1145-1185) for all envs in one launch, observation excluded.
 * actions: device int32 [E,5] rows (kind, a, b, c, d):
 *     kind 0 local_vulnerability  (source, vuln)              -- action dict order of env.py:540-559
 *     kind 1 remote_vulnerability (source, target, vuln)
 *     kind 2 connect              (source, target, port, credential index)
 *     kind 3 skip                 the env is not stepped at all (no step count, no defender, state untouched);
 *                                 reward 0, terminated 0.  This is how marlon's AttackerEnvWrapper handles an
 *                                 action whose node index is not discovered yet (attack_wrapper.py:286-308).
 * reward: device float [E]; terminated: device uint8 [E]; info may be NULL.
 * An env that is done and not auto-reset is left untouched (reward 0, terminated 1): the
 * single-env facade raises the reference's RuntimeError (env.py:1146-1147) on the host. */
int  mcbs_step(mcbs_batch*, const int32_t* actions, float* reward, uint8_t* terminated,
               const mcbs_info_buffers* info, void* stream);

/* n_steps consecutive steps of every env in ONE launch, for action sequences that are known in advance (recorded traces,
 * scripted plans, pre-sampled random agents: marlon.simulate's random-agent loops, marlon/simulate.py:14-35):
 * actions [n_steps, E, 5], reward / terminated [n_steps, E].  Identical in effect to n_steps calls of mcbs_step without info
 * buffers (auto-reset and truncation included); needs the Philox generator when a defender is configured.  The reference has
 * no counterpart; what it saves is the cost between dependent launches. */
int  mcbs_step_many(mcbs_batch*, const int32_t* actions, float* reward, uint8_t* terminated, uint32_t n_steps, void* stream);

/* Random agents on the device: n_steps steps in one launch, each env's action drawn inside the kernel from its own state
 * (valid != 0: the distribution of CyberBattleEnv.sample_valid_action, env.py:959-1047; else uniform over the action space) with
 * Philox keyed by (seed, global env id, first_step + k) — the action mcbs_sample_actions(valid, seed, first_step + k) would give.
 * actions_out [n_steps, E, 5] may be NULL.  marlon.simulate's loop for random agents (marlon/simulate.py:14-35). */
int  mcbs_rollout_random(mcbs_batch*, int32_t valid, uint64_t seed, uint64_t first_step, uint32_t n_steps,
                         int32_t* actions_out, float* reward, uint8_t* terminated, void* stream);

/* Same transition, but the observation is written exactly where the reference assembles it:
 * after the attacker's action and BEFORE the defender acts (env.py:1153 vs 1156-1158).
 * Three launches: attacker phase, observation, defender + goals. */
int  mcbs_step_observe(mcbs_batch*, const int32_t* actions, float* reward, uint8_t* terminated,
                       const mcbs_info_buffers* info, const mcbs_obs_buffers* obs, void* stream);

/* Observation of the current state: the reset observation (env.py:1197-1200) right after
 * mcbs_reset; otherwise the state-aggregated fields with the per-step flags of the last action. */
int  mcbs_observe(mcbs_batch*, const mcbs_obs_buffers* obs, void* stream);

/* mcbs_observe restricted to the envs whose byte in the device array env_mask[E] is non-zero; the buffers of the other
 * envs are left as they are (VecEnv auto-reset: only the envs that were just reset get a fresh observation). */
int  mcbs_observe_masked(mcbs_batch*, const mcbs_obs_buffers* obs, const uint8_t* env_mask, void* stream);

/* CyberBattleEnv.compute_action_mask (env.py:679-683): the three masks (and/or mask_discrete) of the CURRENT state,
 * whatever the last step was (mcbs_observe returns all-zero masks after an out-of-bound step, like the reference's
 * blank observation).  Only the mask_* members of the buffers are used. */
int  mcbs_action_mask(mcbs_batch*, const mcbs_obs_buffers* masks, void* stream);

/* Row stride of mcbs_obs_buffers.mask_discrete in bytes for this batch (0 = dense: mcbs_discrete_action_count bytes per env, the default).
 * The flat mask's length is rarely a multiple of a cache line (Chain-10 @12/12: 14 172 bytes), so dense rows share 128-byte lines with
 * their neighbours; a caller that pads its rows to a multiple of 128 bytes (and aligns the array) gets every env's mask on lines of its
 * own — the connect region is then streamed with aligned non-temporal stores like mask_connect.  Bytes of a row beyond the action count
 * are never written.  Applies to mcbs_observe / mcbs_step_observe / mcbs_action_mask / mcbs_observe_masked / mcbs_attacker_wrapper_step. */
int  mcbs_set_mask_discrete_stride(mcbs_batch*, size_t stride_bytes);

/* StepInfo fields without stepping. */
int  mcbs_step_info(mcbs_batch*, const mcbs_info_buffers* info, void* stream);

/* Random-agent harness (env.py:935-1055): `valid` != 0 draws like sample_valid_action (source
 * among owned nodes, target among discovered, rejection against the action mask); 0 draws every
 * component uniformly in its bound, invalid actions included.  Philox stream separate from the
 * defender's.  actions_out: device int32 [E,5]. */
int  mcbs_sample_actions(mcbs_batch*, int32_t valid, uint64_t seed, uint64_t step, int32_t* actions_out, void* stream);

/* marlon's attacker action encodings -> engine action rows, on the device.
 *   multidiscrete: int64 [E,10] = AttackerEnvWrapper's MultiDiscrete (attack_wrapper.py:206-227,255-267):
 *                  [kind, l_src, l_vuln, r_src, r_tgt, r_vuln, c_src, c_tgt, c_port, c_cred]; or NULL
 *   discrete:      int64 [E]    = MaskedDiscreteAttackerWrapper's Discrete index (action_masking.py:112-142):
 *                  connect block ((src*N+tgt)*P+port)*C+cred, then local src*L+vuln, then remote (src*N+tgt)*R+vuln
 * Exactly one of the two is non-NULL.  An action whose source / target index is not below the env's discovered-node
 * count is turned into a skip row and flagged in invalid[E] (attack_wrapper.py:236-253,286-308). */
int  mcbs_decode_attacker_actions(mcbs_batch*, const int64_t* multidiscrete, const int64_t* discrete,
                                  int32_t* actions_out, uint8_t* invalid_out, void* stream);

/* ---- on-device action mask -> logits (what MaskablePPO does with action_masks(), train_marl_multi.py:259-293) ----
 * logits: device [E, row_stride] float32 (MCBS_LOGITS_F32) or bfloat16 (MCBS_LOGITS_BF16), row e = the Discrete action scores of env e in
 * MaskedDiscreteAttackerWrapper's order (action_masking.py:96-142: connect, local, remote; mcbs_discrete_action_count entries).
 * In place: logits[e, a] = mask(e, a) ? logits[e, a] : fill, where mask is EXACTLY the mask_discrete the last observation call on this
 * batch (mcbs_step_observe / mcbs_observe / mcbs_observe_masked / mcbs_action_mask) wrote or would have written — it is rebuilt from the
 * per-env digest that call left (owned-source bits, discovered-node and cached-credential counts), so the N*N*P*C-byte mask itself need
 * not be requested from the observation at all.  Not available for MCBS_DEFENDER_RANDOM_EVENTS batches (MCBS_ESTATE).
 * MCBS_ESTATE too while the digests cannot be trusted: before the first whole-batch observation, after mcbs_reset (whole batch: until the
 * next observation; by mask: until mcbs_observe_masked has re-observed those envs) and after mcbs_set_state.  (The local-vulnerability
 * block is read through the env's live discovery list, which only agrees with the digest's counts in those states.)
 * Every launching entry point of this header selects the batch's device for its own duration (the caller's current device is restored). */
#define MCBS_LOGITS_F32  0
#define MCBS_LOGITS_BF16 1
uint64_t mcbs_discrete_action_count(const mcbs_batch*);
int  mcbs_mask_logits(mcbs_batch*, void* logits, int32_t dtype, size_t row_stride, float fill, void* stream);

/* ---- learned defender: marlon/baseline_models/env_wrappers/defend_wrapper.py:197-327,329-412,492-534
 * and marlon/defender_agents/defender.py:31-107, for batches created with MCBS_DEFENDER_EXTERNAL ---- */
typedef struct mcbs_defender_obs {   /* DefenderEnvWrapper.observe: four MultiBinary fields, int8, network node order */
    int8_t* infected_nodes;            /* [E, n_nodes]      agent_installed */
    int8_t* incoming_firewall_status;  /* [E, n_nodes * 6]  a rule named RDP/SSH/HTTPS/HTTP/su/sudo exists in the incoming list */
    int8_t* outgoing_firewall_status;  /* [E, n_nodes * 6] */
    int8_t* services_status;           /* [E, n_services]   service.running, node order then service order */
} mcbs_defender_obs;

/* One defender turn for every env: validity of the action (is_defender_action_valid), then
 * LearningDefender.executeAction = DefenderAgentActions.on_attacker_step_taken() followed by the action if it was valid.
 * actions: device int64 [E,12] = DefenderEnvWrapper's MultiDiscrete [5,N,N,6,2,N,6,2,N,3,N,3]:
 *   [0] kind: 0 reimage([1]) 1 block_traffic([2] node,[3] rule name,[4] incoming) 2 allow_traffic([5],[6],[7])
 *       3 stop_service([8],[9]) 4 start_service([10],[11]); kind -1 = the empty action (the defender skips its turn, the
 *       tick still happens); kind <= -2 = the env takes no part in this call (no tick, state untouched).
 * Outputs (device): valid[E], availability[E] (after the tick), evicted[E] (= __defender_goal_reached), obs (may be NULL).
 * Reference defects reproduced as they are: stop/start_service never match a service (defender.py:45-48 hands a
 * ListeningService object to actions.py:782-794) so they change nothing; allow_traffic appends to the INCOMING list in
 * both branches (defender.py:68).  Not reproduced (DESIGN.md "Q14"): the reference's defender keeps acting on the
 * environment object that existed before the first reset(); here it always acts on the live environment. */
int  mcbs_defender_step(mcbs_batch*, const int64_t* actions, uint8_t* valid, double* availability, uint8_t* evicted,
                        const mcbs_defender_obs* obs, void* stream);
int  mcbs_defender_observe(mcbs_batch*, const mcbs_defender_obs* obs, void* stream);

/* Defender draw tape for MCBS_RNG_TAPE: device double [E, draws_per_step] consumed by the next
 * step (scan draws first, then detection draws in consumption order). */
int  mcbs_set_draw_tape(mcbs_batch*, const double* tape, uint32_t draws_per_step);

/* Bookkeeping of marlon's AttackerEnvWrapper.step around the environment step (attack_wrapper.py:286-354), for every env in one
 * launch: step / action counters, the reward modifier of an intercepted action, truncation at max_timesteps, episode returns.
 * All pointers are device arrays of n_envs elements owned by the caller (a batched wrapper keeps them next to its observation). */
typedef struct mcbs_wrapper_buffers {
    const uint8_t* invalid;        /* in : from mcbs_decode_attacker_actions */
    const float*   reward;         /* in : from mcbs_step / mcbs_step_observe */
    const uint8_t* terminated;     /* in */
    int32_t* timesteps;            /* in/out: wrapper steps of the current episode (invalid ones included) */
    int64_t* valid_action_count;   /* in/out */
    int64_t* invalid_action_count; /* in/out */
    double*  episode_returns;      /* in/out: sum of the wrapper's rewards */
    float*   last_cyber_reward;    /* out: the environment's own reward of this step (AttackerEnvWrapper.cyber_rewards[-1]) */
    uint8_t* has_cyber_reward;     /* out: 1 */
    float*   rewards;              /* out: reward + invalid * invalid_action_reward_modifier */
    uint8_t* truncated;            /* out: timesteps >= max_timesteps */
    uint8_t* dones;                /* out: terminated | truncated */
    double*  episode_return_out;   /* out: copies for the info dict, taken before a reset clears the counters */
    int32_t* episode_length_out;   /* out */
    int32_t* n_done;               /* out: one int32, number of envs with dones != 0 (zeroed by the call) */
    uint8_t* executed;             /* out, optional (may be NULL): !invalid — info["cyber_step_executed"] (mcbs_attacker_wrapper_finish only) */
} mcbs_wrapper_buffers;
int  mcbs_attacker_wrapper_post(mcbs_batch*, const mcbs_wrapper_buffers* w, float invalid_action_reward_modifier, int32_t max_timesteps,
                                void* stream);
/* ... and the counters of the envs whose dones flag is set back to zero (what the wrapper's reset() does for them). */
int  mcbs_attacker_wrapper_clear(mcbs_batch*, const mcbs_wrapper_buffers* w, void* stream);

/* dst[i][e] = src[i][e] (rows of row_bytes[i] bytes, device arrays of n_envs rows) for the envs whose byte in env_mask is non-zero and up to
 * eight arrays in one launch: the terminal observation of the envs that just ended — what SB3's DummyVecEnv keeps in
 * infos[i]["terminal_observation"] before it resets an env.  A batched wrapper calls it with mcbs_wrapper_buffers.dones as the mask, then
 * mcbs_reset + mcbs_observe_masked with the same mask: no host round trip to learn which envs ended.  Like those two it scans the mask
 * 64 envs per wavefront, so it costs a few microseconds when no env ended. */
typedef struct mcbs_row_copies {
    uint32_t n, pad;
    const void* src[8];
    void*       dst[8];
    size_t      row_bytes[8];
} mcbs_row_copies;
int  mcbs_copy_rows_masked(mcbs_batch*, const mcbs_row_copies* copies, const uint8_t* env_mask, void* stream);

/* One whole step of marlon's attacker wrappers for the batch — AttackerEnvWrapper.step (attack_wrapper.py:255-372) under
 * MaskedDiscreteAttackerWrapper (action_masking.py:112-142) under SB3's DummyVecEnv — enqueued by ONE call:
 *   mcbs_decode_attacker_actions (exactly one of multidiscrete [n_envs, 10] / discrete [n_envs]; `decoded` [n_envs, 5] receives the rows,
 *   w->invalid the interception flags), mcbs_step_observe (w->reward and w->terminated receive the environment's reward and done flags:
 *   the three `in` arrays of mcbs_wrapper_buffers are written by this call), mcbs_attacker_wrapper_finish.
 * Stream-ordered, no host synchronisation, hipGraph-capturable (three launches: decode + attacker half, observation, defender half + finish). */

/* Everything an auto-resetting attacker wrapper does after the environment step, in ONE launch: mcbs_attacker_wrapper_post for every env
 * (n_done may be NULL here: nothing is counted; `executed` is written if given), then for the envs whose `dones` it has just set — what SB3's DummyVecEnv.step_wait does
 * with an env that reports done (baseline_marlon_agent.py:100-167 runs the wrappers under it) —
 *   keep:   dst[i][e] = src[i][e]        the episode's last observation (infos[e]["terminal_observation"]),
 *   the env is reset (mcbs_reset for it: reset image, episode counter + 1),
 *   fresh:  dst[i][e] = src[i][0]        the reset observation: `src` arrays hold ONE row, the observation of a freshly reset env
 *                                        (every env resets to the same state, so the wrapper keeps row 0 of its first observation),
 *   the env's digest (mcbs_mask_logits) becomes that of a freshly reset env, and its wrapper counters go back to zero (wrapper_clear).
 * keep / fresh may be NULL (or n = 0); with auto_reset == 0 only the bookkeeping runs.  The batch must have been reset as a whole
 * (mcbs_reset with a NULL mask) and observed once before the first call, so that the library holds a reset env's digest:
 * MCBS_ESTATE otherwise.  Replaces five launches and a memset of the round-2 wrapper step. */
int  mcbs_attacker_wrapper_step(mcbs_batch*, const int64_t* multidiscrete, const int64_t* discrete, int32_t* decoded,
                                const mcbs_info_buffers* info, const mcbs_obs_buffers* obs, const mcbs_wrapper_buffers* w,
                                float invalid_action_reward_modifier, int32_t max_timesteps, int32_t auto_reset,
                                const mcbs_row_copies* keep, const mcbs_row_copies* fresh, void* stream);
int  mcbs_attacker_wrapper_finish(mcbs_batch*, const mcbs_wrapper_buffers* w, float invalid_action_reward_modifier, int32_t max_timesteps,
                                  int32_t auto_reset, const mcbs_row_copies* keep, const mcbs_row_copies* fresh, void* stream);
/* How many kernel launches one mcbs_attacker_wrapper_step of this batch takes: 1 for small topologies (packed batch, at most 16 nodes and
 * cached credentials) when no mask field is requested (with_masks == 0) — the whole step, observation included, is then ONE launch
 * (marlon_amd/csrc/mcbs_wrapper_fused.hip) and replaying it from a hipGraph would only add the graph's own launch cost —, else 3. */
int32_t mcbs_attacker_wrapper_step_launches(const mcbs_batch*, int32_t with_masks);

/* The reward shaping of marlon's DefenderEnvWrapper.step (defend_wrapper.py:228-282) around mcbs_defender_step, for every env in one
 * launch and in the wrapper's own order of double-precision operations: invalid-action penalty, minus the attacker's last environment
 * reward, loss_reward when availability first drops below maintain_sla (terminating if reset_on_constraint_broken), a penalty
 * proportional to the worsening while breached, winning_reward when the attacker is evicted; truncation at max_timesteps. */
typedef struct mcbs_defender_wrapper_buffers {
    const uint8_t* valid;          /* in : mcbs_defender_step outputs */
    const double*  availability;   /* in */
    const uint8_t* evicted;        /* in */
    const uint8_t* attacker_has_cyber_reward;  /* in : the attacker wrapper's buffers (mcbs_wrapper_buffers) */
    const float*   attacker_last_cyber_reward; /* in */
    int32_t* timesteps;            /* in/out */
    int64_t* valid_action_count;   /* in/out */
    int64_t* invalid_action_count; /* in/out */
    uint8_t* has_breached_sla;     /* in/out */
    double*  prev_availability;    /* in/out */
    double*  reward;               /* out */
    uint8_t* terminated;           /* out */
    uint8_t* truncated;            /* out */
    uint8_t* breached;             /* out: availability < maintain_sla */
    uint8_t* won;                  /* out: attacker evicted */
} mcbs_defender_wrapper_buffers;
typedef struct mcbs_defender_wrapper_cfg {
    double invalid_action_penalty, loss_reward, sla_worsening_penalty_scale, maintain_sla, winning_reward;
    int32_t reset_on_constraint_broken, max_timesteps;
} mcbs_defender_wrapper_cfg;
int  mcbs_defender_wrapper_post(mcbs_batch*, const mcbs_defender_wrapper_buffers* w, const mcbs_defender_wrapper_cfg* cfg, void* stream);
/* One whole step of marlon's DefenderEnvWrapper for the batch (defend_wrapper.py:197-327): mcbs_defender_step and mcbs_defender_wrapper_post
 * in ONE launch — the shaping takes the turn's results from registers; w->valid / availability / evicted (the `in` members) are written
 * by this call.  When obs is given the observation is written by the same launch (topologies of up to 32 nodes, arrays on 16-byte
 * boundaries) or by a second one. */
int  mcbs_defender_wrapper_step(mcbs_batch*, const int64_t* actions, const mcbs_defender_obs* obs, const mcbs_defender_wrapper_buffers* w,
                                const mcbs_defender_wrapper_cfg* cfg, void* stream);

/* Parity / debugging: canonical per-env state records (layout: mcbs_state_record below),
 * host buffers, synchronous.  The record does not carry what only MCBS_DEFENDER_RANDOM_EVENTS mutates (vulnerability keys,
 * service flags, firewall rule lists): mcbs_set_state puts those back to the topology's initial ones for the envs it writes. */
size_t mcbs_state_record_bytes(const mcbs_batch*);
int  mcbs_get_state(mcbs_batch*, void* host_buf, size_t nbytes);
int  mcbs_set_state(mcbs_batch*, const void* host_buf, size_t nbytes);

/* Kernel timing hook for bench.py: HIP events recorded on `stream` around each mcbs_step launch
 * while enabled; mcbs_timing_read synchronises and returns the summed kernel milliseconds. */
int  mcbs_timing_enable(mcbs_batch*, int32_t on);
int  mcbs_timing_read(mcbs_batch*, double* total_ms, uint64_t* launches);

/* Canonical state record (host side, used by get/set_state and the oracle's dump):
 * fixed header followed by n_nodes node records, the discovery order and the credential cache. */
typedef struct mcbs_state_header { /* 64 bytes */
    uint32_t step_count, done, truncated, episode;
    uint32_t n_discovered, n_creds;
    uint32_t last_outcome_kind, last_escalation;
    uint32_t last_new_nodes, last_new_creds, last_oob, pad0;
    double   cum_reward;
    double   availability;
} mcbs_state_header;

typedef struct mcbs_state_node { /* 32 bytes */
    uint64_t discovered_props;
    uint32_t attacked_ever;   /* bit s: slot s exploited at least once (actions.py:396-407) */
    uint32_t attacked_since;  /* bit s: ... since the node's last re-imaging */
    uint8_t  discovered, installed, ever_owned, running;
    uint8_t  privilege, tags, countdown, pad;
    uint32_t pad1[2];
} mcbs_state_node;
/* followed by uint16 discovery_order[n_nodes] and uint16 credential_cache[max_total_credentials] (triple ids) */

#ifdef __cplusplus
}
#endif
#endif /* MCBS_H */
