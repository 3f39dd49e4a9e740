/* cbs_oracle.c — CPU restatement of the reference's attacker/defender step.  TEST INFRASTRUCTURE.
 *
 * This file is the parity oracle for the HIP engine in marlon_amd/csrc.  It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load
 * or call it.  It restates, one environment at a time and in plain scalar C, the algorithm of
 *
 *   src/CyberBattleSim/cyberbattle/_env/cyberbattle_env.py   (step :1145-1185, __execute_action :707-751,
 *        index translation :568-605, observation :753-773,:793-857,:859-933, action mask :621-677,
 *        goals :1080-1116, reset :375-394,:1187-1209)
 *   src/CyberBattleSim/cyberbattle/simulation/actions.py     (penalties :49-93, AgentActions :127-621,
 *        DefenderAgentActions :681-746)
 *   src/CyberBattleSim/cyberbattle/_env/defender.py          (ScanAndReimageCompromisedMachines :27-55)
 *
 * on purpose in the reference's own terms — explicit per-node records, membership lists searched
 * linearly, time stamps compared with >= , preconditions interpreted from byte code — and not in
 * the bit-mask / truth-table form the GPU kernels use, so that agreement between the two is
 * evidence and not a tautology.
 *
 * Pinning (tests/test_oracle_*.py): the reference's own known answers (ToyCtf command-and-control
 * walkthrough total 389.0, commandcontrol_test.py:71; the 57-action Chain-10 script ending `done`
 * at action 56, cyberbattle_env_test.py:41-114; the assertions of actions_test.py:291-422) and the
 * golden traces under tests/golden/ captured from the imported reference (oracle/refharness/).
 *
 * Wall-clock note: the reference orders ownership / attack / re-imaging events with
 * datetime.now() (actions.py:273,407,711; compared at :402,:521).  Here `now()` is a counter
 * that increases on every call, which preserves every comparison the reference makes.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>

#include "mcbs.h"

#define NONE_T (-1)            /* Python None for time stamps */
#define REIMAGING_DURATION 15  /* actions.py:685 */

/* actions.py:49-93 */
static const double P_SUSPICIOUSNESS = -5.0, P_SCANNING_UNOPEN_PORT = -10.0, P_REPEAT = -1, P_LOCAL_EXPLOIT_FAILED = -20,
                    P_FAILED_REMOTE_EXPLOIT = -50, P_MACHINE_NOT_RUNNING = 0, P_WRONG_PASSWORD = -10,
                    P_BLOCKED_BY_LOCAL_FIREWALL = -10, P_BLOCKED_BY_REMOTE_FIREWALL = -10, P_INVALID_ACTION = -1;
static const double NEW_SUCCESSFULL_ATTACK_REWARD = 7, NODE_DISCOVERED_REWARD = 5, CREDENTIAL_DISCOVERED_REWARD = 3,
                    PROPERTY_DISCOVERED_REWARD = 2;

enum { ST_STOPPED = 0, ST_RUNNING = 1, ST_IMAGING = 2 }; /* model.py:308-313 */

typedef struct {
    /* NodeInfo mutable fields (model.py:316-345) */
    int agent_installed, privilege_level, status;
    int64_t last_reimaging;
    int tag[4];                      /* privilege_k strings appended to node.properties (actions.py:378) */
    /* NodeTrackingInformation (actions.py:114-124) */
    int tracked;                     /* node_id in AgentActions._discovered_nodes */
    int64_t last_owned_at;
    int64_t last_attack[MCBS_MAX_SLOTS];
    int discovered_property[MCBS_MAX_PROPS];
    /* DefenderAgentActions.node_reimaging_progress (actions.py:689) */
    int reimaging, remaining;
} onode;

typedef struct {
    onode* node;
    int* tracked_order; int n_tracked;   /* AgentActions._discovered_nodes insertion order */
    int* discovered;    int n_discovered;/* CyberBattleEnv.__discovered_nodes */
    int* cache;         int n_cache;     /* CyberBattleEnv.__credential_cache (triple ids) */
    int* gathered;                       /* AgentActions._gathered_credentials, by credential-string id */
    /* the firewall rule list objects (model.py:275-305), mutable by the learned defender; a node's incoming / outgoing
     * list is looked up by id because several nodes / directions may hold the SAME list object */
    mcbs_fw_rule** fwl; int* n_fwl; int* cap_fwl;
    /* ExternalRandomEvents (defender.py:58-148) mutates these per env: every node's own vulnerability dictionary (keys as
     * identifier columns, in insertion order) and the running flag of every service */
    uint8_t* keys; uint8_t* kcnt; uint8_t* svc_running;
    int fw_overflow;                     /* a rule list hit its capacity: parity with the reference is lost from here on */
    int64_t clock;
    int stepcount, done, truncated, episode;
    double episode_reward_sum;           /* numpy.sum(__episode_rewards): rewards are exact integers or one rounded
                                            subtraction each; see DESIGN.md "reward arithmetic" */
    double availability;
    /* what the last action produced, for the observation (env.py:863-918) */
    int last_kind, last_level, last_new_nodes, last_new_creds, last_oob;
} oenv;

typedef struct {
    uint8_t* blob;
    const mcbs_topo_header* H;
    const mcbs_node_static* NS;
    const uint8_t* slot_of;
    const mcbs_vuln_slot* SL;
    const mcbs_payload* PL;
    const mcbs_service* SV;
    const uint16_t* AL;
    const mcbs_triple* TR;
    const uint8_t* CODE;
    const mcbs_fw_rule* FWR;
    const uint16_t* FWRANGE;
    const mcbs_ere_tables* ERE; const uint8_t* own_keys; const uint8_t* own_cnt; const uint8_t* lib_sorted;
    mcbs_batch_cfg cfg;
    int n_envs;
    oenv* env;
} oracle;

static int64_t now(oenv* e) { return ++e->clock; }

/* ---------------- Philox4x32-10 (Salmon et al., Random123) for the defender draws ---------------- */
static void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3], k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void cbo_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }

/* 53-bit double from two 32-bit words, the construction CPython's random.random() and numpy's legacy
 * random_sample() use on MT19937 output (defender.py:45,49 draw from those). */
static double to_double53(uint32_t a, uint32_t b) { return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0; }

typedef struct { const oracle* o; const oenv* e; uint64_t env_gid; const double* tape; int tape_len; int next; } draws;
static double next_draw(draws* d) {
    int i = d->next++;
    if (d->o->cfg.rng_kind == MCBS_RNG_TAPE) return (d->tape && i < d->tape_len) ? d->tape[i] : 0.0;
    uint32_t ctr[4] = { (uint32_t)d->env_gid, (uint32_t)d->e->episode, (uint32_t)d->e->stepcount, (uint32_t)(i >> 1) };
    uint32_t key[2] = { (uint32_t)d->o->cfg.seed, (uint32_t)(d->o->cfg.seed >> 32) ^ (uint32_t)(d->env_gid >> 32) };
    uint32_t r[4];
    philox4x32_10(ctr, key, r);
    return (i & 1) ? to_double53(r[2], r[3]) : to_double53(r[0], r[1]);
}

/* ---------------- topology access ---------------- */
static const mcbs_vuln_slot* slot(const oracle* o, int node, int s) { return &o->SL[(size_t)node * o->H->max_slots + s]; }

/* _check_prerequisites (actions.py:158-171): symbol true iff its name is in node.properties */
static int check_prerequisites(const oracle* o, const oenv* e, int target, const mcbs_vuln_slot* v) {
    int stack[64], sp = 0;
    const uint8_t* code = o->CODE + v->code_off;
    for (int i = 0; i < v->code_len; ++i) {
        uint8_t op = code[i];
        if (op < MCBS_OP_TAG_BASE) stack[sp++] = (int)((o->NS[target].props >> op) & 1u);
        else if (op < MCBS_OP_TAG_BASE + 4) stack[sp++] = e->node[target].tag[op - MCBS_OP_TAG_BASE];
        else if (op == MCBS_OP_TRUE) stack[sp++] = 1;
        else if (op == MCBS_OP_FALSE) stack[sp++] = 0;
        else if (op == MCBS_OP_NOT) stack[sp - 1] = !stack[sp - 1];
        else { int b = stack[--sp], a = stack[--sp]; stack[sp++] = (op == MCBS_OP_AND) ? (a && b) : (a || b); }
    }
    return sp == 1 && stack[0];
}

static void track(oenv* e, int n) { /* self._discovered_nodes[node_id] = NodeTrackingInformation() */
    if (!e->node[n].tracked) { e->node[n].tracked = 1; e->tracked_order[e->n_tracked++] = n; }
}

/* __mark_node_as_discovered (actions.py:227-232) */
static int mark_node_as_discovered(oenv* e, int n) {
    int newly = !e->node[n].tracked;
    if (newly) track(e, n);
    return newly;
}

/* __mark_nodeproperties_as_discovered (actions.py:234-245): properties given as a declared-property bit mask */
static int mark_nodeproperties_as_discovered(oenv* e, int n, uint64_t props) {
    int before = 0, after = 0;
    track(e, n);
    for (int p = 0; p < MCBS_MAX_PROPS; ++p) before += e->node[n].discovered_property[p];
    for (int p = 0; p < MCBS_MAX_PROPS; ++p) if ((props >> p) & 1u) e->node[n].discovered_property[p] = 1;
    for (int p = 0; p < MCBS_MAX_PROPS; ++p) after += e->node[n].discovered_property[p];
    return after - before;
}

/* __is_node_owned_history (actions.py:517-522) */
static void is_node_owned_history(const oenv* e, int n, int64_t* last_owned_at, int* currently_owned) {
    const onode* x = &e->node[n];
    *last_owned_at = x->tracked ? x->last_owned_at : NONE_T;
    *currently_owned = *last_owned_at != NONE_T && (x->last_reimaging == NONE_T || *last_owned_at >= x->last_reimaging);
}

/* __mark_node_as_owned (actions.py:251-275) */
static void mark_node_as_owned(const oracle* o, oenv* e, int n, int privilege, int64_t* last_owned_at, int* currently_owned) {
    is_node_owned_history(e, n, last_owned_at, currently_owned);
    if (!*currently_owned) {
        onode* x = &e->node[n];
        track(e, n);
        x->agent_installed = 1;
        x->privilege_level = x->privilege_level > privilege ? x->privilege_level : privilege; /* model.escalate */
        mark_nodeproperties_as_discovered(e, n, o->NS[n].props);  /* privilege tags are filtered out (:235) */
        x->last_owned_at = now(e);
    }
}

/* is_global_vulnerability or is_inplace_vulnerability (actions.py:339-351) on the env's CURRENT dictionaries: only the
 * ExternalRandomEvents defender changes them, otherwise the static slot table says it all */
static int random_events(const oracle* o) { return o->cfg.defender_kind == MCBS_DEFENDER_RANDOM_EVENTS; }
static int has_key(const oracle* o, const oenv* e, int node, int col) {
    const uint8_t* k = e->keys + (size_t)node * o->ERE->key_cap;
    for (int i = 0; i < e->kcnt[node]; ++i) if (k[i] == col) return 1;
    return 0;
}
static int vulnerability_present(const oracle* o, const oenv* e, int node, int col) {
    if (!random_events(o)) return 1;
    return ((o->ERE->lib_cols >> col) & 1u) || has_key(o, e, node, col);
}

typedef struct { double reward; int kind; int level; int slot_node; const mcbs_vuln_slot* v; } action_result;
static action_result result(double r, int kind) { action_result a; a.reward = r; a.kind = kind; a.level = 0; a.slot_node = -1; a.v = NULL; return a; }

/* __process_outcome (actions.py:325-423) + __mark_discovered_entities (:277-310) */
static int process_outcome(const oracle* o, oenv* e, int vuln_col, int node, double failed_penalty, action_result* out) {
    onode* x = &e->node[node];
    if (x->status != ST_RUNNING) { *out = result(P_MACHINE_NOT_RUNNING, MCBS_OUT_NONE); return 0; }
    int s = o->slot_of[(size_t)node * (o->H->n_local + o->H->n_remote) + vuln_col];
    if (s == 0xFF || !vulnerability_present(o, e, node, vuln_col)) { *out = result(P_SUSPICIOUSNESS, MCBS_OUT_NONE); return 0; }
    const mcbs_vuln_slot* v = slot(o, node, s);
    if (!check_prerequisites(o, e, node, v)) { *out = result(failed_penalty, MCBS_OUT_EXPLOIT_FAILED); return 0; }

    double reward = 0;
    if (v->kind == MCBS_OUT_PRIVILEGE_ESCALATION) {
        if (x->tag[v->level]) { *out = result(P_REPEAT, MCBS_OUT_PRIVILEGE_ESCALATION); out->level = v->level; return 0; }
        int64_t last; int cur;
        mark_node_as_owned(o, e, node, v->level, &last, &cur);
        if (last == NONE_T) reward += (double)o->NS[node].value;
        x->tag[v->level] = 1;
    } else if (v->kind == MCBS_OUT_LATERAL_MOVE) {
        int64_t last; int cur;
        mark_node_as_owned(o, e, node, 1, &last, &cur);
        if (last == NONE_T) reward += (double)o->NS[node].value;
    } else if (v->kind == MCBS_OUT_PROBE_SUCCEEDED) {
        reward += mark_nodeproperties_as_discovered(e, node, v->probe_mask) * PROPERTY_DISCOVERED_REWARD;
    }
    track(e, node);

    if (x->last_attack[s] != NONE_T) {                       /* already_executed */
        if (x->last_reimaging == NONE_T || x->last_attack[s] >= x->last_reimaging) reward += P_REPEAT;
    } else {
        reward += NEW_SUCCESSFULL_ATTACK_REWARD;
    }
    x->last_attack[s] = now(e);

    int newly_nodes = 0, newly_creds = 0;
    if (v->kind == MCBS_OUT_LEAKED_CREDENTIALS) {
        for (int i = 0; i < v->payload_cnt; ++i) {
            const mcbs_payload* c = &o->PL[v->payload_off + i];
            if (mark_node_as_discovered(e, c->node)) newly_nodes++;
            if (!e->gathered[c->cred]) { newly_creds++; e->gathered[c->cred] = 1; }
        }
    } else if (v->kind == MCBS_OUT_LEAKED_NODES) {
        for (int i = 0; i < v->payload_cnt; ++i)
            if (mark_node_as_discovered(e, o->PL[v->payload_off + i].node)) newly_nodes++;
    }
    reward += newly_nodes * NODE_DISCOVERED_REWARD;
    reward += newly_creds * CREDENTIAL_DISCOVERED_REWARD;
    reward -= v->cost;
    *out = result(reward, v->kind);
    out->level = v->level; out->slot_node = node; out->v = v;
    return 1;
}

/* exploit_local_vulnerability (actions.py:473-502); throws_on_invalid_actions=False semantics */
static action_result exploit_local(const oracle* o, oenv* e, int node, int local_idx) {
    action_result r;
    if (!e->node[node].agent_installed) return result(P_INVALID_ACTION, MCBS_OUT_NONE);
    process_outcome(o, e, local_idx, node, P_LOCAL_EXPLOIT_FAILED, &r);
    return r;
}

/* exploit_remote_vulnerability (actions.py:425-471) */
static action_result exploit_remote(const oracle* o, oenv* e, int source, int target, int remote_idx) {
    action_result r;
    if (!e->node[source].agent_installed) return result(P_INVALID_ACTION, MCBS_OUT_NONE);
    if (!e->node[target].tracked) return result(P_INVALID_ACTION, MCBS_OUT_NONE);
    process_outcome(o, e, (int)o->H->n_local + remote_idx, target, P_FAILED_REMOTE_EXPLOIT, &r);
    return r;
}

/* _check_service_running_and_authorized (actions.py:608-621) */
static int service_is_running(const oracle* o, const oenv* e, int svc) { return random_events(o) ? e->svc_running[svc] : (o->SV[svc].running ? 1 : 0); }
static int service_running_and_authorized(const oracle* o, const oenv* e, int target, int port, int cred) {
    const mcbs_node_static* t = &o->NS[target];
    for (int i = 0; i < t->svc_cnt; ++i) {
        const mcbs_service* sv = &o->SV[t->svc_off + i];
        if (!service_is_running(o, e, t->svc_off + i) || sv->port != port) continue;
        for (int k = 0; k < sv->allowed_cnt; ++k) if (o->AL[sv->allowed_off + k] == cred) return 1;
    }
    return 0;
}

/* __is_passing_firewall_rules (actions.py:504-515): the first rule naming the port decides; no rule = blocked */
static int list_of(const oracle* o, int node, int dir) { return (int)((o->NS[node].fw_lists >> (dir ? 16 : 0)) & 0xFFFFu); }
static int is_passing_firewall_rules(const oracle* o, const oenv* e, int node, int dir, int port_name) {
    int l = list_of(o, node, dir);
    for (int i = 0; i < e->n_fwl[l]; ++i)
        if (e->fwl[l][i].name == port_name) return e->fwl[l][i].allow ? 1 : 0;
    return 0;
}

/* connect_to_remote_machine (actions.py:524-606) */
static action_result connect_to_remote_machine(const oracle* o, oenv* e, int source, int target, int port, int cred) {
    if (!e->node[source].agent_installed) return result(P_INVALID_ACTION, MCBS_OUT_NONE);
    if (!e->node[target].tracked) return result(P_INVALID_ACTION, MCBS_OUT_NONE);
    if (!e->gathered[cred]) return result(P_INVALID_ACTION, MCBS_OUT_NONE);
    if (!is_passing_firewall_rules(o, e, source, 1, port)) return result(P_BLOCKED_BY_LOCAL_FIREWALL, MCBS_OUT_NONE);
    if (!is_passing_firewall_rules(o, e, target, 0, port)) return result(P_BLOCKED_BY_REMOTE_FIREWALL, MCBS_OUT_NONE);
    int listening = 0;                                   /* port_name in [i.name for i in target_node.services] */
    for (int i = 0; i < o->NS[target].svc_cnt; ++i) if (o->SV[o->NS[target].svc_off + i].port == port) listening = 1;
    if (!listening) return result(P_SCANNING_UNOPEN_PORT, MCBS_OUT_NONE);
    if (e->node[target].status != ST_RUNNING) return result(P_MACHINE_NOT_RUNNING, MCBS_OUT_NONE);
    if (!service_running_and_authorized(o, e, target, port, cred)) return result(P_WRONG_PASSWORD, MCBS_OUT_NONE);
    int64_t last; int already;
    mark_node_as_owned(o, e, target, 1, &last, &already);
    if (already) return result(P_REPEAT, MCBS_OUT_LATERAL_MOVE);
    track(e, target);
    return result(last == NONE_T ? (double)o->NS[target].value : 0.0, MCBS_OUT_LATERAL_MOVE);
}

/* ---------------- defender ---------------- */
/* reimage_node (actions.py:700-712) */
static void reimage_node(oenv* e, int n) {
    onode* x = &e->node[n];
    x->reimaging = 1; x->remaining = REIMAGING_DURATION;
    x->agent_installed = 0; x->privilege_level = 0; x->status = ST_IMAGING; x->last_reimaging = now(e);
}

/* on_attacker_step_taken (actions.py:714-746) */
static void on_attacker_step_taken(const oracle* o, oenv* e) {
    int N = (int)o->H->n_nodes;
    for (int n = 0; n < N; ++n) {
        onode* x = &e->node[n];
        if (!x->reimaging) continue;
        if (x->remaining > 0) x->remaining -= 1;
        else { x->status = ST_RUNNING; x->reimaging = 0; }
    }
    double total_node_weights = 0, network_node_availability = 0;
    for (int n = 0; n < N; ++n) {
        const mcbs_node_static* t = &o->NS[n];
        double total_service_weights = 0, running_service_weights = 0;
        for (int i = 0; i < t->svc_cnt; ++i) {
            const mcbs_service* sv = &o->SV[t->svc_off + i];
            total_service_weights += sv->sla_weight;
            running_service_weights += sv->sla_weight * service_is_running(o, e, t->svc_off + i);
        }
        double adjusted = (e->node[n].status == ST_RUNNING) ? (1 + running_service_weights) / (1 + total_service_weights) : 0.0;
        total_node_weights += t->sla_weight;
        network_node_availability += adjusted * t->sla_weight;
    }
    e->availability = network_node_availability / total_node_weights;
}

/* ScanAndReimageCompromisedMachines.step (defender.py:42-55); random.choices(pop, k) = pop[floor(random()*n)] k times */
static void defender_step(const oracle* o, oenv* e, draws* d) {
    if (o->cfg.defender_kind != MCBS_DEFENDER_SCAN_AND_REIMAGE) return;
    if (e->stepcount % (int)o->cfg.scan_frequency != 0) return;
    int N = (int)o->H->n_nodes, k = (int)o->cfg.scan_capacity;
    int* scanned = (int*)malloc(sizeof(int) * (size_t)(k > 0 ? k : 1));
    for (int i = 0; i < k; ++i) {
        int idx = (int)floor(next_draw(d) * (double)N);
        scanned[i] = idx < N ? idx : N - 1;
    }
    for (int i = 0; i < k; ++i) {
        int n = scanned[i];
        if (e->node[n].status == ST_RUNNING && e->node[n].agent_installed) {
            int detected = next_draw(d) <= o->cfg.scan_probability;
            if (detected && (o->NS[n].flags & MCBS_NODE_REIMAGABLE)) reimage_node(e, n);
        }
    }
    free(scanned);
}

/* ExternalRandomEvents.step (defender.py:61-66): five passes over every node, each node drawing numpy.random.random() <= 0.1
 * and, when it fires, random.choice(seq) = seq[floor(random() * len(seq))] (the harness patches both generators so that the
 * reference consumes the same doubles).  Rules compare equal on (port, permission). */
static int pick(draws* d, int n) { int i = (int)floor(next_draw(d) * (double)n); return i < n ? i : n - 1; }
static int rule_index(const oenv* e, int l, int name, int allow) {
    for (int i = 0; i < e->n_fwl[l]; ++i) if (e->fwl[l][i].name == name && (e->fwl[l][i].allow ? 1 : 0) == allow) return i;
    return -1;
}
static void remove_random_rule(oenv* e, int l, draws* d) {   /* rule = random.choice(list); list.remove(rule): the FIRST equal one goes */
    int c = pick(d, e->n_fwl[l]);
    int i = rule_index(e, l, e->fwl[l][c].name, e->fwl[l][c].allow ? 1 : 0);
    memmove(&e->fwl[l][i], &e->fwl[l][i + 1], sizeof(mcbs_fw_rule) * (size_t)(e->n_fwl[l] - i - 1));
    e->n_fwl[l] -= 1;
}
static void random_events_step(const oracle* o, oenv* e, draws* d) {
    const double p = 0.1;
    int N = (int)o->H->n_nodes, cap = (int)o->ERE->key_cap;
    for (int n = 0; n < N; ++n) {                                  /* patch_vulnerabilities_at_random (:68-75) */
        int fire = next_draw(d) <= p;
        if (fire && e->kcnt[n] > 0) {
            uint8_t* k = e->keys + (size_t)n * cap;
            int c = pick(d, e->kcnt[n]);
            memmove(k + c, k + c + 1, (size_t)(e->kcnt[n] - c - 1));
            e->kcnt[n] -= 1;
        }
    }
    for (int n = 0; n < N; ++n) {                                  /* stop_service_at_random (:77-82), stop_service (actions.py:782-787) */
        int fire = next_draw(d) <= p;
        const mcbs_node_static* t = &o->NS[n];
        if (fire && t->svc_cnt > 0) {
            int port = o->SV[t->svc_off + pick(d, t->svc_cnt)].port;
            for (int i = 0; i < t->svc_cnt; ++i) if (o->SV[t->svc_off + i].port == port) e->svc_running[t->svc_off + i] = 0;
        }
    }
    for (int n = 0; n < N; ++n) {                                  /* plant_vulnerabilities_at_random (:84-93) */
        int fire = next_draw(d) <= p, n_new = 0;
        uint8_t fresh[64];
        for (uint32_t i = 0; i < o->ERE->n_library; ++i) if (!has_key(o, e, n, o->lib_sorted[i])) fresh[n_new++] = o->lib_sorted[i];
        if (fire && n_new > 0) {
            int col = fresh[pick(d, n_new)];
            e->keys[(size_t)n * cap + e->kcnt[n]] = (uint8_t)col;
            e->kcnt[n] += 1;
        }
    }
    for (int n = 0; n < N; ++n) {                                  /* firewall_change_remove (:111-132) */
        int fire = next_draw(d) <= p, lin = list_of(o, n, 0), lout = list_of(o, n, 1);
        if (fire && e->n_fwl[lout] > 0 && e->n_fwl[lin] > 0) {
            int incoming = next_draw(d) <= 0.5;
            remove_random_rule(e, incoming ? lin : lout, d);
        } else if (fire && e->n_fwl[lout] > 0) remove_random_rule(e, lout, d);
        else if (fire && e->n_fwl[lin] > 0) remove_random_rule(e, lin, d);
    }
    for (int n = 0; n < N; ++n) {                                  /* firewall_change_add (:134-148) */
        int fire = next_draw(d) <= p;
        if (!fire) continue;
        int name = o->ERE->sample_name[pick(d, 7)];
        int incoming = next_draw(d) <= 0.5, lin = list_of(o, n, 0), lout = list_of(o, n, 1);
        if (rule_index(e, lin, name, 1) >= 0) continue;            /* both branches test the INCOMING list (:145-148) */
        int l = incoming ? lin : lout;
        if (e->n_fwl[l] >= e->cap_fwl[l]) { e->fw_overflow = 1; continue; }
        e->fwl[l][e->n_fwl[l]].name = (uint8_t)name; e->fwl[l][e->n_fwl[l]].allow = 1; e->n_fwl[l] += 1;
    }
}

/* ---------------- env level ---------------- */
static int find_external_index(const oenv* e, int node) { /* __find_external_index (env.py:603-605) */
    for (int i = 0; i < e->n_discovered; ++i) if (e->discovered[i] == node) return i;
    return -1;
}

static int owned_count(const oracle* o, const oenv* e) { /* get_nodes_with_atleast_privilegelevel(LocalUser) */
    int c = 0;
    for (uint32_t n = 0; n < o->H->n_nodes; ++n) c += e->node[n].privilege_level >= 1;
    return c;
}

static void reset_env(const oracle* o, oenv* e) { /* __reset_environment (env.py:375-394) + AgentActions.__init__ (actions.py:132-152) */
    int N = (int)o->H->n_nodes;
    e->n_tracked = e->n_discovered = e->n_cache = 0;
    e->clock = 0; e->stepcount = 0; e->done = 0; e->truncated = 0; e->episode_reward_sum = 0.0; e->availability = 1.0;
    e->last_kind = MCBS_OUT_NONE; e->last_level = 0; e->last_new_nodes = e->last_new_creds = 0; e->last_oob = 0;
    memset(e->gathered, 0, sizeof(int) * (o->H->n_cred_strings + 1));
    for (uint32_t l = 0; l < o->H->n_fw_lists; ++l) {          /* copy.deepcopy(initial_environment) keeps list aliasing */
        e->n_fwl[l] = o->FWRANGE[l * 2 + 1];
        memcpy(e->fwl[l], o->FWR + o->FWRANGE[l * 2], sizeof(mcbs_fw_rule) * (size_t)e->n_fwl[l]);
    }
    if (o->ERE) {
        memcpy(e->keys, o->own_keys, (size_t)N * o->ERE->key_cap);
        memcpy(e->kcnt, o->own_cnt, (size_t)N);
    }
    for (uint32_t k = 0; k < o->H->n_services; ++k) e->svc_running[k] = o->SV[k].running ? 1 : 0;
    e->fw_overflow = 0;
    for (int n = 0; n < N; ++n) {
        onode* x = &e->node[n];
        memset(x, 0, sizeof(*x));
        x->status = ST_RUNNING; x->last_reimaging = NONE_T; x->last_owned_at = NONE_T;
        for (int s = 0; s < MCBS_MAX_SLOTS; ++s) x->last_attack[s] = NONE_T;
        for (int k = 0; k < 4; ++k) x->tag[k] = (o->NS[n].tags0 >> k) & 1;
        x->agent_installed = (o->NS[n].flags & MCBS_NODE_INSTALLED0) ? 1 : 0;
        x->privilege_level = x->agent_installed ? 0 : o->NS[n].priv0;
    }
    for (int n = 0; n < N; ++n) if (e->node[n].agent_installed) { /* "Mark all owned nodes as discovered" */
        int64_t last; int cur;
        /* the declared initial privilege of an installed node is folded into priv0 = max(initial, LocalUser) */
        mark_node_as_owned(o, e, n, o->NS[n].priv0, &last, &cur);
    }
    for (int n = 0; n < N; ++n) if (e->node[n].agent_installed) e->discovered[e->n_discovered++] = n;
}

typedef struct {
    int32_t* scalars; int32_t* leaked; int32_t* cache_matrix; int32_t* props; int32_t* priv;
    int8_t* mask_local; int8_t* mask_remote; int8_t* mask_connect;
} oobs;

static void update_action_mask(const oracle* o, const oenv* e, const oobs* b) { /* __update_action_mask (env.py:643-677) */
    int Nm = (int)o->cfg.maximum_node_count, L = (int)o->H->n_local, R = (int)o->H->n_remote, P = (int)o->H->n_ports,
        C = (int)o->cfg.maximum_total_credentials;
    for (int si = 0; si < e->n_discovered; ++si) {
        int src = e->discovered[si];
        if (!e->node[src].agent_installed) continue;
        if (b->mask_local) for (int l = 0; l < L; ++l)
            if (random_events(o) ? vulnerability_present(o, e, src, l) : (int)((o->NS[src].local_mask >> l) & 1u)) b->mask_local[si * L + l] = 1;
        for (int ti = 0; ti < e->n_discovered; ++ti) {
            if (b->mask_remote) for (int r = 0; r < R; ++r) b->mask_remote[((size_t)si * Nm + ti) * R + r] = 1;
            if (b->mask_connect) for (int p = 0; p < P; ++p) for (int c = 0; c < e->n_cache && c < C; ++c)
                b->mask_connect[(((size_t)si * Nm + ti) * P + p) * C + c] = 1;
        }
    }
}

/* blank = 1: __get_blank_observation (env.py:753-773); else the body of __observation_reward_from_action_result
 * (env.py:859-933); reset_obs = 1: reset() (env.py:1197-1200) */
static void write_observation(const oracle* o, const oenv* e, const oobs* b, int blank, int reset_obs, int cache_before) {
    int Nm = (int)o->cfg.maximum_node_count, L = (int)o->H->n_local, R = (int)o->H->n_remote, P = (int)o->H->n_ports,
        C = (int)o->cfg.maximum_total_credentials, K = (int)o->cfg.maximum_discoverable_credentials_per_action,
        NP = (int)o->H->n_props;
    if (b->scalars) { memset(b->scalars, 0, sizeof(int32_t) * 7); b->scalars[6] = e->n_discovered; }
    if (b->leaked) memset(b->leaked, 0, sizeof(int32_t) * 4 * (size_t)K);
    if (b->cache_matrix) memset(b->cache_matrix, 0, sizeof(int32_t) * 2 * (size_t)C);
    if (b->props) for (int i = 0; i < Nm * NP; ++i) b->props[i] = 2;
    if (b->priv) memset(b->priv, 0, sizeof(int32_t) * (size_t)Nm);
    if (b->mask_local) memset(b->mask_local, 0, (size_t)Nm * L);
    if (b->mask_remote) memset(b->mask_remote, 0, (size_t)Nm * Nm * R);
    if (b->mask_connect) memset(b->mask_connect, 0, (size_t)Nm * Nm * P * C);
    if (blank) return;

    if (!reset_obs && b->scalars) {
        if (e->last_kind == MCBS_OUT_LEAKED_NODES || e->last_kind == MCBS_OUT_LEAKED_CREDENTIALS) b->scalars[0] = e->last_new_nodes;
        if (e->last_kind == MCBS_OUT_LATERAL_MOVE) b->scalars[1] = 1;
        if (e->last_kind == MCBS_OUT_CUSTOMER_DATA) b->scalars[2] = 1;
        if (e->last_kind == MCBS_OUT_PROBE_SUCCEEDED) b->scalars[3] = 2;
        if (e->last_kind == MCBS_OUT_PROBE_FAILED) b->scalars[3] = 1;
        if (e->last_kind == MCBS_OUT_PRIVILEGE_ESCALATION) b->scalars[4] = e->last_level;
        b->scalars[5] = e->n_cache;
    }
    if (!reset_obs && b->leaked && e->last_kind == MCBS_OUT_LEAKED_CREDENTIALS)
        for (int i = 0; i < e->last_new_creds && i < K; ++i) {
            const mcbs_triple* t = &o->TR[e->cache[cache_before + i]];
            b->leaked[i * 4 + 0] = 1; b->leaked[i * 4 + 1] = cache_before + i;
            b->leaked[i * 4 + 2] = find_external_index(e, t->node); b->leaked[i * 4 + 3] = t->port;
        }
    if (!reset_obs && b->cache_matrix)
        for (int i = 0; i < e->n_cache && i < C; ++i) {
            const mcbs_triple* t = &o->TR[e->cache[i]];
            b->cache_matrix[i * 2] = find_external_index(e, t->node); b->cache_matrix[i * 2 + 1] = t->port;
        }
    if (b->props) { /* __get_property_matrix over AgentActions.discovered_nodes(), zero padded (env.py:811-830) */
        for (int i = 0; i < Nm * NP; ++i) b->props[i] = 0;
        for (int i = 0; i < e->n_tracked && i < Nm; ++i)
            for (int p = 0; p < NP; ++p) b->props[i * NP + p] = e->node[e->tracked_order[i]].discovered_property[p];
    }
    if (b->priv) for (int i = 0; i < e->n_discovered && i < Nm; ++i) b->priv[i] = e->node[e->discovered[i]].privilege_level;
    update_action_mask(o, e, b);
}

typedef struct { double reward, raw; int terminated, truncated, oob, step_count; double availability; } ostep;

/* CyberBattleEnv.step (env.py:1145-1185).  Returns 0, or -1 for the reference's
 * RuntimeError("new episode must be started with env.reset()"). */
static int step_env(const oracle* o, oenv* e, uint64_t env_gid, const int32_t a[5], const double* tape, int tape_len,
                    const oobs* obs, ostep* out) {
    if (a[0] == MCBS_ACTION_SKIP && !(e->done || e->truncated)) {   /* env not stepped (attack_wrapper.py:292-308) */
        out->reward = 0; out->raw = 0; out->terminated = 0; out->truncated = 0; out->oob = 0;
        out->step_count = e->stepcount; out->availability = e->availability;
        return 0;
    }
    if (e->done || e->truncated) {
        out->reward = 0; out->raw = 0; out->terminated = e->done; out->truncated = e->truncated; out->oob = 0;
        out->step_count = e->stepcount; out->availability = e->availability;
        return -1;
    }
    e->stepcount += 1;
    int oob = 0, cache_before = e->n_cache;
    action_result r = result(0, MCBS_OUT_NONE);
    int kind = a[0];
    int L = (int)o->H->n_local, R = (int)o->H->n_remote, P = (int)o->H->n_ports;
    /* __execute_action (env.py:707-751); an index outside the identifier lists raises IndexError in the
     * reference — the engine defines it as the out-of-bound path (documented divergence) */
    if (kind == 0) {
        if (a[1] < 0 || a[1] >= e->n_discovered || a[2] < 0 || a[2] >= L) oob = 1;
        else r = exploit_local(o, e, e->discovered[a[1]], a[2]);
    } else if (kind == 1) {
        if (a[1] < 0 || a[1] >= e->n_discovered || a[2] < 0 || a[2] >= e->n_discovered || a[3] < 0 || a[3] >= R) oob = 1;
        else r = exploit_remote(o, e, e->discovered[a[1]], e->discovered[a[2]], a[3]);
    } else if (kind == 2) {
        if (a[4] < 0 || a[4] >= e->n_cache) r = result(-1, MCBS_OUT_NONE);          /* env.py:736-737 */
        else if (a[1] < 0 || a[1] >= e->n_discovered || a[2] < 0 || a[2] >= e->n_discovered || a[3] < 0 || a[3] >= P) oob = 1;
        else r = connect_to_remote_machine(o, e, e->discovered[a[1]], e->discovered[a[2]], a[3], o->TR[e->cache[a[4]]].cred);
    } else oob = 1;

    double reward;
    e->last_oob = oob;
    if (!oob) {
        /* __observation_reward_from_action_result (env.py:859-933): env-side discovery / credential cache */
        e->last_kind = r.kind; e->last_level = r.level; e->last_new_nodes = 0; e->last_new_creds = 0;
        if (r.v && r.kind == MCBS_OUT_LEAKED_NODES) {
            for (int i = 0; i < r.v->payload_cnt; ++i) {
                int n = o->PL[r.v->payload_off + i].node;
                if (find_external_index(e, n) < 0) { e->discovered[e->n_discovered++] = n; e->last_new_nodes++; }
            }
        } else if (r.v && r.kind == MCBS_OUT_LEAKED_CREDENTIALS) {
            for (int i = 0; i < r.v->payload_cnt; ++i) {
                const mcbs_payload* c = &o->PL[r.v->payload_off + i];
                if (find_external_index(e, c->node) < 0) { e->discovered[e->n_discovered++] = c->node; e->last_new_nodes++; }
                int seen = 0;
                for (int k = 0; k < e->n_cache; ++k) if (e->cache[k] == c->triple) seen = 1;
                if (!seen) { e->cache[e->n_cache++] = c->triple; e->last_new_creds++; }
            }
        }
        if (obs) write_observation(o, e, obs, 0, 0, cache_before);
        reward = r.reward;
        if (o->cfg.defender_kind == MCBS_DEFENDER_SCAN_AND_REIMAGE || random_events(o)) {
            draws d = { o, e, env_gid, tape, tape_len, 0 };
            on_attacker_step_taken(o, e);
            if (random_events(o)) random_events_step(o, e, &d);
            else defender_step(o, e, &d);
        }
        /* goals (env.py:1080-1116,1162-1169) */
        int owned = owned_count(o, e), N = (int)o->H->n_nodes, has_def = o->cfg.defender_kind == MCBS_DEFENDER_SCAN_AND_REIMAGE || random_events(o);
        int attacker_goal = 0;
        if (o->cfg.has_attacker_goal) {
            attacker_goal = 1;
            if (e->episode_reward_sum < o->cfg.goal_reward) attacker_goal = 0;
            else if (owned < (int)o->cfg.goal_own_atleast) attacker_goal = 0;
            else if ((double)owned / (double)N < o->cfg.goal_own_atleast_percent) attacker_goal = 0;
            else if (has_def && e->availability >= o->cfg.goal_low_availability) attacker_goal = 0;
        }
        int constraints_broken = has_def && e->availability < o->cfg.maintain_sla;
        int defender_goal = o->cfg.defender_goal_eviction && owned == 0;
        if (attacker_goal || constraints_broken) { e->done = 1; reward = o->cfg.winning_reward; }
        else if (defender_goal) { e->done = 1; reward = o->cfg.losing_reward; }
        else reward = reward > 0.0 ? reward : 0.0;
    } else {
        e->last_kind = MCBS_OUT_NONE; e->last_level = 0; e->last_new_nodes = e->last_new_creds = 0;
        if (obs) write_observation(o, e, obs, 1, 0, cache_before);
        reward = 0.0;
    }
    e->episode_reward_sum += reward;
    if (!e->done && o->cfg.max_episode_steps && (uint32_t)e->stepcount >= o->cfg.max_episode_steps) e->truncated = 1;
    out->reward = reward; out->raw = oob ? 0.0 : r.reward; out->terminated = e->done; out->truncated = e->truncated; out->oob = oob;
    out->step_count = e->stepcount; out->availability = e->availability;
    if ((e->done || e->truncated) && o->cfg.auto_reset) { int ep = e->episode + 1; reset_env(o, e); e->episode = ep; }
    return 0;
}

/* ---------------- learned defender: DefenderEnvWrapper.is_defender_action_valid (defend_wrapper.py:329-412) and
 * LearningDefender.executeAction (marlon/defender_agents/defender.py:31-107), acting on the live environment ---------------- */
static int rule_exists(const oenv* e, int l, int name) {
    for (int i = 0; i < e->n_fwl[l]; ++i) if (e->fwl[l][i].name == name) return 1;
    return 0;
}

static void defender_turn(const oracle* o, oenv* e, const int64_t a[12], int* valid, double* availability, int* evicted) {
    int N = (int)o->H->n_nodes, kind = (int)a[0], ok = 0, node = -1;
    static const int node_slot[5] = { 1, 2, 5, 8, 10 };
    if (kind >= 0 && kind <= 4) node = (int)a[node_slot[kind]];
    if (kind < 0) ok = 1;                                                            /* empty action: valid no-op */
    else if (kind <= 4 && node >= 0 && node < N && e->node[node].status == ST_RUNNING) {
        if (kind == 0) ok = (o->NS[node].flags & MCBS_NODE_REIMAGABLE) != 0;
        else if (kind == 1) ok = a[3] >= 0 && a[3] < 6 && rule_exists(e, list_of(o, node, a[4] ? 0 : 1), o->H->rule_name[a[3]]);
        else if (kind == 2) ok = 1;
        else if (kind == 3) ok = a[9] >= 0 && a[9] < o->NS[node].svc_cnt;
        else ok = a[11] >= 0 && a[11] < o->NS[node].svc_cnt;
    }
    on_attacker_step_taken(o, e);                                                    /* executeAction always starts with it */
    if (ok && kind >= 0) {
        if (kind == 0) reimage_node(e, node);
        else if (kind == 1) {                                                        /* block_traffic: drop every rule naming the port */
            int l = list_of(o, node, a[4] ? 0 : 1), name = o->H->rule_name[a[3]], w = 0;
            for (int i = 0; i < e->n_fwl[l]; ++i) if (e->fwl[l][i].name != name) e->fwl[l][w++] = e->fwl[l][i];
            e->n_fwl[l] = w;
        } else if (kind == 2 && a[6] >= 0 && a[6] < 6) {                             /* allow_traffic: sic, appends to INCOMING in both cases */
            int examined = list_of(o, node, a[7] ? 0 : 1), target = list_of(o, node, 0), name = o->H->rule_name[a[6]];
            if (!rule_exists(e, examined, name)) {
                e->fwl[target][e->n_fwl[target]].name = (uint8_t)name; e->fwl[target][e->n_fwl[target]].allow = 1; e->n_fwl[target]++;
            }
        }
        /* stop_service / start_service: the reference passes a ListeningService where a port name is expected, nothing matches */
    }
    *valid = ok;
    *availability = e->availability;
    *evicted = o->cfg.defender_goal_eviction && owned_count(o, e) == 0;
}

void cbo_defender_step(void* h, const int64_t* actions, uint8_t* valid, double* availability, uint8_t* evicted) {
    oracle* o = (oracle*)h;
    for (int i = 0; i < o->n_envs; ++i) {
        int v = 0, ev = 0; double av = o->env[i].availability;
        if (actions[(size_t)i * 12] > -2) defender_turn(o, &o->env[i], actions + (size_t)i * 12, &v, &av, &ev);
        if (valid) valid[i] = (uint8_t)v;
        if (availability) availability[i] = av;
        if (evicted) evicted[i] = (uint8_t)ev;
    }
}

/* DefenderEnvWrapper.observe (defend_wrapper.py:492-534) */
void cbo_defender_observe(void* h, int8_t* infected, int8_t* fw_in, int8_t* fw_out, int8_t* services) {
    oracle* o = (oracle*)h; int N = (int)o->H->n_nodes;
    for (int i = 0; i < o->n_envs; ++i) {
        const oenv* e = &o->env[i];
        for (int n = 0; n < N; ++n) {
            if (infected) infected[(size_t)i * N + n] = (int8_t)e->node[n].agent_installed;
            for (int k = 0; k < 6; ++k) {
                if (fw_in) fw_in[((size_t)i * N + n) * 6 + k] = (int8_t)rule_exists(e, list_of(o, n, 0), o->H->rule_name[k]);
                if (fw_out) fw_out[((size_t)i * N + n) * 6 + k] = (int8_t)rule_exists(e, list_of(o, n, 1), o->H->rule_name[k]);
            }
        }
        if (services) for (uint32_t k = 0; k < o->H->n_services; ++k) services[(size_t)i * o->H->n_services + k] = (int8_t)(o->SV[k].running ? 1 : 0);
    }
}

/* =============================== exported API (ctypes) =============================== */
void* cbo_create(const void* blob, size_t nbytes, const mcbs_batch_cfg* cfg) {
    if (nbytes < sizeof(mcbs_topo_header)) return NULL;
    const mcbs_topo_header* h = (const mcbs_topo_header*)blob;
    if (h->magic != MCBS_TOPO_MAGIC || h->total_bytes != nbytes || h->abi_version != MCBS_ABI_VERSION) return NULL;
    if (h->n_nodes > cfg->maximum_node_count) return NULL;
    oracle* o = (oracle*)calloc(1, sizeof(oracle));
    o->blob = (uint8_t*)malloc(nbytes);
    memcpy(o->blob, blob, nbytes);
    o->H = (const mcbs_topo_header*)o->blob;
    o->NS = (const mcbs_node_static*)(o->blob + o->H->off_node);
    o->slot_of = o->blob + o->H->off_slot_of;
    o->SL = (const mcbs_vuln_slot*)(o->blob + o->H->off_slot);
    o->PL = (const mcbs_payload*)(o->blob + o->H->off_payload);
    o->SV = (const mcbs_service*)(o->blob + o->H->off_service);
    o->AL = (const uint16_t*)(o->blob + o->H->off_allowed);
    o->TR = (const mcbs_triple*)(o->blob + o->H->off_triple);
    o->CODE = o->blob + o->H->off_code;
    o->FWR = (const mcbs_fw_rule*)(o->blob + o->H->off_fw_rule);
    o->FWRANGE = (const uint16_t*)(o->blob + o->H->off_fw_range);
    o->ERE = o->H->off_ere ? (const mcbs_ere_tables*)(o->blob + o->H->off_ere) : NULL;
    if (o->ERE) {
        const uint8_t* eb = (const uint8_t*)o->ERE;
        o->own_keys = eb + o->ERE->off_own_keys; o->own_cnt = eb + o->ERE->off_own_cnt; o->lib_sorted = eb + o->ERE->off_lib_sorted;
    }
    o->cfg = *cfg;
    o->n_envs = (int)cfg->n_envs;
    o->env = (oenv*)calloc((size_t)o->n_envs, sizeof(oenv));
    int N = (int)o->H->n_nodes;
    for (int i = 0; i < o->n_envs; ++i) {
        oenv* e = &o->env[i];
        e->node = (onode*)calloc((size_t)N, sizeof(onode));
        e->fwl = (mcbs_fw_rule**)calloc((size_t)o->H->n_fw_lists + 1, sizeof(mcbs_fw_rule*));
        e->n_fwl = (int*)calloc((size_t)o->H->n_fw_lists + 1, sizeof(int));
        e->cap_fwl = (int*)calloc((size_t)o->H->n_fw_lists + 1, sizeof(int));
        for (uint32_t l = 0; l < o->H->n_fw_lists; ++l) {        /* room for every rule the learned defender can append; the random-events
                                                                    defender's capacity is the engine's: initial + MCBS_FW_GROWTH */
            e->cap_fwl[l] = (int)o->FWRANGE[l * 2 + 1] + (cfg->defender_kind == MCBS_DEFENDER_RANDOM_EVENTS ? MCBS_FW_GROWTH : 16 * N + 64);
            e->fwl[l] = (mcbs_fw_rule*)calloc((size_t)e->cap_fwl[l] + 1, sizeof(mcbs_fw_rule));
        }
        e->keys = (uint8_t*)calloc((size_t)N * (o->ERE ? o->ERE->key_cap : 1) + 1, 1);
        e->kcnt = (uint8_t*)calloc((size_t)N + 1, 1);
        e->svc_running = (uint8_t*)calloc((size_t)o->H->n_services + 1, 1);
        e->tracked_order = (int*)calloc((size_t)N, sizeof(int));
        e->discovered = (int*)calloc((size_t)N, sizeof(int));
        e->cache = (int*)calloc((size_t)o->H->n_triples + 1, sizeof(int));
        e->gathered = (int*)calloc((size_t)o->H->n_cred_strings + 1, sizeof(int));
        e->episode = 0;
        reset_env(o, e);
    }
    return o;
}

void cbo_destroy(void* h) {
    oracle* o = (oracle*)h;
    if (!o) return;
    for (int i = 0; i < o->n_envs; ++i) {
        for (uint32_t l = 0; l < o->H->n_fw_lists; ++l) free(o->env[i].fwl[l]);
        free(o->env[i].fwl); free(o->env[i].n_fwl); free(o->env[i].cap_fwl); free(o->env[i].keys); free(o->env[i].kcnt); free(o->env[i].svc_running);
        free(o->env[i].node); free(o->env[i].tracked_order); free(o->env[i].discovered); free(o->env[i].cache); free(o->env[i].gathered);
    }
    free(o->env); free(o->blob); free(o);
}

/* explicit reset = a new episode (same convention as mcbs_reset: the episode index feeds the Philox counter) */
void cbo_reset(void* h, int env) { oracle* o = (oracle*)h; int ep = o->env[env].episode + 1; reset_env(o, &o->env[env]); o->env[env].episode = ep; }

static void obs_slice(const oracle* o, const mcbs_obs_buffers* b, int env, oobs* s) {
    size_t Nm = o->cfg.maximum_node_count, L = o->H->n_local, R = o->H->n_remote, P = o->H->n_ports,
           C = o->cfg.maximum_total_credentials, K = o->cfg.maximum_discoverable_credentials_per_action, NP = o->H->n_props;
    size_t i = (size_t)env;
    s->scalars = b->scalars ? b->scalars + i * 7 : NULL;
    s->leaked = b->leaked_credentials ? b->leaked_credentials + i * K * 4 : NULL;
    s->cache_matrix = b->credential_cache_matrix ? b->credential_cache_matrix + i * C * 2 : NULL;
    s->props = b->discovered_nodes_properties ? b->discovered_nodes_properties + i * Nm * NP : NULL;
    s->priv = b->nodes_privilegelevel ? b->nodes_privilegelevel + i * Nm : NULL;
    s->mask_local = b->mask_local ? b->mask_local + i * Nm * L : NULL;
    s->mask_remote = b->mask_remote ? b->mask_remote + i * Nm * Nm * R : NULL;
    s->mask_connect = b->mask_connect ? b->mask_connect + i * Nm * Nm * P * C : NULL;
}

/* reset observation of one env (env.py:1197-1200) or re-observation of the current state */
void cbo_observe(void* h, int env, const mcbs_obs_buffers* b, int reset_obs) {
    oracle* o = (oracle*)h; oobs s; obs_slice(o, b, env, &s);
    const oenv* e = &o->env[env];
    write_observation(o, e, &s, (!reset_obs && e->last_oob), reset_obs, e->n_cache - e->last_new_creds);
}

/* Step every env once.  actions [E,5]; outputs [E]; tape [E, draws_per_step] or NULL; obs buffers (host) or NULL.
 * Returns the number of envs that were stepped after done (the reference would raise for them). */
int cbo_step(void* h, const int32_t* actions, double* reward, uint8_t* terminated, uint8_t* truncated, uint8_t* oob,
             int32_t* step_count, double* availability, double* raw_reward, const double* tape, int draws_per_step,
             const mcbs_obs_buffers* obs) {
    oracle* o = (oracle*)h; int errors = 0;
    for (int i = 0; i < o->n_envs; ++i) {
        ostep r; oobs s;
        if (obs) obs_slice(o, obs, i, &s);
        if (step_env(o, &o->env[i], o->cfg.env_id_base + (uint64_t)i, actions + (size_t)i * 5,
                     tape ? tape + (size_t)i * draws_per_step : NULL, draws_per_step, obs ? &s : NULL, &r) != 0) errors++;
        if (reward) reward[i] = r.reward;
        if (terminated) terminated[i] = (uint8_t)r.terminated;
        if (truncated) truncated[i] = (uint8_t)r.truncated;
        if (oob) oob[i] = (uint8_t)r.oob;
        if (step_count) step_count[i] = r.step_count;
        if (availability) availability[i] = r.availability;
        if (raw_reward) raw_reward[i] = r.raw;
    }
    return errors;
}

/* Timing helper for bench.py's cpu_baseline leg: K consecutive steps of envs [env_lo, env_hi), env-major (every env's
 * state stays cache-resident for its K steps: the CPU's best case), rewards summed and `terminated` flags counted per env.
 * actions [K, E, 5].
 * Envs never interact, so disjoint ranges may run on different threads of the same handle. */
int cbo_run(void* h, const int32_t* actions, int K, int env_lo, int env_hi, double* reward_sum, int32_t* episodes_ended) {
    oracle* o = (oracle*)h; int errors = 0;
    for (int i = env_lo; i < env_hi; ++i) {
        double acc = 0.0; int32_t ended = 0;
        for (int t = 0; t < K; ++t) {
            ostep r;
            if (step_env(o, &o->env[i], o->cfg.env_id_base + (uint64_t)i, actions + ((size_t)t * o->n_envs + (size_t)i) * 5,
                         NULL, 0, NULL, &r) != 0) errors++;
            acc += r.reward;
            ended += r.terminated != 0;          /* (truncations are not counted: bench.py compares the `terminated` flags) */
        }
        reward_sum[i] = acc;
        if (episodes_ended) episodes_ended[i] = ended;
    }
    return errors;
}

/* Actuator-level entry points (AgentActions without the gym env), to replay commandcontrol_test.py and
 * actions_test.py.  Nodes by network index, vulnerabilities by identifier-list index, credential by string id.
 * out[0] = reward, out[1] = outcome kind. */
void cbo_exploit_local(void* h, int env, int node, int local_idx, double out[2]) {
    oracle* o = (oracle*)h; action_result r = exploit_local(o, &o->env[env], node, local_idx); out[0] = r.reward; out[1] = r.kind;
}
void cbo_exploit_remote(void* h, int env, int source, int target, int remote_idx, double out[2]) {
    oracle* o = (oracle*)h; action_result r = exploit_remote(o, &o->env[env], source, target, remote_idx); out[0] = r.reward; out[1] = r.kind;
}
void cbo_connect(void* h, int env, int source, int target, int port, int cred, double out[2]) {
    oracle* o = (oracle*)h; action_result r = connect_to_remote_machine(o, &o->env[env], source, target, port, cred); out[0] = r.reward; out[1] = r.kind;
}
int cbo_check_prerequisites(void* h, int env, int node, int vuln_col) {
    oracle* o = (oracle*)h; int s = o->slot_of[(size_t)node * (o->H->n_local + o->H->n_remote) + vuln_col];
    return s == 0xFF ? -1 : check_prerequisites(o, &o->env[env], node, slot(o, node, s));
}
int cbo_node_has_tag(void* h, int env, int node, int level) { return ((oracle*)h)->env[env].node[node].tag[level]; }
/* defender actuator, for scripted-defender traces */
void cbo_reimage_node(void* h, int env, int node) { reimage_node(&((oracle*)h)->env[env], node); }

/* Canonical state dump, same record layout as mcbs_get_state (include/mcbs.h). */
size_t cbo_state_record_bytes(void* h) {
    oracle* o = (oracle*)h;
    size_t n = sizeof(mcbs_state_header) + sizeof(mcbs_state_node) * o->H->n_nodes + 2 * o->H->n_nodes + 2 * (size_t)o->cfg.maximum_total_credentials;
    return (n + 15) & ~(size_t)15;
}
void cbo_get_state(void* h, void* buf) {
    oracle* o = (oracle*)h; size_t rb = cbo_state_record_bytes(h); int N = (int)o->H->n_nodes;
    memset(buf, 0, rb * (size_t)o->n_envs);
    for (int i = 0; i < o->n_envs; ++i) {
        const oenv* e = &o->env[i];
        uint8_t* p = (uint8_t*)buf + rb * (size_t)i;
        mcbs_state_header* sh = (mcbs_state_header*)p;
        sh->step_count = (uint32_t)e->stepcount; sh->done = (uint32_t)e->done; sh->truncated = (uint32_t)e->truncated; sh->episode = (uint32_t)e->episode;
        sh->n_discovered = (uint32_t)e->n_discovered; sh->n_creds = (uint32_t)e->n_cache;
        sh->last_outcome_kind = (uint32_t)e->last_kind; sh->last_escalation = (uint32_t)e->last_level;
        sh->last_new_nodes = (uint32_t)e->last_new_nodes; sh->last_new_creds = (uint32_t)e->last_new_creds; sh->last_oob = (uint32_t)e->last_oob;
        sh->cum_reward = e->episode_reward_sum; sh->availability = e->availability;
        mcbs_state_node* sn = (mcbs_state_node*)(p + sizeof(mcbs_state_header));
        for (int n = 0; n < N; ++n) {
            const onode* x = &e->node[n];
            for (int q = 0; q < MCBS_MAX_PROPS; ++q) if (x->discovered_property[q]) sn[n].discovered_props |= (uint64_t)1 << q;
            for (int s = 0; s < MCBS_MAX_SLOTS; ++s) if (x->last_attack[s] != NONE_T) {
                sn[n].attacked_ever |= 1u << s;
                if (x->last_reimaging == NONE_T || x->last_attack[s] >= x->last_reimaging) sn[n].attacked_since |= 1u << s;
            }
            sn[n].discovered = (uint8_t)x->tracked; sn[n].installed = (uint8_t)x->agent_installed;
            sn[n].ever_owned = (uint8_t)(x->last_owned_at != NONE_T); sn[n].running = (uint8_t)(x->status == ST_RUNNING);
            sn[n].privilege = (uint8_t)x->privilege_level;
            sn[n].tags = (uint8_t)(x->tag[0] | x->tag[1] << 1 | x->tag[2] << 2 | x->tag[3] << 3);
            sn[n].countdown = (uint8_t)(x->reimaging ? x->remaining : 0);
        }
        uint16_t* order = (uint16_t*)(p + sizeof(mcbs_state_header) + sizeof(mcbs_state_node) * (size_t)N);
        for (int k = 0; k < N; ++k) order[k] = k < e->n_discovered ? (uint16_t)e->discovered[k] : 0xFFFF;
        uint16_t* cc = order + N;
        for (uint32_t k = 0; k < o->cfg.maximum_total_credentials; ++k) cc[k] = (int)k < e->n_cache ? (uint16_t)e->cache[k] : 0xFFFF;
    }
}
