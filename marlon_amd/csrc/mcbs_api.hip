// mcbs_api.hip — host side of the C ABI declared in include/mcbs.h (libmcbs.so).
// One translation unit: the kernels are included below.  Build: marlon_amd/csrc/Makefile (hipcc, gfx950).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "mcbs.h"
#include "mcbs_device.h"
#include "mcbs_step.hip"
#include "mcbs_step_coop.hip"
#include "mcbs_obs.hip"
#include "mcbs_aux.hip"
#include "mcbs_defend.hip"
#include "mcbs_logits.hip"
#include "mcbs_wrapper_fused.hip"

using namespace mcbs;

static thread_local char g_err[512] = "";

// Calls that must address a particular device (allocation, synchronous copies) select it for their own duration only: the
// caller's current device (PyTorch's, in a single process driving several GPUs) is put back on every exit path.
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device); else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) return fail(MCBS_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e));      \
    } while (0)

// Every entry point that launches on (or synchronises with) a batch's GPU runs with that GPU selected: a single process whose current
// device differs from the batch's (PyTorch driving several GPUs) would otherwise launch on the wrong device or fail with an invalid
// handle.  hipGetDevice is a thread-local read; hipSetDevice is only called when the devices differ, and the caller's device is put back.
#define MCBS_ON_DEVICE(b)                                                                                                      \
    DeviceGuard _dev_guard((b)->cfg.device);                                                                                   \
    if (_dev_guard.err != hipSuccess) return fail(MCBS_EHIP, "cannot select device %d: %s", (b)->cfg.device, hipGetErrorString(_dev_guard.err))

static FastDiv fast_div_host(uint32_t d);
struct HotLayout { uint32_t node, desc, payload, auth, auth_words, triple, avail, fwlist, bytes; };

struct mcbs_topology {
    std::vector<uint8_t> host;
    std::vector<uint8_t> hot_host;  // step kernel's view of the tables (mcbs_device.h "hot image")
    HotLayout hot{};
    uint8_t* dev = nullptr;
    uint8_t* hot_dev = nullptr;
    int32_t device = 0;
    const mcbs_topo_header* H() const { return reinterpret_cast<const mcbs_topo_header*>(host.data()); }
};

struct mcbs_batch {
    const mcbs_topology* topo = nullptr;
    mcbs_batch_cfg cfg{};
    DevState S{};
    Topo T{};
    StepCfg C{};
    StepCfg* C_dev = nullptr;       // device copy read by the step kernel through the scalar cache
    uint32_t* ere_lists_dev = nullptr;
    // developer switches, read ONCE at batch creation (getenv on every launch costs more than the launch itself)
    bool lds_topo = false, no_fused_masks = false, slow_masks = false, no_row_masks = false;
    bool no_fused_defender_obs = false;   // MCBS_NO_FUSED_DEFENDER_OBS=1: the learned defender's observation as a launch of its own
    bool no_quad_obs = false;       // MCBS_NO_QUAD_OBS=1: a wavefront per env for small topologies' observations with mask fields (rounds 1-2)
    bool force_quad_obs = false;    // MCBS_QUAD_OBS=1: obs_quad_kernel also where mask rows are whole cache lines
    bool no_block_masks = false;    // MCBS_NO_BLOCK_MASKS=1: round 2's fused mask writers (rows switched on / off per chunk)
    bool no_fused_wrapper = false;  // MCBS_NO_FUSED_WRAPPER=1: mcbs_attacker_wrapper_step keeps its three launches (tests step both)
    size_t disc_stride = 0;         // mcbs_set_mask_discrete_stride: bytes between two envs' rows of mask_discrete (0: dense)
    bool coop = false;              // mcbs_step runs the G-lanes-per-env kernel (mcbs_step_coop.hip): more than 64 nodes, sets of 2 or 4 words
    uint32_t step_block_override = 0;
    uint8_t* arena = nullptr;       // every per-env column + bodies + init body, one allocation
    size_t arena_bytes = 0;
    ObsDigest* digest = nullptr;
    ObsDigest* reset_digest = nullptr;   // the digest of a freshly reset env, captured by the first observation after a whole-batch reset
    bool all_fresh = true, reset_digest_ok = false;
    // what mcbs_mask_logits may trust: 0 = no observation has written the per-env digests since the batch was created / wholly reset /
    // given a state (mcbs_set_state); 1 = every env's digest is the one its last observation wrote; 2 = some envs were reset by mask
    // since (their digests are stale until mcbs_observe_masked has re-observed them)
    int digest_state = 0;
    const double* tape = nullptr;
    uint32_t tape_dps = 0;
    unsigned long long* stamps = nullptr;   // diagnostic builds: device buffer [waves][8]
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;     // pairs (start, stop) per launch
    size_t ev_used = 0;
    double timed_ms = 0.0;
    uint64_t timed_launches = 0;
};

extern "C" const char* mcbs_last_error(void) { return g_err; }
extern "C" uint32_t mcbs_abi_version(void) { return MCBS_ABI_VERSION; }

// ------------------------------------------------------------------ topology
static int check_section(const mcbs_topo_header* h, uint32_t off, size_t bytes, const char* name) {
    if (off % 16u || (size_t)off + bytes > h->total_bytes) return fail(MCBS_EINVAL, "topology blob: section %s out of range", name);
    return MCBS_OK;
}

extern "C" int mcbs_topology_create(const void* blob, size_t nbytes, int32_t device, mcbs_topology** out) {
    if (!blob || !out) return fail(MCBS_EINVAL, "null argument");
    if (nbytes < sizeof(mcbs_topo_header)) return fail(MCBS_EINVAL, "topology blob too small");
    const mcbs_topo_header* h = static_cast<const mcbs_topo_header*>(blob);
    if (h->magic != MCBS_TOPO_MAGIC) return fail(MCBS_EINVAL, "topology blob: bad magic");
    if (h->abi_version != MCBS_ABI_VERSION) return fail(MCBS_EINVAL, "topology blob: ABI version %u, library %u", h->abi_version, MCBS_ABI_VERSION);
    if (h->total_bytes != nbytes || h->header_bytes != sizeof(mcbs_topo_header)) return fail(MCBS_EINVAL, "topology blob: size mismatch");
    if (h->n_nodes == 0 || h->n_nodes > MCBS_MAX_NODES) return fail(MCBS_ELIMIT, "n_nodes %u outside 1..%d", h->n_nodes, MCBS_MAX_NODES);
    if (h->n_ports == 0 || h->n_ports > MCBS_MAX_PORTS || h->n_props > 60u || h->max_slots == 0 || h->max_slots > MCBS_MAX_SLOTS ||
        h->n_local == 0 || h->n_local > MCBS_MAX_LOCAL_VULNS || h->n_remote == 0 || h->n_cred_strings > MCBS_MAX_CRED_STRINGS ||
        h->n_triples > MCBS_MAX_TRIPLES)
        return fail(MCBS_ELIMIT, "topology exceeds an engine limit");
    int rc;
    if ((rc = check_section(h, h->off_node, sizeof(mcbs_node_static) * h->n_nodes, "node"))) return rc;
    if ((rc = check_section(h, h->off_slot_of, (size_t)h->n_nodes * (h->n_local + h->n_remote), "slot_of"))) return rc;
    if ((rc = check_section(h, h->off_slot, sizeof(mcbs_vuln_slot) * h->n_nodes * h->max_slots, "slot"))) return rc;
    if ((rc = check_section(h, h->off_payload, sizeof(mcbs_payload) * h->n_payload, "payload"))) return rc;
    if ((rc = check_section(h, h->off_service, sizeof(mcbs_service) * h->n_services, "service"))) return rc;
    if ((rc = check_section(h, h->off_allowed, sizeof(uint16_t) * h->n_allowed, "allowed"))) return rc;
    if ((rc = check_section(h, h->off_triple, sizeof(mcbs_triple) * h->n_triples, "triple"))) return rc;
    if ((rc = check_section(h, h->off_init_order, h->n_nodes, "init_order"))) return rc;
    // every index the kernels will dereference is range-checked here, once, on the host
    const uint8_t* b = static_cast<const uint8_t*>(blob);
    const mcbs_node_static* ns = reinterpret_cast<const mcbs_node_static*>(b + h->off_node);
    const mcbs_vuln_slot* sl = reinterpret_cast<const mcbs_vuln_slot*>(b + h->off_slot);
    const mcbs_payload* pl = reinterpret_cast<const mcbs_payload*>(b + h->off_payload);
    const mcbs_service* sv = reinterpret_cast<const mcbs_service*>(b + h->off_service);
    const uint8_t* so = b + h->off_slot_of;
    for (uint32_t n = 0; n < h->n_nodes; ++n) {
        if ((uint32_t)ns[n].svc_off + ns[n].svc_cnt > h->n_services) return fail(MCBS_EINVAL, "node %u: service range", n);
        if (ns[n].n_slots > h->max_slots) return fail(MCBS_EINVAL, "node %u: slot count", n);
        for (uint32_t c = 0; c < h->n_local + h->n_remote; ++c) {
            const uint8_t s = so[(size_t)n * (h->n_local + h->n_remote) + c];
            if (s != 0xFF && s >= ns[n].n_slots) return fail(MCBS_EINVAL, "node %u: slot_of out of range", n);
        }
        for (uint32_t s = 0; s < ns[n].n_slots; ++s) {
            const mcbs_vuln_slot& v = sl[(size_t)n * h->max_slots + s];
            if ((size_t)v.payload_off + v.payload_cnt > h->n_payload) return fail(MCBS_EINVAL, "node %u slot %u: payload range", n, s);
            if (v.level > 3 || v.kind > MCBS_OUT_OTHER) return fail(MCBS_EINVAL, "node %u slot %u: bad kind/level", n, s);
            if (v.payload_cnt > 1023) return fail(MCBS_ELIMIT, "node %u slot %u: payload too long", n, s);
        }
    }
    for (uint32_t i = 0; i < h->n_payload; ++i)
        if (pl[i].node >= h->n_nodes || (h->n_cred_strings && pl[i].cred >= h->n_cred_strings && pl[i].cred != 0) ||
            (h->n_triples && pl[i].triple >= h->n_triples && pl[i].triple != 0) || pl[i].port >= h->n_ports)
            return fail(MCBS_EINVAL, "payload %u out of range", i);
    for (uint32_t i = 0; i < h->n_services; ++i)
        if ((uint32_t)sv[i].allowed_off + sv[i].allowed_cnt > h->n_allowed || sv[i].port >= h->n_ports)
            return fail(MCBS_EINVAL, "service %u out of range", i);
    const mcbs_triple* tr = reinterpret_cast<const mcbs_triple*>(b + h->off_triple);
    for (uint32_t i = 0; i < h->n_triples; ++i)
        if (tr[i].node >= h->n_nodes || tr[i].cred >= h->n_cred_strings || tr[i].port >= h->n_ports)
            return fail(MCBS_EINVAL, "triple %u out of range", i);
    if ((rc = check_section(h, h->off_fw_list0, 2u * (size_t)h->n_fw_lists, "fw_list0"))) return rc;
    if ((rc = check_section(h, h->off_fw_range, 2u * sizeof(uint16_t) * (size_t)h->n_fw_lists, "fw_range"))) return rc;
    if ((rc = check_section(h, h->off_fw_rule, sizeof(mcbs_fw_rule) * (size_t)h->n_fw_rules, "fw_rule"))) return rc;
    if (h->n_names > 256u) return fail(MCBS_ELIMIT, "more than 256 firewall port names");
    {   // every rule list lies inside the rule array and every rule names a known port (mcbs_batch_create and the random-events
        // kernels index with these)
        const uint16_t* fr = reinterpret_cast<const uint16_t*>(b + h->off_fw_range);
        const mcbs_fw_rule* fwr = reinterpret_cast<const mcbs_fw_rule*>(b + h->off_fw_rule);
        for (uint32_t l = 0; l < h->n_fw_lists; ++l)
            if ((uint32_t)fr[2 * l] + fr[2 * l + 1] > h->n_fw_rules) return fail(MCBS_EINVAL, "firewall rule list %u out of range", l);
        for (uint32_t i = 0; i < h->n_fw_rules; ++i)
            if (fwr[i].name >= h->n_names) return fail(MCBS_EINVAL, "firewall rule %u: port name out of range", i);
        for (int i = 0; i < 6; ++i)
            if (h->rule_port[i] != 0xFFu && h->rule_port[i] >= h->n_ports) return fail(MCBS_EINVAL, "rule_port[%d] out of range", i);
    }
    for (uint32_t n = 0; n < h->n_nodes; ++n)
        if ((ns[n].fw_lists & 0xFFFFu) >= h->n_fw_lists || (ns[n].fw_lists >> 16) >= h->n_fw_lists) return fail(MCBS_EINVAL, "node %u: firewall list id", n);
    if (h->off_ere) {          // ExternalRandomEvents tables: every index a kernel will use as a bit position or an array subscript
        if ((rc = check_section(h, h->off_ere, sizeof(mcbs_ere_tables), "ere"))) return rc;
        const mcbs_ere_tables* et = reinterpret_cast<const mcbs_ere_tables*>(b + h->off_ere);
        const uint32_t W = h->n_local + h->n_remote;
        const size_t room = h->total_bytes - h->off_ere;
        if (W > 64u || et->n_library > W || et->key_cap == 0 || et->key_cap > 255u ||
            (size_t)et->off_own_keys + (size_t)et->key_cap * h->n_nodes > room || (size_t)et->off_own_cnt + h->n_nodes > room ||
            (size_t)et->off_lib_sorted + et->n_library > room)
            return fail(MCBS_EINVAL, "topology blob: random-events tables out of range");
        const uint8_t* eb = reinterpret_cast<const uint8_t*>(et);
        for (uint32_t n = 0; n < h->n_nodes; ++n) {
            const uint32_t cnt = eb[et->off_own_cnt + n];
            if (cnt + et->n_library > et->key_cap) return fail(MCBS_EINVAL, "node %u: vulnerability key list longer than its capacity", n);
            for (uint32_t k = 0; k < cnt; ++k)
                if (eb[et->off_own_keys + (size_t)n * et->key_cap + k] >= W) return fail(MCBS_EINVAL, "node %u: vulnerability key out of range", n);
        }
        for (uint32_t i = 0; i < et->n_library; ++i) if (eb[et->off_lib_sorted + i] >= W) return fail(MCBS_EINVAL, "library column out of range");
        for (int i = 0; i < 7; ++i) if (et->sample_name[i] >= h->n_names) return fail(MCBS_EINVAL, "sample port name out of range");
        if (W < 64u && (et->lib_cols >> W)) return fail(MCBS_EINVAL, "library column mask out of range");
    }
    const uint8_t* io = b + h->off_init_order;
    if (h->n_init_owned > h->n_nodes) return fail(MCBS_EINVAL, "init order");
    for (uint32_t i = 0; i < h->n_init_owned; ++i) if (io[i] >= h->n_nodes) return fail(MCBS_EINVAL, "init order");

    mcbs_topology* t = new (std::nothrow) mcbs_topology();
    if (!t) return fail(MCBS_ENOMEM, "out of memory");
    t->host.assign(b, b + nbytes);
    t->device = device;
    {   // hot image: 32-byte node records, flattened (node, column) descriptors, then the list sections verbatim
        const uint32_t N = h->n_nodes, W = h->n_local + h->n_remote;
        HotLayout& L = t->hot;
        uint32_t off = 0;
        auto take = [&](size_t bytes) { uint32_t o = off; off = (uint32_t)((off + bytes + 15) / 16 * 16); return o; };
        L.node = take(sizeof(HotNode) * N);
        // every section holds at least one (zero) record: the step kernel looks tables up with clamped indices for every lane
        L.desc = take(sizeof(HotDesc) * ((size_t)N * W + 1));
        L.payload = take(sizeof(mcbs_payload) * (h->n_payload + 1));
        // authorisation table replacing the service / allowed-credential lists (actions.py:608-621): per (node, port) the set
        // of credential strings that some RUNNING service on that port accepts (service state never changes, mcbs_defend.hip)
        // ... indexed by the TRIPLE id of the cached credential the action names (bit t: the credential string of triple t is accepted),
        // so that the look-up does not wait for the triple -> credential-string load
        const uint32_t P1 = h->n_ports ? h->n_ports : 1u;
        L.auth_words = (h->n_triples + 63u) / 64u ? (h->n_triples + 63u) / 64u : 1u;
        L.auth = take(sizeof(uint64_t) * (size_t)N * P1 * L.auth_words);
        L.triple = take(sizeof(mcbs_triple) * (h->n_triples + 1));
        L.avail = take(sizeof(double) * N);
        L.fwlist = take(sizeof(uint32_t) * N);
        L.bytes = off;
        t->hot_host.assign(off, 0);
        uint8_t* hb = t->hot_host.data();
        for (uint32_t n = 0; n < N; ++n) {
            HotNode hn{};
            hn.props = ns[n].props; hn.value = ns[n].value; hn.fw_in_allow = ns[n].fw_in_allow; hn.fw_out_allow = ns[n].fw_out_allow;
            hn.listen = ns[n].listen; hn.svc_off = ns[n].svc_off; hn.svc_cnt = ns[n].svc_cnt; hn.flags = ns[n].flags;
            memcpy(hb + L.node + sizeof(HotNode) * n, &hn, sizeof(hn));
            memcpy(hb + L.avail + sizeof(double) * n, &ns[n].avail_term, sizeof(double));
            memcpy(hb + L.fwlist + sizeof(uint32_t) * n, &ns[n].fw_lists, sizeof(uint32_t));
            for (uint32_t c = 0; c < W; ++c) {
                HotDesc d{};
                const uint8_t s = so[(size_t)n * W + c];
                d.kind = 0xFF;
                if (s != 0xFF) {
                    const mcbs_vuln_slot& v = sl[(size_t)n * h->max_slots + s];
                    d.cost = v.cost; d.probe_mask = v.probe_mask; d.payload_off = v.payload_off; d.payload_cnt = v.payload_cnt;
                    d.precond_tt = v.precond_tt; d.kind = v.kind; d.level = v.level; d.slot = s;
                    for (uint32_t i = 0; i < 4u && i < v.payload_cnt; ++i) d.inline_payload[i] = pl[v.payload_off + i];
                }
                memcpy(hb + L.desc + sizeof(HotDesc) * ((size_t)n * W + c), &d, sizeof(d));
            }
        }
        memcpy(hb + L.payload, pl, sizeof(mcbs_payload) * h->n_payload);
        const uint16_t* allowed = reinterpret_cast<const uint16_t*>(b + h->off_allowed);
        uint64_t* auth = reinterpret_cast<uint64_t*>(hb + L.auth);
        for (uint32_t n = 0; n < N; ++n)
            for (uint32_t i = ns[n].svc_off; i < (uint32_t)ns[n].svc_off + ns[n].svc_cnt; ++i) {
                if (!sv[i].running) continue;
                for (uint32_t k = 0; k < sv[i].allowed_cnt; ++k) {
                    const uint32_t c = allowed[sv[i].allowed_off + k];
                    for (uint32_t t3 = 0; t3 < h->n_triples; ++t3)         // every cached credential that carries this string
                        if (tr[t3].cred == c) auth[((size_t)n * P1 + sv[i].port) * L.auth_words + (t3 >> 6)] |= 1ull << (t3 & 63u);
                }
            }
        memcpy(hb + L.triple, tr, sizeof(mcbs_triple) * h->n_triples);
    }
    DeviceGuard guard(device);
    hipError_t e = guard.err;
    if (e == hipSuccess) e = hipMalloc(&t->dev, nbytes);
    if (e == hipSuccess) e = hipMemcpy(t->dev, blob, nbytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&t->hot_dev, t->hot_host.size());
    if (e == hipSuccess) e = hipMemcpy(t->hot_dev, t->hot_host.data(), t->hot_host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (t->dev) (void)hipFree(t->dev);
        if (t->hot_dev) (void)hipFree(t->hot_dev);
        delete t;
        return fail(MCBS_EHIP, "topology upload failed: %s", hipGetErrorString(e));
    }
    *out = t;
    return MCBS_OK;
}

extern "C" void mcbs_topology_destroy(mcbs_topology* t) {
    if (!t) return;
    DeviceGuard guard(t->device);
    if (t->dev) (void)hipFree(t->dev);
    if (t->hot_dev) (void)hipFree(t->hot_dev);
    delete t;
}

// ------------------------------------------------------------------ batch
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" int mcbs_batch_create(const mcbs_topology* topo, const mcbs_batch_cfg* cfg, mcbs_batch** out) {
    if (!topo || !cfg || !out) return fail(MCBS_EINVAL, "null argument");
    if (cfg->abi_version != MCBS_ABI_VERSION) return fail(MCBS_EINVAL, "cfg ABI version %u, library %u", cfg->abi_version, MCBS_ABI_VERSION);
    const mcbs_topo_header* h = topo->H();
    if (cfg->n_envs == 0) return fail(MCBS_EINVAL, "n_envs must be positive");
    if (cfg->device != topo->device) return fail(MCBS_EINVAL, "batch and topology are on different devices");
    // CyberBattleEnv.validate_environment (cyberbattle_env.py:416-435)
    if (h->n_nodes > cfg->maximum_node_count)
        return fail(MCBS_EINVAL, "Network node count (%u) exceeds the specified limit of %u.", h->n_nodes, cfg->maximum_node_count);
    if (cfg->maximum_node_count > MCBS_MAX_NODES) return fail(MCBS_ELIMIT, "maximum_node_count %u > %d", cfg->maximum_node_count, MCBS_MAX_NODES);
    if (h->max_leak_per_action > cfg->maximum_discoverable_credentials_per_action)
        return fail(MCBS_EINVAL, "Some action in the environment returns %u credentials which exceeds the maximum number of discoverable credentials of %u",
                    h->max_leak_per_action, cfg->maximum_discoverable_credentials_per_action);
    if (cfg->maximum_total_credentials == 0 || cfg->maximum_total_credentials > 65535u) return fail(MCBS_ELIMIT, "maximum_total_credentials out of range");
    if (h->n_triples > cfg->maximum_total_credentials)
        return fail(MCBS_ELIMIT, "the topology can leak %u distinct credentials but maximum_total_credentials is %u "
                    "(the reference would overflow its observation space)", h->n_triples, cfg->maximum_total_credentials);
    if (cfg->maximum_discoverable_credentials_per_action > 1023u) return fail(MCBS_ELIMIT, "maximum_discoverable_credentials_per_action too large");
    if (cfg->defender_kind > MCBS_DEFENDER_RANDOM_EVENTS) return fail(MCBS_EINVAL, "unknown defender kind");
    const bool random_events = cfg->defender_kind == MCBS_DEFENDER_RANDOM_EVENTS;
    if (random_events && !h->off_ere) return fail(MCBS_EINVAL, "the topology blob carries no ExternalRandomEvents tables (off_ere)");
    if (cfg->defender_kind == MCBS_DEFENDER_SCAN_AND_REIMAGE && cfg->scan_frequency == 0) return fail(MCBS_EINVAL, "scan_frequency must be positive");
    if (cfg->rng_kind > MCBS_RNG_TAPE) return fail(MCBS_EINVAL, "unknown rng kind");
    if (h->n_cred_strings > MCBS_MAX_CRED_STRINGS || h->n_triples > MCBS_MAX_TRIPLES)
        return fail(MCBS_ELIMIT, "at most %d credential strings / %d triples (topology has %u / %u)", MCBS_MAX_CRED_STRINGS, MCBS_MAX_TRIPLES,
                    h->n_cred_strings, h->n_triples);

    mcbs_batch* b = new (std::nothrow) mcbs_batch();
    if (!b) return fail(MCBS_ENOMEM, "out of memory");
    b->topo = topo;
    b->cfg = *cfg;
    b->lds_topo = getenv("MCBS_LDS_TOPO") != nullptr; b->no_fused_masks = getenv("MCBS_NO_FUSED_MASKS") != nullptr;
    b->slow_masks = getenv("MCBS_SLOW_MASKS") != nullptr; b->no_row_masks = getenv("MCBS_NO_ROW_MASKS") != nullptr;
    b->no_fused_wrapper = getenv("MCBS_NO_FUSED_WRAPPER") != nullptr; b->no_block_masks = getenv("MCBS_NO_BLOCK_MASKS") != nullptr;
    b->no_quad_obs = getenv("MCBS_NO_QUAD_OBS") != nullptr; b->no_fused_defender_obs = getenv("MCBS_NO_FUSED_DEFENDER_OBS") != nullptr; b->force_quad_obs = getenv("MCBS_QUAD_OBS") != nullptr;
    if (const char* ov = getenv("MCBS_STEP_BLOCK")) b->step_block_override = (uint32_t)atoi(ov);   // experiments only (64, 128 or 256)
    const uint32_t E = cfg->n_envs, N = h->n_nodes;
    DevState& S = b->S;
    S.E = E; S.N = N; S.NW = (N + 63) / 64;
    S.SW = (h->n_cred_strings + 63) / 64; if (!S.SW) S.SW = 1;
    S.TW = (h->n_triples + 63) / 64; if (!S.TW) S.TW = 1;
    S.Cmax = cfg->maximum_total_credentials;
    S.off_disc = 0;                                                             // u8 discovery order, >= 16 bytes
    // both lists have one slack entry past their capacity: the step kernel appends unconditionally and only advances the count
    S.off_cred = (uint32_t)align_up((size_t)N + 1, 16);                         // u16 credential cache, >= 32 bytes
    S.off_rows = (uint32_t)align_up((size_t)S.off_cred + (2u * (h->n_triples + 1u) > 32u ? 2u * (h->n_triples + 1u) : 32u), 16);
    const bool external = cfg->defender_kind == MCBS_DEFENDER_EXTERNAL;
    // small topologies (Chain-10, ToyCtf): the eight sets of an env as 16-bit fields of one 16-byte word and 4-byte node rows,
    // so that the header and the WHOLE body are level-1 loads (mcbs_device.h); lists stay u8 / u16 with < 16 entries each
    S.tiny_p = h->n_props; S.tiny_v = h->max_slots;
    S.packed = (N <= 16u && h->n_cred_strings <= 16u && h->n_triples < 16u && S.tiny_p + 4u + 2u * S.tiny_v <= 32u &&
                !getenv("MCBS_NO_PACKED_SETS")) ? 1u : 0u;
    const size_t row_bytes = S.packed ? 4u : sizeof(Row);
    S.off_fw = external ? (uint32_t)align_up((size_t)S.off_rows + row_bytes * N, 16) : 0u;
    size_t body_end = external ? (size_t)S.off_fw + 2u * h->n_fw_lists : (size_t)S.off_rows + row_bytes * N;
    // ExternalRandomEvents overlay (mcbs_ere.hip): key lists, key counts, presence masks, service bits, rule lists {count, entries}
    StepCfg& C0 = b->C;
    std::vector<uint32_t> ere_lists;
    if (random_events) {
        const mcbs_ere_tables* et = reinterpret_cast<const mcbs_ere_tables*>(topo->host.data() + h->off_ere);
        C0.ere_key_cap = et->key_cap; C0.ere_n_library = et->n_library; C0.ere_lib_cols = et->lib_cols; C0.off_ere = h->off_ere;
        C0.ere_off_present = (uint32_t)align_up(body_end, 8);
        C0.ere_off_svc = C0.ere_off_present + 8u * N;
        C0.ere_off_keys = C0.ere_off_svc + 4u * N;
        C0.ere_off_kcnt = C0.ere_off_keys + et->key_cap * N;
        C0.ere_off_fw = (uint32_t)align_up((size_t)C0.ere_off_kcnt + N, 2);
        const uint16_t* fr = reinterpret_cast<const uint16_t*>(topo->host.data() + h->off_fw_range);
        uint32_t off = 0;
        for (uint32_t l = 0; l < h->n_fw_lists; ++l) {
            const uint32_t cap = fr[l * 2 + 1] + MCBS_FW_GROWTH;
            if (off > 0xFFFFu) { delete b; return fail(MCBS_ELIMIT, "firewall rule lists too large for the random-events overlay"); }
            ere_lists.push_back(off | (cap << 16));
            off += 2u * (1u + cap);
        }
        body_end = (size_t)C0.ere_off_fw + off;
        for (uint32_t n = 0; n < N; ++n) {
            const mcbs_node_static* nsn = reinterpret_cast<const mcbs_node_static*>(topo->host.data() + h->off_node) + n;
            if (nsn->svc_cnt > 32) { delete b; return fail(MCBS_ELIMIT, "node %u has more than 32 services", n); }
        }
    }
    S.body_stride = (uint32_t)align_up(body_end, S.packed ? 16 : 64);

    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_h0 = take(16ull * E), o_h1 = take(16ull * E), o_ep = take(4ull * E), o_pend = take(8ull * E);
    uint32_t wt = S.NW > S.SW ? S.NW : S.SW;
    S.wide = S.TW > 4u ? 1u : 0u;             // the cached-triple set does not fit 4 words: own column array (DevState::cach)
    if (!S.wide && S.TW > wt) wt = S.TW;
    S.WT = wt <= 1 ? 1 : (wt == 2 ? 2 : 4);
    // large topologies: G = WT lanes per env (mcbs_step_coop.hip).  More than 64 nodes guarantees the list regions its level-1 loads
    // cover (16 G discovery-order bytes, 32 G credential-cache bytes) lie inside the env's body; MCBS_NO_COOP=1: the one-lane kernel
    // It pays while the one-lane kernel would leave SIMDs idle: measured on MI355X (profiles/round3_notes.md) 4.43 vs 4.96 us for
    // 8 192 Chain-100 envs and 6.0 vs 7.4 us for 16 384 Random-256 envs, but 9.2 vs 9.0 us at 65 536 and 24.7 vs 22.9 us at 131 072 envs
    // (the chip is full either way and the G lanes' redundant scalar work then costs issue slots): up to 512 one-lane wavefronts.
    uint32_t coop_max_envs = 32768u;
    if (const char* ov = getenv("MCBS_COOP_MAX_ENVS")) coop_max_envs = (uint32_t)strtoul(ov, nullptr, 10);   // experiments
    b->coop = !S.packed && S.WT >= 2u && N > 64u && !S.wide && !b->lds_topo && !getenv("MCBS_NO_COOP") && E <= coop_max_envs &&
              (cfg->defender_kind == MCBS_DEFENDER_NONE || cfg->defender_kind == MCBS_DEFENDER_SCAN_AND_REIMAGE);
    const size_t o_masks = take(S.packed ? 16ull * E : 8ull * M_COUNT * S.WT * E);
    const size_t o_cach = S.wide ? take(8ull * S.TW * E) : 0;
    const bool has_def = cfg->defender_kind != MCBS_DEFENDER_NONE;   // in-env or external: both re-image nodes
    const size_t o_ring = has_def ? take(8ull * 16 * S.WT * E) : 0;
    const size_t o_init = take(S.body_stride);
    const size_t o_digest = take(sizeof(ObsDigest) * (size_t)E);
    const size_t o_rdigest = take(sizeof(ObsDigest));
    const size_t o_body = take((size_t)S.body_stride * E + 64);   // + 64: packed batches fetch a fixed 64 bytes of rows per env
    b->arena_bytes = off;

    DeviceGuard guard(cfg->device);
    hipError_t e = guard.err;
    if (e == hipSuccess) e = hipMalloc(&b->arena, b->arena_bytes);
    if (e != hipSuccess) { delete b; return fail(MCBS_EHIP, "state allocation of %zu bytes failed: %s", off, hipGetErrorString(e)); }
    uint8_t* a = b->arena;
    S.h0 = reinterpret_cast<uint4*>(a + o_h0); S.h1 = reinterpret_cast<double2*>(a + o_h1);
    S.episode = reinterpret_cast<uint32_t*>(a + o_ep); S.pending = reinterpret_cast<double*>(a + o_pend);
    S.masks = reinterpret_cast<uint64_t*>(a + o_masks);
    S.cach = S.wide ? reinterpret_cast<uint64_t*>(a + o_cach) : nullptr;
    S.ring = has_def ? reinterpret_cast<uint64_t*>(a + o_ring) : nullptr;
    S.body = a + o_body; S.init_body = a + o_init;
    b->digest = reinterpret_cast<ObsDigest*>(a + o_digest);
    b->reset_digest = reinterpret_cast<ObsDigest*>(a + o_rdigest);
    b->T.base = topo->dev;
    b->T.hot = topo->hot_dev;

    // reset image of one env body: rows of the initially owned nodes know all their properties and carry
    // max(initial privilege, LocalUser) (actions.py:149-152,263-270); every row carries its literal tags
    std::vector<uint8_t> init(S.body_stride, 0);
    const mcbs_node_static* ns = reinterpret_cast<const mcbs_node_static*>(topo->host.data() + h->off_node);
    for (uint32_t n = 0; n < N; ++n) {
        Row r{};
        const bool owned0 = (ns[n].flags & MCBS_NODE_INSTALLED0) != 0;
        r.props_tags = (owned0 ? ns[n].props : 0ull) | ((uint64_t)(ns[n].tags0 & 0xFu) << 60);
        S.row_put(init.data(), n, r);
    }
    memcpy(init.data() + S.off_disc, topo->host.data() + h->off_init_order, h->n_init_owned);
    if (external) memcpy(init.data() + S.off_fw, topo->host.data() + h->off_fw_list0, 2u * h->n_fw_lists);
    if (random_events) {
        const uint8_t* tb = topo->host.data();
        const mcbs_ere_tables* et = reinterpret_cast<const mcbs_ere_tables*>(tb + h->off_ere);
        const uint8_t* own_keys = reinterpret_cast<const uint8_t*>(et) + et->off_own_keys;
        const uint8_t* own_cnt = reinterpret_cast<const uint8_t*>(et) + et->off_own_cnt;
        const mcbs_service* sv0 = reinterpret_cast<const mcbs_service*>(tb + h->off_service);
        memcpy(init.data() + b->C.ere_off_keys, own_keys, (size_t)et->key_cap * N);
        memcpy(init.data() + b->C.ere_off_kcnt, own_cnt, N);
        for (uint32_t n = 0; n < N; ++n) {
            uint64_t present = 0;
            for (uint32_t k = 0; k < own_cnt[n]; ++k) present |= 1ull << own_keys[(size_t)n * et->key_cap + k];
            memcpy(init.data() + b->C.ere_off_present + 8u * n, &present, 8);
            uint32_t run = 0;
            for (uint32_t i = 0; i < ns[n].svc_cnt; ++i) if (sv0[ns[n].svc_off + i].running) run |= 1u << i;
            memcpy(init.data() + b->C.ere_off_svc + 4u * n, &run, 4);
        }
        const mcbs_fw_rule* fwr = reinterpret_cast<const mcbs_fw_rule*>(tb + h->off_fw_rule);
        const uint16_t* fr = reinterpret_cast<const uint16_t*>(tb + h->off_fw_range);
        for (uint32_t l = 0; l < h->n_fw_lists; ++l) {
            uint16_t* L = reinterpret_cast<uint16_t*>(init.data() + b->C.ere_off_fw + (ere_lists[l] & 0xFFFFu));
            L[0] = fr[l * 2 + 1];
            for (uint32_t i = 0; i < fr[l * 2 + 1]; ++i) L[1 + i] = (uint16_t)(fwr[fr[l * 2] + i].name | ((fwr[fr[l * 2] + i].allow ? 1u : 0u) << 8));
        }
        hipError_t e2 = hipMalloc(&b->ere_lists_dev, sizeof(uint32_t) * (ere_lists.size() + 1));
        if (e2 == hipSuccess) e2 = hipMemcpy(b->ere_lists_dev, ere_lists.data(), sizeof(uint32_t) * ere_lists.size(), hipMemcpyHostToDevice);
        if (e2 != hipSuccess) { (void)hipFree(b->arena); delete b; return fail(MCBS_EHIP, "random-events tables upload failed: %s", hipGetErrorString(e2)); }
        b->C.ere_lists = b->ere_lists_dev;
    }
    e = hipMemcpy(a + o_init, init.data(), init.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(b->arena); delete b; return fail(MCBS_EHIP, "init image upload failed: %s", hipGetErrorString(e)); }

    StepCfg& C = b->C;
    C.goal_reward = cfg->goal_reward; C.goal_low_availability = cfg->goal_low_availability;
    C.goal_own_atleast_percent = cfg->goal_own_atleast_percent; C.maintain_sla = cfg->maintain_sla;
    C.goal_own_pct_min = N + 1u;
    for (uint32_t k = 0; k <= N; ++k)
        if (!((double)k / (double)N < cfg->goal_own_atleast_percent)) { C.goal_own_pct_min = k; break; }     // (monotone in k: N > 0)
    C.winning_reward = cfg->winning_reward; C.losing_reward = cfg->losing_reward; C.scan_probability = cfg->scan_probability;
    C.total_sla_weight = h->total_sla_weight; C.full_availability = h->full_availability; C.full_sum = h->full_sum;
    C.seed = cfg->seed; C.env_id_base = cfg->env_id_base;
    C.has_attacker_goal = cfg->has_attacker_goal; C.goal_own_atleast = cfg->goal_own_atleast;
    C.defender_goal_eviction = cfg->defender_goal_eviction; C.defender_kind = cfg->defender_kind;
    C.scan_capacity = cfg->scan_capacity; C.scan_frequency = cfg->scan_frequency ? cfg->scan_frequency : 1;
    C.auto_reset = cfg->auto_reset; C.max_episode_steps = cfg->max_episode_steps; C.rng_kind = cfg->rng_kind;
    C.avail_any_order = h->avail_any_order;
    C.L = h->n_local; C.R = h->n_remote; C.P = h->n_ports; C.V = h->max_slots; C.n_props = h->n_props;
    C.K = cfg->maximum_discoverable_credentials_per_action;
    C.off_node = h->off_node; C.off_slot_of = h->off_slot_of; C.off_slot = h->off_slot; C.off_payload = h->off_payload;
    C.off_service = h->off_service; C.off_allowed = h->off_allowed; C.off_triple = h->off_triple;
    C.hot_node = topo->hot.node; C.hot_desc = topo->hot.desc; C.hot_payload = topo->hot.payload; C.hot_auth = topo->hot.auth;
    memcpy(C.rule_port, h->rule_port, 8); C.n_services = h->n_services; C.n_fw_lists = h->n_fw_lists; C.hot_fwlist = topo->hot.fwlist;
    C.auth_words = topo->hot.auth_words; C.hot_triple = topo->hot.triple; C.hot_avail = topo->hot.avail; C.hot_bytes = topo->hot.bytes;
    C.avail_uniform = h->avail_any_order ? 1u : 0u;
    C.avail_term0 = ns[0].avail_term;
    for (uint32_t n = 0; n < N; ++n) {
        if (ns[n].flags & MCBS_NODE_REIMAGABLE) C.reimagable[n >> 6] |= 1ull << (n & 63u);
        if (ns[n].avail_term != ns[0].avail_term) C.avail_uniform = 0u;
    }

    C.n_init = h->n_init_owned;
    if (S.packed && S.body_stride <= sizeof(C.init_image)) {
        C.init_image_ok = 1u;
        memcpy(C.init_image, init.data(), S.body_stride);
        uint16_t f[M_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint8_t* order0 = topo->host.data() + h->off_init_order;
        for (uint32_t i = 0; i < h->n_init_owned; ++i) {
            const uint32_t n = order0[i];
            f[M_DISC] |= 1u << n; f[M_INST] |= 1u << n; f[M_EVER] |= 1u << n;
            if (ns[n].priv0 & 1u) f[M_PLO] |= 1u << n;
            if (ns[n].priv0 & 2u) f[M_PHI] |= 1u << n;
        }
        f[M_RUN] = (uint16_t)((1u << N) - 1u);
        for (int q = 0; q < 4; ++q) C.init_packed[q] = (uint32_t)f[2 * q] | ((uint32_t)f[2 * q + 1] << 16);
    }

    e = hipMalloc(&b->C_dev, sizeof(StepCfg));
    if (e == hipSuccess) e = hipMemcpy(b->C_dev, &b->C, sizeof(StepCfg), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(b->arena); delete b; return fail(MCBS_EHIP, "config upload failed: %s", hipGetErrorString(e)); }
    hipLaunchKernelGGL(reset_kernel, dim3((E + 127) / 128), dim3(128), 0, 0, S, b->T, (const uint8_t*)nullptr, 0);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(b->arena); delete b; return fail(MCBS_EHIP, "initial reset failed: %s", hipGetErrorString(e)); }
    *out = b;
    return MCBS_OK;
}

extern "C" void mcbs_batch_destroy(mcbs_batch* b) {
    if (!b) return;
    DeviceGuard guard(b->cfg.device);
    for (hipEvent_t ev : b->ev) (void)hipEventDestroy(ev);
    if (b->arena) (void)hipFree(b->arena);
    if (b->C_dev) (void)hipFree(b->C_dev);
    if (b->ere_lists_dev) (void)hipFree(b->ere_lists_dev);
    delete b;
}

static int launch_ok(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MCBS_EHIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return MCBS_OK;
}

extern "C" int mcbs_reset(mcbs_batch* b, const uint8_t* env_mask, void* stream) {
    if (!b) return fail(MCBS_EINVAL, "null batch");
    MCBS_ON_DEVICE(b);
    hipLaunchKernelGGL(reset_kernel, dim3((b->S.E + 127) / 128), dim3(128), 0, (hipStream_t)stream, b->S, b->T, env_mask, 1);
    if (!env_mask) b->all_fresh = true;
    b->digest_state = !env_mask ? 0 : (b->digest_state == 1 ? 2 : b->digest_state);
    return launch_ok("reset");
}

extern "C" int mcbs_rewind(mcbs_batch* b, void* stream) {
    if (!b) return fail(MCBS_EINVAL, "null batch");
    MCBS_ON_DEVICE(b);
    hipLaunchKernelGGL(reset_kernel, dim3((b->S.E + 127) / 128), dim3(128), 0, (hipStream_t)stream, b->S, b->T, (const uint8_t*)nullptr, 0);
    b->all_fresh = true;
    b->digest_state = 0;
    return launch_ok("rewind");
}

extern "C" int mcbs_set_mask_discrete_stride(mcbs_batch* b, size_t stride_bytes) {
    if (!b) return fail(MCBS_EINVAL, "null batch");
    if (stride_bytes && stride_bytes < mcbs_discrete_action_count(b)) return fail(MCBS_EINVAL, "row stride shorter than the Discrete action count");
    b->disc_stride = stride_bytes;
    return MCBS_OK;
}

extern "C" int mcbs_set_draw_tape(mcbs_batch* b, const double* tape, uint32_t draws_per_step) {
    if (!b) return fail(MCBS_EINVAL, "null batch");
    b->tape = tape;
    b->tape_dps = tape ? draws_per_step : 0;
    return MCBS_OK;
}

static StepIO make_io(mcbs_batch* b, const int32_t* actions, float* reward, uint8_t* terminated, const mcbs_info_buffers* info) {
    StepIO io{};
    io.actions = actions; io.reward = reward; io.terminated = terminated;
    if (info) {
        io.availability = info->network_availability; io.step_count = info->step_count; io.truncated = info->truncated;
        io.oob = info->out_of_bound; io.raw_reward = info->raw_reward;
    }
    io.tape = b->tape; io.tape_dps = b->tape_dps;
#ifdef MCBS_DIAG
    io.stamps = b->stamps;
#endif
    // Kernel-argument budget: DevState + Topo + config pointer + StepIO are followed by the hidden launch arguments; the block
    // size the kernel reads first sits 12 bytes in.  When that crossed byte 256 the step took 5.73 instead of 5.47 us — a step
    // function (64 or 128 bytes more cost the same), as if only the first 256 bytes of the arguments are prefetched for the waves.
#ifndef MCBS_DIAG
    static_assert(sizeof(DevState) + sizeof(Topo) + sizeof(void*) + sizeof(StepIO) + 16 <= 256, "step kernel arguments spill into a fifth cache line");
#endif
    return io;
}

static int timing_begin(mcbs_batch* b, hipStream_t st, size_t* slot) {
    *slot = (size_t)-1;
    if (!b->timing) return MCBS_OK;
    if (b->ev_used + 2 > b->ev.size()) {
        for (int i = 0; i < 2; ++i) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); b->ev.push_back(ev); }
    }
    *slot = b->ev_used;
    b->ev_used += 2;
    HIP_TRY(hipEventRecord(b->ev[*slot], st));
    return MCBS_OK;
}

static int timing_end(mcbs_batch* b, hipStream_t st, size_t slot) {
    if (slot == (size_t)-1) return MCBS_OK;
    HIP_TRY(hipEventRecord(b->ev[slot + 1], st));
    return MCBS_OK;
}

// Kernel variant: words per set kept in registers (1, 2 or 4; 0 = packed batch), where the topology's hot image is read from, and
// which defender is configured (its code and loads are compiled out otherwise).
template <int PHASE, int WT, int DEF, bool MANY = false>
static void launch_step_v(mcbs_batch* b, const StepIO& io, hipStream_t st, const RollArgs& roll = RollArgs{}) {
    const uint32_t E = b->S.E, lds = b->C.hot_bytes;
    // Where the step kernel reads the topology's hot image from.  Measured on MI355X (profiles/round2_notes.md, tools/sweep_shapes.sh):
    // through L1 / L2 with 64-thread workgroups beats a per-workgroup LDS copy at every BASELINE shape — 5.4 vs 5.6 us at 65 536 Chain-10
    // envs, 5.5 vs 7.4 us for the Chain-100 shard (45 KB image, 128 wavefronts), 7.8 vs 8.5 us for Random-256 — once no table read is left
    // behind a store (leak payload prefetched, defender look-ups from scalars).  One-wavefront workgroups spread over all CUs; a copy per
    // workgroup only pays when many wavefronts share it, and then the image is L1-resident anyway.  MCBS_LDS_TOPO=1 selects the staged
    // variant (images up to 60 KB) for experiments.
    if (b->lds_topo && lds <= 60000u) {
        // workgroup size: as large as still leaves one workgroup per CU (256) — every workgroup stages its own copy of the hot
        // image, so at 65 536 envs 64-thread workgroups re-read it 4x as often as 256-thread ones, while 512 threads would leave half
        // of the CUs idle
        uint32_t block = lds <= 8192u ? 64u : 256u;
        while (block < 256u && E / (block * 2u) >= 256u) block *= 2u;
        if (b->step_block_override) block = b->step_block_override;
        const uint32_t shm = lds + (b->S.wide ? block * b->S.TW * 8u : 0u);
        if constexpr (MANY) hipLaunchKernelGGL((step_many_kernel<WT, true, DEF>), dim3((E + block - 1) / block), dim3(block), shm, st, b->S, b->T, b->C_dev, io, roll);
        else hipLaunchKernelGGL((step_kernel<PHASE, WT, true, DEF>), dim3((E + block - 1) / block), dim3(block), shm, st, b->S, b->T, b->C_dev, io);
    } else {
        const uint32_t block = 64u;                      // fixed (the kernel relies on it); beyond one wavefront per SIMD the shape no longer
                                                         // matters: 22.7 (64) vs 22.2 us (256) at 131 072 Random-256 envs
        const uint32_t shm = b->S.wide ? block * b->S.TW * 8u : 0u;
        if constexpr (MANY) hipLaunchKernelGGL((step_many_kernel<WT, false, DEF>), dim3((E + block - 1) / block), dim3(block), shm, st, b->S, b->T, b->C_dev, io, roll);
        else hipLaunchKernelGGL((step_kernel<PHASE, WT, false, DEF>), dim3((E + block - 1) / block), dim3(block), shm, st, b->S, b->T, b->C_dev, io);
    }
}

template <int PHASE, int WT, bool MANY = false>
static void launch_step_nw(mcbs_batch* b, const StepIO& io, hipStream_t st, const RollArgs& roll) {
    if (b->cfg.defender_kind == MCBS_DEFENDER_SCAN_AND_REIMAGE) launch_step_v<PHASE, WT, MCBS_DEFENDER_SCAN_AND_REIMAGE, MANY>(b, io, st, roll);
    else if (b->cfg.defender_kind == MCBS_DEFENDER_RANDOM_EVENTS) launch_step_v<PHASE, WT, MCBS_DEFENDER_RANDOM_EVENTS, MANY>(b, io, st, roll);
    else if (b->cfg.defender_kind == MCBS_DEFENDER_EXTERNAL) launch_step_v<PHASE, WT, MCBS_DEFENDER_EXTERNAL, MANY>(b, io, st, roll);
    else launch_step_v<PHASE, WT, MCBS_DEFENDER_NONE, MANY>(b, io, st, roll);
}

template <int G>
static void launch_step_coop(mcbs_batch* b, const StepIO& io, hipStream_t st) {
    const uint32_t epw = 64u / G;
    const dim3 grid((b->S.E + epw - 1u) / epw), block(64);
    if (b->cfg.defender_kind == MCBS_DEFENDER_SCAN_AND_REIMAGE)
        hipLaunchKernelGGL((step_coop_kernel<G, MCBS_DEFENDER_SCAN_AND_REIMAGE>), grid, block, 0, st, b->S, b->T, b->C_dev, io);
    else hipLaunchKernelGGL((step_coop_kernel<G, MCBS_DEFENDER_NONE>), grid, block, 0, st, b->S, b->T, b->C_dev, io);
}

template <int PHASE, bool MANY = false>
static int launch_step(mcbs_batch* b, const StepIO& io, hipStream_t st, const char* what, const RollArgs& roll = RollArgs{}) {
    b->all_fresh = false;
    if (PHASE == 0 && !MANY && b->coop) {
        if (b->S.WT == 2) launch_step_coop<2>(b, io, st); else launch_step_coop<4>(b, io, st);
        return launch_ok(what);
    }
    if (b->S.packed) launch_step_nw<PHASE, 0, MANY>(b, io, st, roll);       // WT 0: packed sets (one word of registers each)
    else if (b->S.WT == 1) launch_step_nw<PHASE, 1, MANY>(b, io, st, roll);
    else if (b->S.WT == 2) launch_step_nw<PHASE, 2, MANY>(b, io, st, roll);
    else launch_step_nw<PHASE, 4, MANY>(b, io, st, roll);
    return launch_ok(what);
}

extern "C" int mcbs_step(mcbs_batch* b, const int32_t* actions, float* reward, uint8_t* terminated,
                         const mcbs_info_buffers* info, void* stream) {
    if (!b || !actions || !reward || !terminated) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    if (b->cfg.rng_kind == MCBS_RNG_TAPE && b->cfg.defender_kind != MCBS_DEFENDER_NONE && !b->tape)
        return fail(MCBS_ESTATE, "rng_kind is TAPE but no draw tape was set (mcbs_set_draw_tape)");
    hipStream_t st = (hipStream_t)stream;
    const StepIO io = make_io(b, actions, reward, terminated, info);
    size_t slot;
    int rc = timing_begin(b, st, &slot);
    if (rc) return rc;
    if ((rc = launch_step<0>(b, io, st, "step"))) return rc;
    return timing_end(b, st, slot);
}

// n_steps consecutive steps in one launch: scripted / recorded / pre-sampled action sequences (replaying a trace, random-agent
// baselines, evaluating fixed plans), where nothing has to be observed between steps.  Results are those of n_steps calls of
// mcbs_step; what is saved is the per-launch cost between dependent launches (1.5 us of a 5.5 us step at 65 536 envs).
extern "C" int mcbs_step_many(mcbs_batch* b, const int32_t* actions, float* reward, uint8_t* terminated, uint32_t n_steps, void* stream) {
    if (!b || !actions || !reward || !terminated) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    if (n_steps == 0) return MCBS_OK;
    if (b->cfg.rng_kind == MCBS_RNG_TAPE && b->cfg.defender_kind != MCBS_DEFENDER_NONE)
        return fail(MCBS_ESTATE, "mcbs_step_many needs the Philox generator: a draw tape holds one step's draws");
    hipStream_t st = (hipStream_t)stream;
    StepIO io = make_io(b, actions, reward, terminated, nullptr);
    io.n_steps = n_steps;
    size_t slot;
    int rc = timing_begin(b, st, &slot);
    if (rc) return rc;
    if ((rc = launch_step<0, true>(b, io, st, "step (many)"))) return rc;
    return timing_end(b, st, slot);
}

// Random agents entirely on the device: every step's action is drawn from the env's own state inside the step kernel
// (sample_action, the distribution of CyberBattleEnv.sample_valid_action or uniform over the action space), K steps per launch.
// This is marlon.simulate's loop for random agents (marlon/simulate.py:14-35) without a launch per sample and per step.
extern "C" int mcbs_rollout_random(mcbs_batch* b, int32_t valid, uint64_t seed, uint64_t first_step, uint32_t n_steps,
                                   int32_t* actions_out, float* reward, uint8_t* terminated, void* stream) {
    if (!b || !reward || !terminated) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    if (n_steps == 0) return MCBS_OK;
    if (b->cfg.rng_kind == MCBS_RNG_TAPE && b->cfg.defender_kind != MCBS_DEFENDER_NONE)
        return fail(MCBS_ESTATE, "mcbs_rollout_random needs the Philox generator: a draw tape holds one step's draws");
    hipStream_t st = (hipStream_t)stream;
    RollArgs roll;
    roll.mode = valid ? 2u : 1u; roll.nmax = b->cfg.maximum_node_count; roll.cmax = b->cfg.maximum_total_credentials;
    roll.seed = seed; roll.step0 = first_step;
    StepIO io = make_io(b, actions_out, reward, terminated, nullptr);
    io.n_steps = n_steps;
    return launch_step<0, true>(b, io, st, "rollout", roll);   // the random agent's parameters are kernel arguments: no host copy, no shared state
}

static int launch_masks(mcbs_batch* b, const mcbs_obs_buffers* o, hipStream_t st, const uint8_t* env_mask, bool masks_only);

// The first whole-batch observation after a whole-batch reset also keeps env 0's digest as "the digest of a freshly reset env"
// (mcbs_attacker_wrapper_finish hands it to the envs it resets): one 64-byte device copy, once per batch.
static int capture_reset_digest(mcbs_batch* b, hipStream_t st, bool whole_batch) {
    if (!whole_batch || !b->all_fresh || b->reset_digest_ok) return MCBS_OK;
    HIP_TRY(hipMemcpyAsync(b->reset_digest, b->digest, sizeof(ObsDigest), hipMemcpyDeviceToDevice, st));
    b->reset_digest_ok = true;
    return MCBS_OK;
}

static int launch_obs_inner(mcbs_batch* b, const mcbs_obs_buffers* o, hipStream_t st, bool masks_only, const uint8_t* env_mask);
static int launch_obs(mcbs_batch* b, const mcbs_obs_buffers* o, hipStream_t st, bool masks_only = false, const uint8_t* env_mask = nullptr) {
    const int rc = launch_obs_inner(b, o, st, masks_only, env_mask);
    if (rc) return rc;
    if (!env_mask) b->digest_state = 1;                       // every observation kernel (masks-only ones too) leaves the env's digest
    else if (b->digest_state == 2) b->digest_state = 1;       // the envs reset by mask have been re-observed (the caller's protocol)
    return capture_reset_digest(b, st, !masks_only && !env_mask);
}

static int launch_obs_inner(mcbs_batch* b, const mcbs_obs_buffers* o, hipStream_t st, bool masks_only, const uint8_t* env_mask) {
    ObsIO O{};
    O.masks_only = masks_only ? 1u : 0u;
    O.env_mask = env_mask;
    O.scalars = o->scalars; O.leaked = o->leaked_credentials; O.cache_matrix = o->credential_cache_matrix;
    O.props = o->discovered_nodes_properties; O.priv = o->nodes_privilegelevel; O.mask_local = o->mask_local;
    O.mask_remote = o->mask_remote; O.mask_connect = o->mask_connect; O.mask_discrete = o->mask_discrete;
    O.Nmax = b->cfg.maximum_node_count; O.Cmax = b->cfg.maximum_total_credentials; O.K = b->cfg.maximum_discoverable_credentials_per_action;
    // small action spaces: the per-env wavefront also writes mask_remote / mask_connect (mcbs_obs.hip)
    const size_t rows = (size_t)O.Nmax * O.Nmax, RL = (size_t)b->C.P * O.Cmax;
    const bool small = rows <= 256 && !b->no_fused_masks;
    O.fuse_remote = small && o->mask_remote && (rows * b->C.R) % 4 == 0 && reinterpret_cast<uintptr_t>(o->mask_remote) % 4 == 0;
    const size_t ML = (size_t)O.Nmax * b->C.L, MR = rows * b->C.R, M = rows * RL;
    const bool dwords_ok = small && RL >= 4 && RL + 4 <= 1040 && M % 4 == 0;     // the row pattern fits its kilobyte of LDS
    O.fuse_connect = 0;
    if (small && o->mask_connect) {
        size_t g16 = 16;
        while (RL % g16) g16 >>= 1;                                                                                          // gcd(RL, 16)
        if (RL % 16 == 0 && RL / 16 <= 64 && reinterpret_cast<uintptr_t>(o->mask_connect) % 16 == 0) O.fuse_connect = 1;   // 16-byte chunks
        else if (RL >= 16 && RL / g16 <= 64 && M % 16 == 0 && reinterpret_cast<uintptr_t>(o->mask_connect) % 16 == 0 && !b->slow_masks) {
            O.fuse_connect = 3; O.conn_pc = (uint32_t)(RL / g16);                                                            // 16-byte chunks, any RL >= 16
        } else if (dwords_ok && reinterpret_cast<uintptr_t>(o->mask_connect) % 4 == 0) O.fuse_connect = 2;                  // dwords, any RL
    }
    // per-source blocks (mcbs_obs.hip stream_blocks): where round 2 switched row bytes on and off per chunk (connect rows that are not a
    // multiple of 16 bytes; the remote mask as dwords), a chunk is now one LDS read of a block variant.  Needs dword-sized blocks of at
    // least one chunk and whole-chunk masks; MCBS_NO_BLOCK_MASKS=1: round 2's writers
    uint32_t blk_bytes = 0;
    const size_t BLc = (size_t)O.Nmax * RL, BLr = (size_t)O.Nmax * b->C.R;
    // largest block whose four variants still fit a workgroup's LDS next to the staging areas (four wavefronts, 60 KB in all, 4 KB of it
    // obs_quad_kernel's static arrays): 3 161 bytes for a 16-node topology, less when the credential lists are long
    size_t blk_cap = 0;
    {
        const size_t fixed = obs_stage_bytes(b->S.N, b->topo->H()->n_triples, 0), budget = (61440u - 4096u) / 4u;
        if (budget > fixed + 16u * 12u) blk_cap = ((budget - fixed) / 16u) * 4u - 31u;
        if (blk_cap > 4064u) blk_cap = 4064u;
    }
    if (!b->no_block_masks) {
        if (O.fuse_connect == 3 && BLc % 4 == 0 && BLc >= 16 && BLc <= blk_cap) { O.fuse_connect = 4; blk_bytes = (uint32_t)BLc; }
        if (O.fuse_remote && BLr % 4 == 0 && BLr >= 16 && BLr <= blk_cap && MR % 16 == 0 && reinterpret_cast<uintptr_t>(o->mask_remote) % 16 == 0) {
            O.fuse_remote = 2;
            if (BLr > blk_bytes) blk_bytes = (uint32_t)BLr;
        }
    }
    O.fuse_discrete = dwords_ok && o->mask_discrete && ML % 4 == 0 && MR % 4 == 0 && b->C.L > 0 && b->C.R > 0 &&
                      reinterpret_cast<uintptr_t>(o->mask_discrete) % 4 == 0;
    {   // row stride of mask_discrete: dense unless the caller padded its rows (mcbs_set_mask_discrete_stride)
        const size_t D = M + ML + MR, stride = b->disc_stride ? b->disc_stride : D;
        if (stride > 0xFFFFFFFFull) return fail(MCBS_ELIMIT, "mask_discrete row stride too large");
        O.disc_stride = (uint32_t)stride;
        if (O.fuse_discrete && stride % 4 != 0) O.fuse_discrete = 0;
        O.nt_discrete = (O.fuse_discrete && stride % 128 == 0 && reinterpret_cast<uintptr_t>(o->mask_discrete) % 128 == 0 && M % 128 == 0) ? 1u : 0u;
    }
    O.disc_blocks = (!b->no_block_masks && O.fuse_discrete && RL % 16 != 0 && BLc % 4 == 0 && BLc >= 16 && BLc <= blk_cap) ? 1u : 0u;
    if (O.disc_blocks && BLc > blk_bytes) blk_bytes = (uint32_t)BLc;
    O.disc_remote_blocks = (!b->no_block_masks && O.fuse_discrete && BLr % 4 == 0 && BLr >= 16 && BLr <= blk_cap) ? 1u : 0u;
    if (O.disc_remote_blocks && BLr > blk_bytes) blk_bytes = (uint32_t)BLr;
    O.blk_region = blk_bytes ? ((blk_bytes + 16u + 15u) / 16u) * 4u : 0u;
    O.dBLc = fast_div_host(BLc ? (uint32_t)BLc : 1u); O.dBLr = fast_div_host(BLr ? (uint32_t)BLr : 1u);
    O.nt_connect = (O.fuse_connect == 1 && M % 128 == 0 && reinterpret_cast<uintptr_t>(o->mask_connect) % 128 == 0) ? 1u : 0u;
    auto fdh = [](size_t d) { return fast_div_host(d ? (uint32_t)d : 1u); };
    O.dNP = fdh(b->C.n_props); O.dL = fdh(b->C.L); O.dR = fdh(b->C.R); O.dNm = fdh(O.Nmax); O.dRL = fdh(RL); O.dC = fdh(O.Cmax);
    O.dPC = fdh(O.conn_pc); O.dCPR = fdh(RL >> 4);
    const uint32_t obs_shm = 4u * obs_stage_bytes(b->S.N, b->topo->H()->n_triples, O.blk_region);
    const bool no_masks = !o->mask_local && !o->mask_remote && !o->mask_connect && !o->mask_discrete;
    const bool groups_ok = !env_mask && !masks_only && b->S.NW == 1u && b->S.N <= 16u && O.Nmax <= 16u && O.Cmax <= 16u &&
                           b->topo->H()->n_triples <= 15u && b->cfg.defender_kind != MCBS_DEFENDER_RANDOM_EVENTS && !b->no_fused_masks;
    if (groups_ok && no_masks) {
        // ~1 KB per env, no mask field: sixteen lanes per env instead of a wavefront (mcbs_obs.hip obs_tiny_kernel)
        hipLaunchKernelGGL(obs_tiny_kernel, dim3((b->S.E + 15u) / 16u), dim3(256), 0, st, b->S, b->T, b->C_dev, O, b->digest);
        return launch_ok("obs_tiny");
    }
    // ... with mask fields: the small fields by 16-lane groups, the masks of the wavefront's four envs one after the other (obs_quad_kernel).
    // Measured (tools/obs_field_matrix.py, profiles/round3_notes.md section 7): ToyCtf's whole observation 43 -> 40 us at 16 384 envs and
    // 190 -> 145 us at 65 536 (its 10 080-byte mask rows share cache lines with their neighbours, and four envs' small fields go out in one
    // store instead of four partial lines); Chain-10, whose mask rows are whole 128-byte lines written with non-temporal stores, streams
    // ~10 % FASTER with a wavefront per env (152-168 against 171-185 us at 65 536 envs) and keeps it.  MCBS_QUAD_OBS=1 / MCBS_NO_QUAD_OBS=1
    // force one or the other.
    // The flat Discrete mask of a batch too small to give every SIMD several four-env wavefronts also stays (ToyCtf, 16 384 envs: 50 us
    // against 52-54; 65 536 envs: 207 -> 171 us).
    const bool line_rows = (o->mask_connect && O.nt_connect) || (o->mask_discrete && O.nt_discrete);
    const bool few_waves = o->mask_discrete && b->S.E < 32768u;
    if (groups_ok && !b->no_quad_obs && ((!line_rows && !few_waves) || b->force_quad_obs)) {
        const uint32_t quad_shm = 4u * (272u + 1040u + 16u * O.blk_region);
        hipLaunchKernelGGL(obs_quad_kernel, dim3((b->S.E + 15u) / 16u), dim3(256), quad_shm, st, b->S, b->T, b->C_dev, O, b->digest);
    } else if (env_mask) {     // sparse by nature (the envs a VecEnv just reset): 64 envs' mask bytes per wavefront
        const uint32_t waves = (b->S.E + 63u) / 64u;
        hipLaunchKernelGGL(obs_scan_kernel, dim3((waves + 3u) / 4u), dim3(256), obs_shm, st, b->S, b->T, b->C_dev, O, b->digest);
    } else {
        hipLaunchKernelGGL(obs_small_kernel, dim3((b->S.E + 3) / 4), dim3(256), obs_shm, st, b->S, b->T, b->C_dev, O, b->digest);
    }
    int rc = launch_ok("obs_small");
    if (rc) return rc;
    mcbs_obs_buffers rest = *o;                // what the fused wavefront has not written
    if (O.fuse_remote) rest.mask_remote = nullptr;
    if (O.fuse_connect) rest.mask_connect = nullptr;
    if (O.fuse_discrete) rest.mask_discrete = nullptr;
    return launch_masks(b, &rest, st, env_mask, masks_only);
}

static inline bool getenv_rows_off(const mcbs_batch* b) { return b->no_row_masks; }

template <int REGION>
static int launch_region(mcbs_batch* b, int8_t* dst, size_t env_stride, size_t region_off, size_t len, hipStream_t st, const uint8_t* env_mask, bool masks_only) {
    const uint32_t Nm = b->cfg.maximum_node_count, Cm = b->cfg.maximum_total_credentials;
    const uintptr_t base = reinterpret_cast<uintptr_t>(dst) + region_off;
    int W = 1;
    if (base % 16 == 0 && env_stride % 16 == 0 && len % 16 == 0) W = 16;
    else if (base % 4 == 0 && env_stride % 4 == 0 && len % 4 == 0) W = 4;
    const size_t chunks = (len + W - 1) / W;
    const dim3 grid(b->S.E, (unsigned)((chunks + 255) / 256));
    if (grid.y > 65535u) return fail(MCBS_ELIMIT, "mask region too large for one launch");
    const uint32_t RL = REGION == 0 ? b->C.P * Cm : (REGION == 1 ? Nm * b->C.R : 0u), Cc = REGION == 0 ? Cm : RL;
    if (W == 16 && REGION == 0 && RL % 16u == 0 && RL >= 16u && !b->slow_masks && !getenv_rows_off(b)) {
        // whole rows of 16-byte chunks: pattern row + per-row on/off bytes in LDS (mcbs_obs.hip: mask_connect_rows_kernel)
        const uint32_t cpr = RL / 16u, rows = Nm * Nm;
        uint32_t rpb = 128u;                                          // rows per workgroup: >= 16 KB of output each
        while (rpb * RL < 16384u && rpb < rows) rpb *= 2u;
        const uint32_t lds = cpr * 16u + ((rpb + 15u) & ~15u);
        if (lds <= 60000u) {
            const uint32_t gx = (rows + rpb - 1u) / rpb;
            uint32_t gy = b->S.E;
            if ((uint64_t)gx * gy > (1u << 20)) gy = (1u << 20) / gx ? (1u << 20) / gx : 1u;
            hipLaunchKernelGGL(mask_connect_rows_kernel, dim3(gx, gy), dim3(256), lds, st, b->S, b->digest, dst, env_stride, region_off,
                               Nm, Cm, RL, rpb, env_mask, masks_only ? 0u : 1u);
            return launch_ok("mask (rows)");
        }
    }
    if (W == 16 && REGION != 2 && RL >= 16u && len < (1ull << 31) && !b->slow_masks) {
        auto fd = [](uint32_t d) { return fast_div_host(d); };
        const uint32_t flat = chunks < 256 && (uint64_t)chunks * b->S.E < (1ull << 31) ? 1u : 0u;
        uint32_t gx = (uint32_t)((chunks + 1023) / 1024 < 8 ? (chunks + 1023) / 1024 : 8);
        uint32_t gy = 4096u / gx;
        if (gy > b->S.E) gy = b->S.E;
        if (flat) { gx = 1; gy = (uint32_t)(((uint64_t)chunks * b->S.E + 255) / 256 < 4096 ? ((uint64_t)chunks * b->S.E + 255) / 256 : 4096); }
        const dim3 fgrid(gx, gy);
#define MCBS_LAUNCH_FAST(REG_, FLAT_) hipLaunchKernelGGL((mask_fast_kernel<REG_, FLAT_>), fgrid, dim3(256), 0, st, b->S, b->digest, dst, env_stride, \
        region_off, (uint32_t)len, RL, Cc, Nm, b->C.R, fd(RL), fd(Cc), fd(Nm), env_mask, masks_only ? 0u : 1u, fd((uint32_t)chunks))
        if (REGION == 0) { if (flat) MCBS_LAUNCH_FAST(0, true); else MCBS_LAUNCH_FAST(0, false); }
        else { if (flat) MCBS_LAUNCH_FAST(1, true); else MCBS_LAUNCH_FAST(1, false); }
#undef MCBS_LAUNCH_FAST
        return launch_ok("mask (fast)");
    }
    if (W == 16) hipLaunchKernelGGL((mask_kernel<16, REGION>), grid, dim3(256), 0, st, b->S, b->T, b->C_dev, b->digest, dst, env_stride, region_off, Nm, Cm, env_mask, masks_only ? 0u : 1u);
    else if (W == 4) hipLaunchKernelGGL((mask_kernel<4, REGION>), grid, dim3(256), 0, st, b->S, b->T, b->C_dev, b->digest, dst, env_stride, region_off, Nm, Cm, env_mask, masks_only ? 0u : 1u);
    else hipLaunchKernelGGL((mask_kernel<1, REGION>), grid, dim3(256), 0, st, b->S, b->T, b->C_dev, b->digest, dst, env_stride, region_off, Nm, Cm, env_mask, masks_only ? 0u : 1u);
    return launch_ok("mask");
}

static int launch_masks(mcbs_batch* b, const mcbs_obs_buffers* o, hipStream_t st, const uint8_t* env_mask, bool masks_only) {
    const size_t Nm = b->cfg.maximum_node_count, Cm = b->cfg.maximum_total_credentials;
    const size_t M = Nm * Nm * b->C.P * Cm, ML = Nm * b->C.L, MR = Nm * Nm * b->C.R;
    int rc = MCBS_OK;
    if (o->mask_connect && (rc = launch_region<0>(b, o->mask_connect, M, 0, M, st, env_mask, masks_only))) return rc;
    if (o->mask_remote && (rc = launch_region<1>(b, o->mask_remote, MR, 0, MR, st, env_mask, masks_only))) return rc;
    if (o->mask_discrete) {   // connect | local | remote (action_masking.py:96-110)
        const size_t D = b->disc_stride ? b->disc_stride : M + ML + MR;       // (env stride; the regions' offsets inside a row are unchanged)
        if ((rc = launch_region<0>(b, o->mask_discrete, D, 0, M, st, env_mask, masks_only))) return rc;
        if ((rc = launch_region<2>(b, o->mask_discrete, D, M, ML, st, env_mask, masks_only))) return rc;
        if ((rc = launch_region<1>(b, o->mask_discrete, D, M + ML, MR, st, env_mask, masks_only))) return rc;
    }
    return rc;
}

extern "C" int mcbs_observe(mcbs_batch* b, const mcbs_obs_buffers* obs, void* stream) {
    if (!b || !obs) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    return launch_obs(b, obs, (hipStream_t)stream);
}

extern "C" int mcbs_observe_masked(mcbs_batch* b, const mcbs_obs_buffers* obs, const uint8_t* env_mask, void* stream) {
    if (!b || !obs) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    return launch_obs(b, obs, (hipStream_t)stream, false, env_mask);
}

extern "C" int mcbs_action_mask(mcbs_batch* b, const mcbs_obs_buffers* masks, void* stream) {
    if (!b || !masks) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    return launch_obs(b, masks, (hipStream_t)stream, true);
}

extern "C" int mcbs_step_observe(mcbs_batch* b, const int32_t* actions, float* reward, uint8_t* terminated,
                                 const mcbs_info_buffers* info, const mcbs_obs_buffers* obs, void* stream) {
    if (!b || !actions || !reward || !terminated || !obs) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    if (b->cfg.rng_kind == MCBS_RNG_TAPE && b->cfg.defender_kind != MCBS_DEFENDER_NONE && !b->tape)
        return fail(MCBS_ESTATE, "rng_kind is TAPE but no draw tape was set (mcbs_set_draw_tape)");
    hipStream_t st = (hipStream_t)stream;
    const StepIO io = make_io(b, actions, reward, terminated, info);
    int rc = launch_step<1>(b, io, st, "step (attacker phase)");
    if (rc) return rc;
    if ((rc = launch_obs(b, obs, st))) return rc;
    return launch_step<2>(b, io, st, "step (defender phase)");
}

extern "C" int mcbs_attacker_wrapper_post(mcbs_batch* b, const mcbs_wrapper_buffers* w, float modifier, int32_t max_timesteps, void* stream) {
    if (!b || !w) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    const void* const* p = reinterpret_cast<const void* const*>(w);
    for (size_t i = 0; i + 1 < sizeof(*w) / sizeof(void*); ++i)          // `executed` (the last member) is not written by this call
        if (!p[i]) return fail(MCBS_EINVAL, "mcbs_wrapper_buffers: every array but `executed` is required");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(w->n_done, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(wrapper_post_kernel, dim3((b->S.E + 255) / 256), dim3(256), 0, st, b->S.E, *w, modifier, max_timesteps);
    return launch_ok("wrapper post");
}

extern "C" int mcbs_attacker_wrapper_clear(mcbs_batch* b, const mcbs_wrapper_buffers* w, void* stream) {
    if (!b || !w || !w->dones || !w->timesteps || !w->valid_action_count || !w->invalid_action_count || !w->episode_returns || !w->has_cyber_reward)
        return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    hipLaunchKernelGGL(wrapper_clear_kernel, dim3((b->S.E + 255) / 256), dim3(256), 0, (hipStream_t)stream, b->S.E, *w);
    return launch_ok("wrapper clear");
}

extern "C" int mcbs_copy_rows_masked(mcbs_batch* b, const mcbs_row_copies* copies, const uint8_t* env_mask, void* stream) {
    if (!b || !copies || !env_mask) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    if (copies->n == 0) return MCBS_OK;
    if (copies->n > 8) return fail(MCBS_EINVAL, "at most eight arrays per call");
    for (uint32_t i = 0; i < copies->n; ++i) if (!copies->src[i] || !copies->dst[i]) return fail(MCBS_EINVAL, "null array");
    const uint32_t waves = (b->S.E + 63u) / 64u;
    hipLaunchKernelGGL(copy_rows_masked_kernel, dim3((waves + 3u) / 4u), dim3(256), 0, (hipStream_t)stream, *copies, env_mask, b->S.E);
    return launch_ok("copy rows");
}

static int finish_args(mcbs_batch* b, const mcbs_wrapper_buffers* w, float modifier, int32_t max_timesteps, int32_t auto_reset,
                       const mcbs_row_copies* keep, const mcbs_row_copies* fresh, WrapperFinishArgs* A) {
    if (!b || !w) return fail(MCBS_EINVAL, "null argument");
    const void* const* p = reinterpret_cast<const void* const*>(w);
    for (size_t i = 0; i + 2 < sizeof(*w) / sizeof(void*); ++i)          // n_done and executed (the last two members) may be NULL
        if (!p[i]) return fail(MCBS_EINVAL, "mcbs_wrapper_buffers: every array but n_done and executed is required");
    mcbs_row_copies none{};
    const mcbs_row_copies* rc[2] = {keep ? keep : &none, fresh ? fresh : &none};
    for (const mcbs_row_copies* c : rc) {
        if (c->n > 8) return fail(MCBS_EINVAL, "at most eight arrays per list");
        for (uint32_t i = 0; i < c->n; ++i) if (!c->src[i] || !c->dst[i]) return fail(MCBS_EINVAL, "null array");
    }
    if (auto_reset && !b->reset_digest_ok)
        return fail(MCBS_ESTATE, "no reset observation yet: reset the whole batch (mcbs_reset, NULL mask) and observe it once before the first call");
    A->w = *w; A->modifier = modifier; A->max_timesteps = max_timesteps; A->auto_reset = auto_reset; A->pad = 0;
    A->keep = *rc[0]; A->fresh = *rc[1]; A->digest = b->digest; A->reset_digest = b->reset_digest;
    return MCBS_OK;
}

extern "C" int mcbs_attacker_wrapper_finish(mcbs_batch* b, const mcbs_wrapper_buffers* w, float modifier, int32_t max_timesteps, int32_t auto_reset,
                                            const mcbs_row_copies* keep, const mcbs_row_copies* fresh, void* stream) {
    WrapperFinishArgs A;
    const int rc = finish_args(b, w, modifier, max_timesteps, auto_reset, keep, fresh, &A);
    if (rc) return rc;
    MCBS_ON_DEVICE(b);
    b->all_fresh = false;
    hipLaunchKernelGGL(wrapper_finish_kernel, dim3((b->S.E + 255) / 256), dim3(256), 0, (hipStream_t)stream, b->S, b->T, A);
    return launch_ok("wrapper finish");
}

template <int WT, int DEF>
static void launch_step2_finish_v(mcbs_batch* b, const StepIO& io, const WrapperFinishArgs& A, hipStream_t st) {
    const uint32_t shm = b->S.wide ? 64u * b->S.TW * 8u : 0u;
    hipLaunchKernelGGL((step2_finish_kernel<WT, DEF>), dim3((b->S.E + 63u) / 64u), dim3(64), shm, st, b->S, b->T, b->C_dev, io, A);
}
template <int WT>
static void launch_step2_finish_nw(mcbs_batch* b, const StepIO& io, const WrapperFinishArgs& A, hipStream_t st) {
    if (b->cfg.defender_kind == MCBS_DEFENDER_SCAN_AND_REIMAGE) launch_step2_finish_v<WT, MCBS_DEFENDER_SCAN_AND_REIMAGE>(b, io, A, st);
    else if (b->cfg.defender_kind == MCBS_DEFENDER_RANDOM_EVENTS) launch_step2_finish_v<WT, MCBS_DEFENDER_RANDOM_EVENTS>(b, io, A, st);
    else if (b->cfg.defender_kind == MCBS_DEFENDER_EXTERNAL) launch_step2_finish_v<WT, MCBS_DEFENDER_EXTERNAL>(b, io, A, st);
    else launch_step2_finish_v<WT, MCBS_DEFENDER_NONE>(b, io, A, st);
}

template <int WT, int DEF>
static void launch_decode_step1_v(mcbs_batch* b, const StepIO& io, const int64_t* md, const int64_t* discrete, uint8_t* invalid, hipStream_t st) {
    const uint32_t shm = b->S.wide ? 64u * b->S.TW * 8u : 0u;
    hipLaunchKernelGGL((decode_step1_kernel<WT, DEF>), dim3((b->S.E + 63u) / 64u), dim3(64), shm, st, b->S, b->T, b->C_dev, io,
                       b->cfg.maximum_node_count, b->cfg.maximum_total_credentials, md, discrete, invalid);
}
template <int WT>
static void launch_decode_step1_nw(mcbs_batch* b, const StepIO& io, const int64_t* md, const int64_t* discrete, uint8_t* invalid, hipStream_t st) {
    if (b->cfg.defender_kind == MCBS_DEFENDER_SCAN_AND_REIMAGE) launch_decode_step1_v<WT, MCBS_DEFENDER_SCAN_AND_REIMAGE>(b, io, md, discrete, invalid, st);
    else if (b->cfg.defender_kind == MCBS_DEFENDER_RANDOM_EVENTS) launch_decode_step1_v<WT, MCBS_DEFENDER_RANDOM_EVENTS>(b, io, md, discrete, invalid, st);
    else if (b->cfg.defender_kind == MCBS_DEFENDER_EXTERNAL) launch_decode_step1_v<WT, MCBS_DEFENDER_EXTERNAL>(b, io, md, discrete, invalid, st);
    else launch_decode_step1_v<WT, MCBS_DEFENDER_NONE>(b, io, md, discrete, invalid, st);
}

static bool fused_wrapper_batch_ok(const mcbs_batch* b) {
    return !(b->no_fused_wrapper || b->lds_topo || b->no_fused_masks || !b->S.packed || !b->C.init_image_ok || b->S.N > 16u ||
             b->cfg.maximum_node_count > 16u || b->cfg.maximum_total_credentials > 16u || b->topo->H()->n_triples > 15u ||
             b->cfg.defender_kind == MCBS_DEFENDER_RANDOM_EVENTS);
}

extern "C" int32_t mcbs_attacker_wrapper_step_launches(const mcbs_batch* b, int32_t with_masks) {
    if (!b) return 0;
    return (!with_masks && fused_wrapper_batch_ok(b)) ? 1 : 3;
}

// The whole wrapper step in ONE launch (mcbs_wrapper_fused.hip) when the batch and the request fit it: packed batch whose reset image is
// held in the config, at most 16 nodes / cached credentials, no mask field, observation rows of whole 16-byte vectors, every requested
// field paired with its terminal array and its reset row (auto_reset).  Returns 1 when it launched, 0 when the caller should run the
// three launches, < 0 on error.
static int try_fused_wrapper_step(mcbs_batch* b, const int64_t* multidiscrete, const int64_t* discrete, int32_t* decoded, const StepIO& io,
                                  const mcbs_obs_buffers* o, const mcbs_wrapper_buffers* w, float modifier, int32_t max_timesteps, int32_t auto_reset,
                                  const mcbs_row_copies* keep, const mcbs_row_copies* fresh, hipStream_t st) {
    const mcbs_topo_header* h = b->topo->H();
    const uint32_t Nm = b->cfg.maximum_node_count, Cm = b->cfg.maximum_total_credentials, K = b->cfg.maximum_discoverable_credentials_per_action;
    if (!fused_wrapper_batch_ok(b) || o->mask_local || o->mask_remote || o->mask_connect || o->mask_discrete) return 0;
    const uint32_t NP = b->C.n_props;
    FusedArgs A{};
    A.w = *w; A.modifier = modifier; A.max_timesteps = max_timesteps; A.auto_reset = auto_reset;
    A.Nmax = Nm; A.Cmax = Cm; A.K = K; A.NP = NP;
    A.md = multidiscrete; A.discrete = discrete; A.decoded = decoded;
    int32_t* fields[FUSED_FIELDS] = {o->scalars, o->leaked_credentials, o->credential_cache_matrix, o->discovered_nodes_properties, o->nodes_privilegelevel};
    const uint32_t dw[FUSED_FIELDS] = {7u, K * 4u, Cm * 2u, Nm * NP, Nm};
    uint32_t off = 0;
    for (uint32_t f = 0; f < FUSED_FIELDS; ++f) {
        A.obs[f] = fields[f]; A.dwords[f] = dw[f]; A.fresh_off[f] = off;
        const bool vec = f > 0 && dw[f] % 4u == 0 && !(f == 3 && NP < 4u);          // whole 16-byte vectors per env row, else dword by dword
        A.d_u4[f] = fast_div_host(vec ? dw[f] / 4u : (dw[f] ? dw[f] : 1u));
        if (!fields[f]) continue;
        if (dw[f] == 0 || (vec && reinterpret_cast<uintptr_t>(fields[f]) % 16u != 0)) return 0;
        if (auto_reset) {
            int ki = -1, fi = -1;
            for (uint32_t i = 0; keep && i < keep->n; ++i) if (keep->src[i] == fields[f] && keep->row_bytes[i] == 4ull * dw[f]) ki = (int)i;
            for (uint32_t i = 0; fresh && i < fresh->n; ++i) if (fresh->dst[i] == fields[f] && fresh->row_bytes[i] == 4ull * dw[f]) fi = (int)i;
            if (ki < 0 || fi < 0) return 0;
            A.term[f] = static_cast<int32_t*>(keep->dst[ki]);
            A.fresh[f] = static_cast<const int32_t*>(fresh->src[fi]);
            if (vec && reinterpret_cast<uintptr_t>(A.term[f]) % 16u != 0) return 0;
        }
        off += (dw[f] + 3u) & ~3u;
    }
    if (off > FUSED_FRESH_DWORDS) return 0;
    if (auto_reset) {          // every array the generic finish would copy must be one the fused kernel handles (no extra fields)
        uint32_t n_fields = 0;
        for (uint32_t f = 0; f < FUSED_FIELDS; ++f) n_fields += fields[f] ? 1u : 0u;
        if ((keep ? keep->n : 0u) != n_fields || (fresh ? fresh->n : 0u) != n_fields) return 0;
    }
    A.dNP = fast_div_host(NP ? NP : 1u);
    A.digest = b->digest; A.reset_digest = b->reset_digest;
    A.triples = reinterpret_cast<const mcbs_triple*>(b->topo->dev + h->off_triple);
    A.n_triples = h->n_triples;
    b->all_fresh = false;
    const dim3 grid((b->S.E + 63u) / 64u), block(FUSED_THREADS);
    switch (b->cfg.defender_kind) {
    case MCBS_DEFENDER_SCAN_AND_REIMAGE: hipLaunchKernelGGL((wrapper_fused_kernel<MCBS_DEFENDER_SCAN_AND_REIMAGE>), grid, block, 0, st, b->S, b->T, b->C_dev, io, A); break;
    case MCBS_DEFENDER_EXTERNAL: hipLaunchKernelGGL((wrapper_fused_kernel<MCBS_DEFENDER_EXTERNAL>), grid, block, 0, st, b->S, b->T, b->C_dev, io, A); break;
    default: hipLaunchKernelGGL((wrapper_fused_kernel<MCBS_DEFENDER_NONE>), grid, block, 0, st, b->S, b->T, b->C_dev, io, A); break;
    }
    const int rc = launch_ok("wrapper step (one launch)");
    if (rc) return rc;
    b->digest_state = 1;       // every env's digest was written (or, for an intercepted action, stands)
    return 1;
}

extern "C" int mcbs_attacker_wrapper_step(mcbs_batch* b, const int64_t* multidiscrete, const int64_t* discrete, int32_t* decoded,
                                          const mcbs_info_buffers* info, const mcbs_obs_buffers* obs, const mcbs_wrapper_buffers* w, float modifier,
                                          int32_t max_timesteps, int32_t auto_reset, const mcbs_row_copies* keep, const mcbs_row_copies* fresh,
                                          void* stream) {
    if (!b || !w || !decoded || !obs) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    WrapperFinishArgs A;
    int rc = finish_args(b, w, modifier, max_timesteps, auto_reset, keep, fresh, &A);      // (all checks before the first launch)
    if (rc) return rc;
    if (b->cfg.rng_kind == MCBS_RNG_TAPE && b->cfg.defender_kind != MCBS_DEFENDER_NONE && !b->tape)
        return fail(MCBS_ESTATE, "rng_kind is TAPE but no draw tape was set (mcbs_set_draw_tape)");
    if (!multidiscrete == !discrete) return fail(MCBS_EINVAL, "need exactly one action encoding");
    hipStream_t st = (hipStream_t)stream;
    const StepIO io = make_io(b, decoded, const_cast<float*>(w->reward), const_cast<uint8_t*>(w->terminated), info);
    uint8_t* invalid = const_cast<uint8_t*>(w->invalid);
    if ((rc = try_fused_wrapper_step(b, multidiscrete, discrete, decoded, io, obs, w, modifier, max_timesteps, auto_reset, keep, fresh, st)) != 0)
        return rc < 0 ? rc : MCBS_OK;
    if (b->lds_topo) {                       // (developer switch: the staged variant keeps separate launches)
        if ((rc = mcbs_decode_attacker_actions(b, multidiscrete, discrete, decoded, invalid, stream))) return rc;
        if ((rc = launch_step<1>(b, io, st, "step (attacker phase)"))) return rc;
    } else {
        b->all_fresh = false;
        if (b->S.packed) launch_decode_step1_nw<0>(b, io, multidiscrete, discrete, invalid, st);
        else if (b->S.WT == 1) launch_decode_step1_nw<1>(b, io, multidiscrete, discrete, invalid, st);
        else if (b->S.WT == 2) launch_decode_step1_nw<2>(b, io, multidiscrete, discrete, invalid, st);
        else launch_decode_step1_nw<4>(b, io, multidiscrete, discrete, invalid, st);
        if ((rc = launch_ok("decode + step (attacker phase)"))) return rc;
    }
    if ((rc = launch_obs(b, obs, st))) return rc;
    if (b->lds_topo) {                       // (developer switch: the staged variant keeps the two separate launches)
        if ((rc = launch_step<2>(b, io, st, "step (defender phase)"))) return rc;
        hipLaunchKernelGGL(wrapper_finish_kernel, dim3((b->S.E + 255) / 256), dim3(256), 0, st, b->S, b->T, A);
        return launch_ok("wrapper finish");
    }
    if (b->S.packed) launch_step2_finish_nw<0>(b, io, A, st);
    else if (b->S.WT == 1) launch_step2_finish_nw<1>(b, io, A, st);
    else if (b->S.WT == 2) launch_step2_finish_nw<2>(b, io, A, st);
    else launch_step2_finish_nw<4>(b, io, A, st);
    return launch_ok("step (defender phase) + wrapper finish");
}

extern "C" int mcbs_defender_wrapper_post(mcbs_batch* b, const mcbs_defender_wrapper_buffers* w, const mcbs_defender_wrapper_cfg* cfg, void* stream) {
    if (!b || !w || !cfg) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    const void* const* p = reinterpret_cast<const void* const*>(w);
    for (size_t i = 0; i < sizeof(*w) / sizeof(void*); ++i) if (!p[i]) return fail(MCBS_EINVAL, "mcbs_defender_wrapper_buffers: every array is required");
    hipLaunchKernelGGL(defender_wrapper_post_kernel, dim3((b->S.E + 255) / 256), dim3(256), 0, (hipStream_t)stream, b->S.E, *w, *cfg);
    return launch_ok("defender wrapper post");
}

extern "C" int mcbs_step_info(mcbs_batch* b, const mcbs_info_buffers* info, void* stream) {
    if (!b || !info) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    StepIO io = make_io(b, nullptr, nullptr, nullptr, info);
    hipLaunchKernelGGL(info_kernel, dim3((b->S.E + 255) / 256), dim3(256), 0, (hipStream_t)stream, b->S, io);
    return launch_ok("info");
}

extern "C" int mcbs_sample_actions(mcbs_batch* b, int32_t valid, uint64_t seed, uint64_t step, int32_t* actions_out, void* stream) {
    if (!b || !actions_out) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    hipLaunchKernelGGL(sample_kernel, dim3((b->S.E + 127) / 128), dim3(128), 0, (hipStream_t)stream, b->S, b->T, b->C_dev, (int)valid, seed, step,
                       b->cfg.maximum_node_count, b->cfg.maximum_total_credentials, actions_out);
    return launch_ok("sample");
}

extern "C" int mcbs_decode_attacker_actions(mcbs_batch* b, const int64_t* multidiscrete, const int64_t* discrete,
                                            int32_t* actions_out, uint8_t* invalid_out, void* stream) {
    if (!b || !actions_out || !invalid_out || (!multidiscrete == !discrete)) return fail(MCBS_EINVAL, "need exactly one action encoding and both outputs");
    MCBS_ON_DEVICE(b);
    hipLaunchKernelGGL(decode_kernel, dim3((b->S.E + 255) / 256), dim3(256), 0, (hipStream_t)stream, b->S, b->C_dev,
                       b->cfg.maximum_node_count, b->cfg.maximum_total_credentials, multidiscrete, discrete, actions_out, invalid_out);
    return launch_ok("decode");
}

// ------------------------------------------------------------------ action mask -> logits
static FastDiv fast_div_host(uint32_t d) {   // n / d = (t + ((n - t) >> sh1)) >> sh2 with t = mulhi(n, mul)
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    FastDiv f;
    f.mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1ull);
    f.sh1 = l < 1u ? l : 1u;
    f.sh2 = l > 0u ? l - 1u : 0u;
    return f;
}

extern "C" uint64_t mcbs_discrete_action_count(const mcbs_batch* b) {
    if (!b) return 0;
    const uint64_t N = b->cfg.maximum_node_count, C = b->cfg.maximum_total_credentials;
    return N * N * b->C.P * C + N * b->C.L + N * N * b->C.R;
}

extern "C" int mcbs_mask_logits(mcbs_batch* b, void* logits, int32_t dtype, size_t row_stride, float fill, void* stream) {
    if (!b || !logits) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    if (dtype != MCBS_LOGITS_F32 && dtype != MCBS_LOGITS_BF16) return fail(MCBS_EINVAL, "logits dtype must be MCBS_LOGITS_F32 or MCBS_LOGITS_BF16");
    if (b->cfg.defender_kind == MCBS_DEFENDER_RANDOM_EVENTS)
        return fail(MCBS_ESTATE, "mcbs_mask_logits: under ExternalRandomEvents a node's local-vulnerability mask changes with the defender's "
                                 "edits and is not part of the observation digest; use the materialised mask (mcbs_observe)");
    if (b->digest_state != 1)
        return fail(MCBS_ESTATE, b->digest_state == 0 ? "mcbs_mask_logits: no observation has been taken since the batch was created, wholly reset or given a state "
                                                        "(the mask is rebuilt from the digest the last observation left per env)"
                                                      : "mcbs_mask_logits: envs were reset by mask and not re-observed (mcbs_observe_masked) since");
    const uint64_t A64 = mcbs_discrete_action_count(b);
    if (A64 >= (1ull << 31)) return fail(MCBS_ELIMIT, "Discrete action space too large for one launch");
    if (row_stride < A64) return fail(MCBS_EINVAL, "row_stride %zu is shorter than the %llu Discrete actions", row_stride, (unsigned long long)A64);
    LogitsGeom G;
    G.N = b->cfg.maximum_node_count; G.C = b->cfg.maximum_total_credentials; G.L = b->C.L; G.R = b->C.R; G.RL = b->C.P * G.C;
    G.M = G.N * G.N * G.RL; G.ML = G.N * G.L; G.A = (uint32_t)A64;
    G.dRL = fast_div_host(G.RL); G.dC = fast_div_host(G.C); G.dN = fast_div_host(G.N); G.dL = fast_div_host(G.L); G.dR = fast_div_host(G.R);
    hipStream_t st = (hipStream_t)stream;
    const uintptr_t p0 = reinterpret_cast<uintptr_t>(logits);
    // one wavefront per env, four per workgroup; very large action spaces split their chunks of 64 spans over grid.x
    auto grid_for = [&](uint32_t gw) { const uint32_t chunks = (((G.A + gw - 1u) / gw + 15u + 63u) / 64u + 63u) / 64u;   /* spans are shifted by up to 15 groups */ return dim3(chunks < 64u ? chunks : 64u, (b->S.E + 3u) / 4u); };
    const dim3 block(256);
#define MCBS_LOGITS_LAUNCH(LT_, GW_, VEC_, PTR_, FILL_) \
    hipLaunchKernelGGL((mask_logits_kernel<LT_, GW_, VEC_>), grid_for(GW_), block, 0, st, b->S, b->T, b->C_dev, b->digest, PTR_, row_stride, FILL_, G)
    if (dtype == MCBS_LOGITS_F32) {
        float* lp = static_cast<float*>(logits);
        if ((row_stride * 4) % 16 == 0 && p0 % 16 == 0) MCBS_LOGITS_LAUNCH(float, 4u, true, lp, fill);
        else MCBS_LOGITS_LAUNCH(float, 4u, false, lp, fill);
    } else {
        uint32_t bits;                                   // float -> bfloat16, round to nearest even
        memcpy(&bits, &fill, 4);
        const uint16_t f16 = (bits & 0x7FFFFFFFu) > 0x7F800000u ? (uint16_t)((bits >> 16) | 0x40u) : (uint16_t)((bits + 0x7FFFu + ((bits >> 16) & 1u)) >> 16);
        uint16_t* lp = static_cast<uint16_t*>(logits);
        if ((row_stride * 2) % 16 == 0 && p0 % 16 == 0) MCBS_LOGITS_LAUNCH(uint16_t, 8u, true, lp, f16);
        else if ((row_stride * 2) % 8 == 0 && p0 % 8 == 0) MCBS_LOGITS_LAUNCH(uint16_t, 4u, true, lp, f16);     // rows only 8-byte aligned (Chain-10: 14 172 actions)
        else MCBS_LOGITS_LAUNCH(uint16_t, 4u, false, lp, f16);
    }
#undef MCBS_LOGITS_LAUNCH
    return launch_ok("mask logits");
}

// ------------------------------------------------------------------ learned defender
// The observation written by the turn kernel's own workgroups (one launch per turn): topologies of up to 32 nodes and 256 services,
// output arrays on 16-byte boundaries (the 128 envs of a workgroup are one contiguous region of each array, stored as 16-byte vectors).
// MCBS_NO_FUSED_DEFENDER_OBS=1: the separate launch.
static DefObs fused_defender_obs(const mcbs_batch* b, const mcbs_defender_obs* o) {
    DefObs d{};
    if (!o) return d;
    d.infected = o->infected_nodes; d.fw_in = o->incoming_firewall_status; d.fw_out = o->outgoing_firewall_status; d.services = o->services_status;
    d.n_services = b->C.n_services;
    const uint32_t N = b->S.N ? b->S.N : 1u;
    d.dN = fast_div_host(N); d.d6N = fast_div_host(6u * N); d.dS = fast_div_host(d.n_services ? d.n_services : 1u);
    auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    const bool aligned = al(d.infected) && al(d.fw_in) && al(d.fw_out) && al(d.services);
    d.fused = (b->S.N <= 32u && d.n_services <= 256u && aligned && !b->no_fused_defender_obs) ? 1u : 0u;
    return d;
}

static int launch_defender_obs(mcbs_batch* b, const mcbs_defender_obs* o, hipStream_t st) {
    const uint32_t total = b->S.E * b->S.N;
    hipLaunchKernelGGL(defender_obs_kernel, dim3((total + 255) / 256), dim3(256), 0, st, b->S, b->T, o->infected_nodes,
                       o->incoming_firewall_status, o->outgoing_firewall_status, o->services_status, b->C.n_services);
    return launch_ok("defender observation");
}

extern "C" int mcbs_defender_step(mcbs_batch* b, const int64_t* actions, uint8_t* valid, double* availability, uint8_t* evicted,
                                  const mcbs_defender_obs* obs, void* stream) {
    if (!b || !actions) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    if (b->cfg.defender_kind != MCBS_DEFENDER_EXTERNAL) return fail(MCBS_ESTATE, "batch was not created with MCBS_DEFENDER_EXTERNAL");
    hipStream_t st = (hipStream_t)stream;
    b->all_fresh = false;
    const dim3 grid((b->S.E + 127) / 128), block(128);
    const DefObs dob = fused_defender_obs(b, obs);
    if (b->S.WT == 1) hipLaunchKernelGGL((defender_kernel<1>), grid, block, 0, st, b->S, b->T, b->C_dev, actions, valid, availability, evicted, dob);
    else if (b->S.WT == 2) hipLaunchKernelGGL((defender_kernel<2>), grid, block, 0, st, b->S, b->T, b->C_dev, actions, valid, availability, evicted, dob);
    else hipLaunchKernelGGL((defender_kernel<4>), grid, block, 0, st, b->S, b->T, b->C_dev, actions, valid, availability, evicted, dob);
    int rc = launch_ok("defender step");
    if (rc || !obs || dob.fused) return rc;
    return launch_defender_obs(b, obs, st);
}

extern "C" int mcbs_defender_wrapper_step(mcbs_batch* b, const int64_t* actions, const mcbs_defender_obs* obs, const mcbs_defender_wrapper_buffers* w,
                                          const mcbs_defender_wrapper_cfg* cfg, void* stream) {
    if (!b || !actions || !w || !cfg) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    if (b->cfg.defender_kind != MCBS_DEFENDER_EXTERNAL) return fail(MCBS_ESTATE, "batch was not created with MCBS_DEFENDER_EXTERNAL");
    const void* const* p = reinterpret_cast<const void* const*>(w);
    for (size_t i = 0; i < sizeof(*w) / sizeof(void*); ++i) if (!p[i]) return fail(MCBS_EINVAL, "mcbs_defender_wrapper_buffers: every array is required");
    hipStream_t st = (hipStream_t)stream;
    b->all_fresh = false;
    const dim3 grid((b->S.E + 127) / 128), block(128);
    const DefObs dob = fused_defender_obs(b, obs);
    if (b->S.WT == 1) hipLaunchKernelGGL((defender_turn_post_kernel<1>), grid, block, 0, st, b->S, b->T, b->C_dev, actions, *w, *cfg, dob);
    else if (b->S.WT == 2) hipLaunchKernelGGL((defender_turn_post_kernel<2>), grid, block, 0, st, b->S, b->T, b->C_dev, actions, *w, *cfg, dob);
    else hipLaunchKernelGGL((defender_turn_post_kernel<4>), grid, block, 0, st, b->S, b->T, b->C_dev, actions, *w, *cfg, dob);
    int rc = launch_ok("defender turn + reward shaping");
    if (rc || !obs || dob.fused) return rc;
    return launch_defender_obs(b, obs, st);
}

extern "C" int mcbs_defender_observe(mcbs_batch* b, const mcbs_defender_obs* obs, void* stream) {
    if (!b || !obs) return fail(MCBS_EINVAL, "null argument");
    MCBS_ON_DEVICE(b);
    if (b->cfg.defender_kind != MCBS_DEFENDER_EXTERNAL) return fail(MCBS_ESTATE, "batch was not created with MCBS_DEFENDER_EXTERNAL");
    return launch_defender_obs(b, obs, (hipStream_t)stream);
}

// ------------------------------------------------------------------ state export / import (debug, synchronous)
extern "C" size_t mcbs_state_record_bytes(const mcbs_batch* b) {
    if (!b) return 0;
    const size_t n = sizeof(mcbs_state_header) + sizeof(mcbs_state_node) * b->S.N + 2ull * b->S.N + 2ull * b->cfg.maximum_total_credentials;
    return align_up(n, 16);
}

extern "C" int mcbs_get_state(mcbs_batch* b, void* host_buf, size_t nbytes) {
    if (!b || !host_buf) return fail(MCBS_EINVAL, "null argument");
    const size_t rb = mcbs_state_record_bytes(b);
    const DevState& S = b->S;
    if (nbytes < rb * S.E) return fail(MCBS_EINVAL, "state buffer too small: need %zu bytes", rb * S.E);
    DeviceGuard guard(b->cfg.device);
    HIP_TRY(guard.err);
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint8_t> host(b->arena_bytes);
    HIP_TRY(hipMemcpy(host.data(), b->arena, b->arena_bytes, hipMemcpyDeviceToHost));
    auto at = [&](const void* devptr) { return host.data() + (static_cast<const uint8_t*>(devptr) - b->arena); };
    const uint4* h0 = reinterpret_cast<const uint4*>(at(S.h0));
    const double2* h1 = reinterpret_cast<const double2*>(at(S.h1));
    const uint32_t* ep = reinterpret_cast<const uint32_t*>(at(S.episode));
    const void* mk = at(S.masks);
    auto has = [&](int k, uint32_t n, uint32_t e) { return ((S.set_word(mk, k, n >> 6, e) >> (n & 63u)) & 1ull) != 0; };
    (void)has;
    const uint64_t* ring = S.ring ? reinterpret_cast<const uint64_t*>(at(S.ring)) : nullptr;
    const uint8_t* body = at(S.body);
    memset(host_buf, 0, rb * S.E);
    for (uint32_t e = 0; e < S.E; ++e) {
        uint8_t* p = static_cast<uint8_t*>(host_buf) + rb * e;
        mcbs_state_header* sh = reinterpret_cast<mcbs_state_header*>(p);
        const uint32_t flags = h0[e].y;
        sh->step_count = h0[e].x; sh->done = flags & F_DONE ? 1 : 0; sh->truncated = flags & F_TRUNC ? 1 : 0; sh->episode = ep[e];
        sh->n_discovered = h0[e].z & 0xFFFFu; sh->n_creds = h0[e].z >> 16;
        sh->last_outcome_kind = (flags >> F_KIND_SHIFT) & 0xFu; sh->last_escalation = (flags >> F_LEVEL_SHIFT) & 3u;
        sh->last_new_nodes = (flags >> F_NEWNODES_SHIFT) & 0x3FFu; sh->last_new_creds = (flags >> F_NEWCREDS_SHIFT) & 0x3FFu;
        sh->last_oob = flags & F_OOB ? 1 : 0;
        sh->cum_reward = h1[e].x; sh->availability = h1[e].y;
        mcbs_state_node* sn = reinterpret_cast<mcbs_state_node*>(p + sizeof(mcbs_state_header));
        const uint8_t* eb = body + (size_t)e * S.body_stride;
        for (uint32_t n = 0; n < S.N; ++n) {
            const Row r = S.row_get(eb, n);
            const uint64_t bit = 1ull << (n & 63u);
            sn[n].discovered_props = r.props_tags & ROW_PROPS_MASK; sn[n].attacked_ever = r.ever; sn[n].attacked_since = r.since;
            sn[n].discovered = has(M_DISC, n, e); sn[n].installed = has(M_INST, n, e);
            sn[n].ever_owned = has(M_EVER, n, e); sn[n].running = has(M_RUN, n, e);
            sn[n].privilege = (uint8_t)((has(M_PLO, n, e) ? 1 : 0) | (has(M_PHI, n, e) ? 2 : 0));
            sn[n].tags = (uint8_t)(r.props_tags >> 60);
            sn[n].countdown = 0;
            if (!sn[n].running && ring) {   // remaining steps = distance from the defender clock to the node's ring slot
                const uint32_t d = (h0[e].w >> 16) & 15u;
                for (uint32_t s = 0; s < 16u; ++s)
                    if (ring[((size_t)s * S.WT + (n >> 6)) * S.E + e] & bit) sn[n].countdown = (uint8_t)((s + 16u - d) & 15u);
            }
        }
        uint16_t* order = reinterpret_cast<uint16_t*>(p + sizeof(mcbs_state_header) + sizeof(mcbs_state_node) * S.N);
        for (uint32_t k = 0; k < S.N; ++k) order[k] = k < sh->n_discovered ? eb[S.off_disc + k] : 0xFFFF;
        uint16_t* cc = order + S.N;
        const uint16_t* cl = reinterpret_cast<const uint16_t*>(eb + S.off_cred);
        for (uint32_t k = 0; k < b->cfg.maximum_total_credentials; ++k) cc[k] = k < sh->n_creds ? cl[k] : 0xFFFF;
    }
    return MCBS_OK;
}

extern "C" int mcbs_set_state(mcbs_batch* b, const void* host_buf, size_t nbytes) {
    if (!b || !host_buf) return fail(MCBS_EINVAL, "null argument");
    const size_t rb = mcbs_state_record_bytes(b);
    const DevState& S = b->S;
    if (nbytes < rb * S.E) return fail(MCBS_EINVAL, "state buffer too small: need %zu bytes", rb * S.E);
    const mcbs_topo_header* th = b->topo->H();
    DeviceGuard guard(b->cfg.device);
    HIP_TRY(guard.err);
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint8_t> host(b->arena_bytes);
    HIP_TRY(hipMemcpy(host.data(), b->arena, b->arena_bytes, hipMemcpyDeviceToHost));   // keeps init image, digest, episode
    auto at = [&](const void* devptr) { return host.data() + (static_cast<const uint8_t*>(devptr) - b->arena); };
    uint4* h0 = reinterpret_cast<uint4*>(at(S.h0));
    double2* h1 = reinterpret_cast<double2*>(at(S.h1));
    uint32_t* ep = reinterpret_cast<uint32_t*>(at(S.episode));
    void* mk = at(S.masks);
    uint64_t* cach_h = S.wide ? reinterpret_cast<uint64_t*>(at(S.cach)) : nullptr;
    auto add = [&](int k, uint32_t n, uint32_t e) {
        if (cach_h && k == M_CACH) cach_h[(size_t)(n >> 6) * S.E + e] |= 1ull << (n & 63u);
        else S.put_word(mk, k, n >> 6, e, S.set_word(mk, k, n >> 6, e) | (1ull << (n & 63u)));
    };
    uint64_t* ring = S.ring ? reinterpret_cast<uint64_t*>(at(S.ring)) : nullptr;
    uint8_t* body = at(S.body);
    const mcbs_triple* tr = reinterpret_cast<const mcbs_triple*>(b->topo->host.data() + th->off_triple);
    for (uint32_t e = 0; e < S.E; ++e) {
        const uint8_t* p = static_cast<const uint8_t*>(host_buf) + rb * e;
        const mcbs_state_header* sh = reinterpret_cast<const mcbs_state_header*>(p);
        const mcbs_state_node* sn = reinterpret_cast<const mcbs_state_node*>(p + sizeof(mcbs_state_header));
        const uint16_t* order = reinterpret_cast<const uint16_t*>(p + sizeof(mcbs_state_header) + sizeof(mcbs_state_node) * S.N);
        const uint16_t* cc = order + S.N;
        if (sh->n_discovered > S.N || sh->n_creds > th->n_triples) return fail(MCBS_EINVAL, "env %u: list lengths out of range", e);
        for (uint32_t w = 0; w < S.WT; ++w) {
            for (int k = 0; k < M_COUNT; ++k) S.put_word(mk, k, w, e, 0ull);
            if (cach_h && w == 0) for (uint32_t cw = 0; cw < S.TW; ++cw) cach_h[(size_t)cw * S.E + e] = 0ull;
            if (ring) for (uint32_t s = 0; s < 16u; ++s) ring[((size_t)s * S.WT + w) * S.E + e] = 0;
        }
        uint8_t* eb = body + (size_t)e * S.body_stride;
        if (b->cfg.defender_kind == MCBS_DEFENDER_RANDOM_EVENTS) {   // the canonical record does not carry the random-events overlay
            const uint8_t* init = at(S.init_body);                   // (vulnerability keys, service bits, rule lists): back to the
            memcpy(eb + b->C.ere_off_present, init + b->C.ere_off_present, S.body_stride - b->C.ere_off_present);   // topology's initial ones
        }
        uint32_t owned = 0;
        const uint32_t dclk = h0[e].w >> 16;      // the defender clock is not part of the canonical record: keep the current phase
        for (uint32_t n = 0; n < S.N; ++n) {
            const uint64_t bit = 1ull << (n & 63u);
            if (sn[n].privilege > 3 || sn[n].countdown > 15) return fail(MCBS_EINVAL, "env %u node %u: bad privilege / countdown", e, n);
            if (sn[n].discovered) add(M_DISC, n, e);
            if (sn[n].installed) add(M_INST, n, e);
            if (sn[n].ever_owned) add(M_EVER, n, e);
            if (sn[n].running) add(M_RUN, n, e);
            if (sn[n].privilege & 1) add(M_PLO, n, e);
            if (sn[n].privilege & 2) add(M_PHI, n, e);
            owned += sn[n].privilege >= 1;
            if (!sn[n].running && ring) ring[((size_t)((dclk + sn[n].countdown) & 15u) * S.WT + (n >> 6)) * S.E + e] |= bit;
            Row r{};
            r.props_tags = (sn[n].discovered_props & ROW_PROPS_MASK) | ((uint64_t)(sn[n].tags & 0xFu) << 60);
            r.ever = sn[n].attacked_ever; r.since = sn[n].attacked_since;
            S.row_put(eb, n, r);
        }
        for (uint32_t i = 0; i < sh->n_discovered; ++i) {
            if (order[i] >= S.N) return fail(MCBS_EINVAL, "env %u: discovery order entry out of range", e);
            eb[S.off_disc + i] = (uint8_t)order[i];
        }
        uint16_t* cl = reinterpret_cast<uint16_t*>(eb + S.off_cred);
        for (uint32_t i = 0; i < sh->n_creds; ++i) {
            if (cc[i] >= th->n_triples) return fail(MCBS_EINVAL, "env %u: credential cache entry out of range", e);
            cl[i] = cc[i];
            add(M_CACH, cc[i], e);
            const uint32_t c = tr[cc[i]].cred;                     // a cached triple implies its credential string is gathered
            add(M_GATH, c, e);
        }
        const uint32_t flags = (sh->done ? F_DONE : 0u) | (sh->truncated ? F_TRUNC : 0u) | (sh->last_oob ? F_OOB : 0u) |
                               ((sh->last_outcome_kind & 0xFu) << F_KIND_SHIFT) | ((sh->last_escalation & 3u) << F_LEVEL_SHIFT) |
                               ((sh->last_new_nodes & 0x3FFu) << F_NEWNODES_SHIFT) | ((sh->last_new_creds & 0x3FFu) << F_NEWCREDS_SHIFT);
        h0[e] = make_uint4(sh->step_count, flags, sh->n_discovered | (sh->n_creds << 16), owned | (dclk << 16));
        h1[e] = make_double2(sh->cum_reward, sh->availability);
        ep[e] = sh->episode;
    }
    HIP_TRY(hipMemcpy(b->arena, host.data(), b->arena_bytes, hipMemcpyHostToDevice));
    b->all_fresh = false;
    b->digest_state = 0;
    return MCBS_OK;
}

// diagnostic builds (-DMCBS_DIAG): per-wavefront s_memtime stamps of the next step launches; not part of the public ABI
extern "C" int mcbs_diag_set_stamps(mcbs_batch* b, unsigned long long* dev_buf) {
    if (!b) return fail(MCBS_EINVAL, "null batch");
    b->stamps = dev_buf;
    return MCBS_OK;
}

// ------------------------------------------------------------------ timing
extern "C" int mcbs_timing_enable(mcbs_batch* b, int32_t on) {
    if (!b) return fail(MCBS_EINVAL, "null batch");
    b->timing = on != 0;
    b->ev_used = 0; b->timed_ms = 0.0; b->timed_launches = 0;
    return MCBS_OK;
}

extern "C" int mcbs_timing_read(mcbs_batch* b, double* total_ms, uint64_t* launches) {
    if (!b) return fail(MCBS_EINVAL, "null batch");
    MCBS_ON_DEVICE(b);
    for (size_t i = 0; i + 1 < b->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(b->ev[i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, b->ev[i], b->ev[i + 1]));
        b->timed_ms += ms;
        b->timed_launches += 1;
    }
    b->ev_used = 0;
    if (total_ms) *total_ms = b->timed_ms;
    if (launches) *launches = b->timed_launches;
    return MCBS_OK;
}
