// mcbs_aux.hip — reset, StepInfo export and the random-agent action samplers (gfx950).
#pragma once
#include "mcbs_device.h"
#include "mcbs_sample.hip"

namespace mcbs {

// CyberBattleEnv.reset (cyberbattle_env.py:1187-1209) for the envs selected by env_mask (NULL = all).
// Same wave-cooperative copy of the reset image as the auto-reset at the end of step_kernel.
// episode_mode 0: episode counter := 0 (batch creation); 1: episode += 1.
__global__ __launch_bounds__(128) void reset_kernel(DevState S, Topo T, const uint8_t* env_mask, int episode_mode) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool need = e < S.E && (env_mask == nullptr || env_mask[e] != 0);
    const uint64_t rm = __ballot(need);
    if (!rm) return;
    const uint32_t lane = threadIdx.x & 63u, wave_base = e - lane;
    uint64_t m = rm;
    while (m) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        uint8_t* dst = S.body + (size_t)(wave_base + l) * S.body_stride;
        for (uint32_t off = lane * 16u; off < S.body_stride; off += 64u * 16u)
            *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(S.init_body + off);
    }
    if (need) reset_header(S, T, e, episode_mode ? S.episode[e] + 1u : 0u);
}

// StepInfo (cyberbattle_env.py:1176-1182) of the current state
__global__ __launch_bounds__(256) void info_kernel(DevState S, StepIO io) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S.E) return;
    const uint4 h0 = S.h0[e];
    if (io.availability) io.availability[e] = S.h1[e].y;
    if (io.step_count) io.step_count[e] = (int32_t)h0.x;
    if (io.truncated) io.truncated[e] = (h0.y & F_TRUNC) ? 1 : 0;
    if (io.oob) io.oob[e] = (h0.y & F_OOB) ? 1 : 0;
    if (io.raw_reward) io.raw_reward[e] = (float)S.pending[e];
    if (io.terminated) io.terminated[e] = (h0.y & F_DONE) ? 1 : 0;
}

// Random-agent harness: sample_action (mcbs_sample.hip), one lane per env.
__global__ __launch_bounds__(128) void sample_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, int valid, uint64_t seed, uint64_t step,
                                                    uint32_t Nmax, uint32_t Cmax, int32_t* out) {
    const StepCfg& C = *Cp;   // device copy: by value it would push the arguments past 256 bytes (profiles/round1_notes.md)
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S.E) return;
    int32_t a[5];
    sample_action(S, T, C, e, valid, seed, step, Nmax, Cmax, a);
    int32_t* o = out + (size_t)e * 5;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = a[4];
}

// AttackerEnvWrapper.step's decode + out-of-range interception (attack_wrapper.py:255-308, :236-253) and
// MaskedDiscreteAttackerWrapper._decode (action_masking.py:112-142), one lane per env.
__device__ __forceinline__ void decode_body(const DevState& S, const StepCfg& C, uint32_t Nmax, uint32_t Cmax, const int64_t* md, const int64_t* discrete,
                                            int32_t* out, uint8_t* invalid, uint32_t e) {
    const int64_t nd = (int64_t)(S.h0[e].z & 0xFFFFu);
    int64_t kind, a = 0, b = 0, c = 0, d = 0;
    if (md) {
        // all ten components are read and the row is picked with selects: a three-way branch here was
        // mis-structurised by ROCm 7.2's compiler (the kind >= 2 lanes kept uninitialised addresses)
        const int64_t* v = md + (size_t)e * 10;
        int64_t x[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) x[i] = v[i];
        kind = x[0];
        const bool k0 = kind == 0, k1 = kind == 1;
        a = k0 ? x[1] : (k1 ? x[3] : x[6]);
        b = k0 ? x[2] : (k1 ? x[4] : x[7]);
        c = k0 ? 0 : (k1 ? x[5] : x[8]);
        d = (k0 || k1) ? 0 : x[9];
    } else {
        const int64_t N = Nmax, P = C.P, Cm = Cmax, L = C.L, R = C.R;
        const int64_t connect_size = N * N * P * Cm, local_size = N * L;
        const int64_t idx = discrete[e];
        const bool is_c = idx < connect_size, is_l = !is_c && idx < connect_size + local_size;     // selects, no branches (see above)
        const int64_t rel = is_c ? idx : (is_l ? idx - connect_size : idx - connect_size - local_size);
        const int64_t inner = is_c ? Cm : (is_l ? L : R);
        int64_t x0, q, qp, qn, qm, qr;
        if (idx >= 0 && idx < (1ll << 31)) {               // every index a policy can emit: the same quotients in 32-bit arithmetic
            const uint32_t r32 = (uint32_t)rel, i32 = (uint32_t)inner, q32 = r32 / i32, p32 = (uint32_t)P, n32 = (uint32_t)N;
            const uint32_t qp32 = q32 / p32;
            x0 = r32 - q32 * i32; q = q32; qp = qp32; qr = q32 - qp32 * p32;
            qn = is_c ? qp32 / n32 : q32 / n32; qm = is_c ? qp32 - (uint32_t)qn * n32 : q32 - (uint32_t)qn * n32;
        } else {
            x0 = rel % inner; q = rel / inner; qp = q / P; qr = q % P;
            qn = is_c ? qp / N : q / N; qm = is_c ? qp % N : q % N;
        }
        kind = is_c ? 2 : (is_l ? 0 : 1);
        a = is_l ? q : qn;
        b = is_l ? x0 : qm;
        c = is_c ? qr : (is_l ? 0 : x0);
        d = is_c ? x0 : 0;
    }
    bool ok;                                               // _action_in_discovered_range
    if (kind == 0) ok = a < nd;
    else if (kind == 1 || kind == 2) ok = a < nd && b < nd;
    else ok = false;
    int32_t* o = out + (size_t)e * 5;
    o[0] = ok ? (int32_t)kind : MCBS_ACTION_SKIP;
    o[1] = (int32_t)a; o[2] = (int32_t)b; o[3] = (int32_t)c; o[4] = (int32_t)d;
    invalid[e] = ok ? 0 : 1;
}

__global__ __launch_bounds__(256) void decode_kernel(DevState S, const StepCfg* __restrict__ Cp, uint32_t Nmax, uint32_t Cmax, const int64_t* md,
                                                    const int64_t* discrete, int32_t* out, uint8_t* invalid) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;   // (StepCfg by pointer: by value it would push the arguments past 256 bytes)
    if (e < S.E) decode_body(S, *Cp, Nmax, Cmax, md, discrete, out, invalid, e);
}

// AttackerEnvWrapper.step's bookkeeping (attack_wrapper.py:286-354) for the whole batch: one launch instead of ~20 element-wise ones
__global__ __launch_bounds__(256) void wrapper_post_kernel(uint32_t E, mcbs_wrapper_buffers w, float modifier, int32_t max_timesteps) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    bool done = false;
    if (e < E) {
        const bool invalid = w.invalid[e] != 0;
        const float r = w.reward[e];
        const int32_t t = w.timesteps[e] + 1;
        const float shaped = r + (invalid ? modifier : 0.0f);
        const double ret = w.episode_returns[e] + (double)shaped;
        const bool trunc = t >= max_timesteps;
        done = (w.terminated[e] != 0) || trunc;
        w.timesteps[e] = t;
        if (invalid) w.invalid_action_count[e] += 1; else w.valid_action_count[e] += 1;
        w.episode_returns[e] = ret;
        w.last_cyber_reward[e] = r;
        w.has_cyber_reward[e] = 1;
        w.rewards[e] = shaped;
        w.truncated[e] = trunc ? 1 : 0;
        w.dones[e] = done ? 1 : 0;
        w.episode_return_out[e] = ret;
        w.episode_length_out[e] = t;
    }
    const uint64_t m = __ballot(done);
    if (m && (threadIdx.x & 63u) == 0) atomicAdd(w.n_done, (int32_t)__popcll(m));
}

__global__ __launch_bounds__(256) void wrapper_clear_kernel(uint32_t E, mcbs_wrapper_buffers w) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E || !w.dones[e]) return;
    w.timesteps[e] = 0; w.valid_action_count[e] = 0; w.invalid_action_count[e] = 0; w.episode_returns[e] = 0.0; w.has_cyber_reward[e] = 0;
}

// dst[i][r] = src[i][r] for the rows whose mask byte is set: the terminal observation of the envs that just ended (what DummyVecEnv puts
// in infos[i]["terminal_observation"] before it resets the env), without a host round trip to find out which envs those are.
// A wavefront SCANS 64 rows' mask bytes (one coalesced load + ballot) and copies the flagged rows one after the other with all 64
// lanes (16-byte accesses when aligned): E / 64 wavefronts in all, so the launch costs next to nothing when no env ended.
__global__ __launch_bounds__(256) void copy_rows_masked_kernel(mcbs_row_copies rc, const uint8_t* __restrict__ mask, uint32_t n_rows) {
    const uint32_t r0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u, lane = threadIdx.x & 63u;
    if (r0 >= n_rows) return;
    uint64_t m = __ballot(r0 + lane < n_rows && mask[r0 + lane] != 0);
    while (m) {
        const uint32_t r = r0 + (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        for (uint32_t f = 0; f < rc.n; ++f) {
            const size_t nb = rc.row_bytes[f];
            const uint8_t* s = static_cast<const uint8_t*>(rc.src[f]) + (size_t)r * nb;
            uint8_t* d = static_cast<uint8_t*>(rc.dst[f]) + (size_t)r * nb;
            if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15u) == 0) {
                const size_t nv = nb >> 4;
                for (size_t i = lane; i < nv; i += 64u) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
                for (size_t i = (nv << 4) + lane; i < nb; i += 64u) d[i] = s[i];
            } else {
                for (size_t i = lane; i < nb; i += 64u) d[i] = s[i];
            }
        }
    }
}

// mcbs_attacker_wrapper_finish: wrapper_post for every env, then — for the envs it has just flagged done — terminal observation kept,
// env reset (reset_kernel's work), reset observation and digest put in place, counters cleared (wrapper_clear): one wavefront per 64
// envs, the flagged ones handled one after the other by all 64 lanes.
__device__ __forceinline__ void copy_row_wave(const uint8_t* s, uint8_t* d, size_t nb, uint32_t lane) {
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15u) == 0) {
        const size_t nv = nb >> 4;
        for (size_t i = lane; i < nv; i += 64u) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
        for (size_t i = (nv << 4) + lane; i < nb; i += 64u) d[i] = s[i];
    } else {
        for (size_t i = lane; i < nb; i += 64u) d[i] = s[i];
    }
}

struct WrapperFinishArgs {           // by value: kernel arguments of the two kernels below
    mcbs_wrapper_buffers w;
    float modifier;
    int32_t max_timesteps, auto_reset, pad;
    mcbs_row_copies keep, fresh;
    ObsDigest* digest;
    const ObsDigest* reset_digest;
};

// env e = this lane's env (e >= S.E: no env, the lane only helps with the copies); every lane of the wavefront must call it
__device__ __forceinline__ void wrapper_finish_body(const DevState& S, const Topo& T, const WrapperFinishArgs& A, uint32_t e, uint32_t lane) {
    const mcbs_wrapper_buffers& w = A.w;
    bool done = false;
    if (e < S.E) {                                       // wrapper_post_kernel, word for word
        const bool invalid = w.invalid[e] != 0;
        const float r = w.reward[e];
        const int32_t t = w.timesteps[e] + 1;
        const float shaped = r + (invalid ? A.modifier : 0.0f);
        const double ret = w.episode_returns[e] + (double)shaped;
        const bool trunc = t >= A.max_timesteps;
        done = (w.terminated[e] != 0) || trunc;
        const bool clear = done && A.auto_reset;
        w.timesteps[e] = clear ? 0 : t;
        const int64_t nv = w.valid_action_count[e] + (invalid ? 0 : 1), ni = w.invalid_action_count[e] + (invalid ? 1 : 0);
        w.valid_action_count[e] = clear ? 0 : nv;
        w.invalid_action_count[e] = clear ? 0 : ni;
        w.episode_returns[e] = clear ? 0.0 : ret;
        w.last_cyber_reward[e] = r;
        w.has_cyber_reward[e] = clear ? 0 : 1;
        w.rewards[e] = shaped;
        w.truncated[e] = trunc ? 1 : 0;
        w.dones[e] = done ? 1 : 0;
        w.episode_return_out[e] = ret;
        w.episode_length_out[e] = t;
        if (w.executed) w.executed[e] = invalid ? 0 : 1;
    }
    uint64_t m = A.auto_reset ? __ballot(done) : 0ull;
    if (!m) return;
    const uint32_t wave_base = e - lane;
    while (m) {
        const uint32_t r = wave_base + (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        for (uint32_t f = 0; f < A.keep.n; ++f) {        // the episode's last observation (its loads have landed before any store below is issued)
            const size_t nb = A.keep.row_bytes[f];
            copy_row_wave(static_cast<const uint8_t*>(A.keep.src[f]) + (size_t)r * nb, static_cast<uint8_t*>(A.keep.dst[f]) + (size_t)r * nb, nb, lane);
        }
        for (uint32_t f = 0; f < A.fresh.n; ++f) {       // the observation of a freshly reset env
            const size_t nb = A.fresh.row_bytes[f];
            copy_row_wave(static_cast<const uint8_t*>(A.fresh.src[f]), static_cast<uint8_t*>(A.fresh.dst[f]) + (size_t)r * nb, nb, lane);
        }
        uint8_t* dst = S.body + (size_t)r * S.body_stride;
        for (uint32_t off = lane * 16u; off < S.body_stride; off += 64u * 16u)
            *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(S.init_body + off);
        if (lane < 4u) reinterpret_cast<uint4*>(A.digest + r)[lane] = reinterpret_cast<const uint4*>(A.reset_digest)[lane];
    }
    if (done) reset_header(S, T, e, S.episode[e] + 1u);
}

__global__ __launch_bounds__(256) void wrapper_finish_kernel(DevState S, Topo T, WrapperFinishArgs A) {
    wrapper_finish_body(S, T, A, blockIdx.x * blockDim.x + threadIdx.x, threadIdx.x & 63u);
}

// The defender / goals half of a split step (step_kernel<2>) and the wrapper's finish in ONE launch: one-wavefront workgroups as
// launch_step_v uses them (hot image through L1 / L2), lane = env in both halves; the finish reads the reward and the done flag its own
// lane has just stored.  One graph node less per wrapper step.
template <int WTP, int DEFK>
__global__ __launch_bounds__(64) void step2_finish_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, StepIO io, WrapperFinishArgs A) {
    NoHook nh;
    step_body<2, WTP, false, DEFK, false>(S, T, Cp, io, RollArgs{}, nh);
    wrapper_finish_body(S, T, A, blockIdx.x * 64u + threadIdx.x, threadIdx.x);
}

// The wrappers' decode + interception and the attacker half of a split step (step_kernel<1>) in ONE launch: the lane decodes its env's
// policy action into the engine row (and the invalid flag), stores it, and the step reads it back — same lane, same address.
template <int WTP, int DEFK>
__global__ __launch_bounds__(64) void decode_step1_kernel(DevState S, Topo T, const StepCfg* __restrict__ Cp, StepIO io, uint32_t Nmax, uint32_t Cmax,
                                                         const int64_t* md, const int64_t* discrete, uint8_t* invalid) {
    const uint32_t e = blockIdx.x * 64u + threadIdx.x;
    if (e < S.E) decode_body(S, *Cp, Nmax, Cmax, md, discrete, const_cast<int32_t*>(io.actions), invalid, e);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the row's stores stay ahead of the step's loads of it
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    NoHook nh;
    step_body<1, WTP, false, DEFK, false>(S, T, Cp, io, RollArgs{}, nh);
}

} // namespace mcbs
