"""ctypes binding of libmcbs.so (include/mcbs.h) over PyTorch-ROCm tensors.

PyTorch is plumbing here (device memory, streams); every computation is done by the HIP kernels
behind the C ABI.  There is NO CPU fallback: if the native library is missing or a call fails,
this module raises.  The CPU oracle under oracle/ is test infrastructure and is never imported
from here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, Optional

import numpy as np

from ._abi import BatchCfg, DefenderObs, EnvSpec, InfoBuffers, ObsBuffers, split_state, state_record_bytes
from .flatten import FlatTopology

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmcbs.so")

EXPORTS = [
    "mcbs_last_error", "mcbs_abi_version", "mcbs_topology_create", "mcbs_topology_destroy", "mcbs_batch_create",
    "mcbs_batch_destroy", "mcbs_reset", "mcbs_rewind", "mcbs_step", "mcbs_step_observe", "mcbs_observe", "mcbs_observe_masked", "mcbs_action_mask", "mcbs_step_info",
    "mcbs_step_many", "mcbs_rollout_random", "mcbs_attacker_wrapper_post", "mcbs_attacker_wrapper_clear", "mcbs_defender_wrapper_post", "mcbs_sample_actions", "mcbs_decode_attacker_actions", "mcbs_defender_step", "mcbs_defender_observe", "mcbs_set_draw_tape", "mcbs_state_record_bytes", "mcbs_get_state", "mcbs_set_state",
    "mcbs_timing_enable", "mcbs_timing_read", "mcbs_mask_logits", "mcbs_discrete_action_count", "mcbs_copy_rows_masked", "mcbs_attacker_wrapper_finish", "mcbs_attacker_wrapper_step",
    "mcbs_attacker_wrapper_step_launches", "mcbs_set_mask_discrete_stride", "mcbs_defender_wrapper_step",
]

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


class McbsError(RuntimeError):
    pass


def load_library(path: Optional[str] = None):
    """Load libmcbs.so; raise loudly when it is absent (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MCBS_LIBRARY") or LIB_PATH      # MCBS_LIBRARY: developer override (e.g. the -DMCBS_DIAG build)
    if not os.path.exists(p):
        raise NativeLibraryMissing(
            f"{p} not found: the HIP extension has not been built. Build it with `make -C marlon_amd/csrc` "
            f"(hipcc, gfx950). There is no CPU fallback for the step engine.")
    # torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so), and the first copy a process maps is the one every later
    # library with that soname binds to.  Loaded before torch, libmcbs.so pulled in /opt/rocm's runtime and torch then ran on a runtime it
    # was not built against ("no ROCm-capable device is detected" from the first allocation; build() followed by smoke() in one process).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    lib.mcbs_last_error.restype = C.c_char_p
    lib.mcbs_abi_version.restype = C.c_uint32
    lib.mcbs_topology_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(C.c_void_p)]
    lib.mcbs_topology_destroy.argtypes = [C.c_void_p]
    lib.mcbs_batch_create.argtypes = [C.c_void_p, C.POINTER(BatchCfg), C.POINTER(C.c_void_p)]
    lib.mcbs_batch_destroy.argtypes = [C.c_void_p]
    lib.mcbs_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mcbs_rewind.argtypes = [C.c_void_p, C.c_void_p]
    lib.mcbs_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(InfoBuffers), C.c_void_p]
    lib.mcbs_step_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.mcbs_attacker_wrapper_post.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p]
    lib.mcbs_defender_wrapper_post.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mcbs_attacker_wrapper_clear.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mcbs_rollout_random.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mcbs_step_observe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(InfoBuffers),
                                      C.POINTER(ObsBuffers), C.c_void_p]
    lib.mcbs_observe.argtypes = [C.c_void_p, C.POINTER(ObsBuffers), C.c_void_p]
    lib.mcbs_observe_masked.argtypes = [C.c_void_p, C.POINTER(ObsBuffers), C.c_void_p, C.c_void_p]
    lib.mcbs_action_mask.argtypes = [C.c_void_p, C.POINTER(ObsBuffers), C.c_void_p]
    lib.mcbs_step_info.argtypes = [C.c_void_p, C.POINTER(InfoBuffers), C.c_void_p]
    lib.mcbs_sample_actions.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.mcbs_decode_attacker_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mcbs_defender_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DefenderObs), C.c_void_p]
    lib.mcbs_defender_observe.argtypes = [C.c_void_p, C.POINTER(DefenderObs), C.c_void_p]
    lib.mcbs_defender_wrapper_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mcbs_set_draw_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.mcbs_state_record_bytes.restype = C.c_size_t
    lib.mcbs_state_record_bytes.argtypes = [C.c_void_p]
    lib.mcbs_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.mcbs_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.mcbs_discrete_action_count.restype = C.c_uint64
    lib.mcbs_discrete_action_count.argtypes = [C.c_void_p]
    lib.mcbs_mask_logits.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_size_t, C.c_float, C.c_void_p]
    lib.mcbs_copy_rows_masked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mcbs_attacker_wrapper_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32,
                                               C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mcbs_attacker_wrapper_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mcbs_attacker_wrapper_step_launches.restype = C.c_int32
    lib.mcbs_attacker_wrapper_step_launches.argtypes = [C.c_void_p, C.c_int32]
    lib.mcbs_set_mask_discrete_stride.argtypes = [C.c_void_p, C.c_size_t]
    lib.mcbs_timing_enable.argtypes = [C.c_void_p, C.c_int32]
    lib.mcbs_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise NativeLibraryMissing(f"{p} does not export {name}")
    if path is None:
        _lib = lib
    return lib


def _check(lib, rc: int, what: str) -> None:
    if rc != 0:
        raise McbsError(f"{what} failed ({rc}): {lib.mcbs_last_error().decode(errors='replace')}")


def obs_field_shapes(topo: FlatTopology, spec: EnvSpec) -> Dict[str, tuple]:
    N, Cm, K = spec.maximum_node_count, spec.maximum_total_credentials, spec.maximum_discoverable_credentials_per_action
    L, R, P = len(topo.local_vulnerabilities), len(topo.remote_vulnerabilities), len(topo.ports)
    return {
        "scalars": ((7,), "int32"), "leaked_credentials": ((K, 4), "int32"), "credential_cache_matrix": ((Cm, 2), "int32"),
        "discovered_nodes_properties": ((N, len(topo.properties)), "int32"), "nodes_privilegelevel": ((N,), "int32"),
        "mask_local": ((N, L), "int8"), "mask_remote": ((N, N, R), "int8"), "mask_connect": ((N, N, P, Cm), "int8"),
        "mask_discrete": ((N * N * P * Cm + N * L + N * N * R,), "int8"),
    }


class BatchEngine:
    """n_envs CyberBattleSim environments sharing one topology, advanced on one MI355X."""

    def __init__(self, topo: FlatTopology, spec: EnvSpec, device: Optional[str] = None):
        import torch

        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise McbsError("no HIP device visible to PyTorch: the step engine needs a GPU (there is no CPU fallback)")
        self.device = torch.device(device if device is not None else f"cuda:{spec.device}")
        spec.device = self.device.index or 0
        self.topo, self.spec = topo, spec
        self.E = spec.n_envs
        blob = np.frombuffer(topo.blob, dtype=np.uint8)
        self._topo_h = C.c_void_p()
        _check(self.lib, self.lib.mcbs_topology_create(blob.ctypes.data, blob.size, spec.device, C.byref(self._topo_h)), "mcbs_topology_create")
        self._cfg = spec.to_cfg()
        self._h = C.c_void_p()
        rc = self.lib.mcbs_batch_create(self._topo_h, C.byref(self._cfg), C.byref(self._h))
        if rc != 0:
            msg = self.lib.mcbs_last_error().decode(errors="replace")
            self.lib.mcbs_topology_destroy(self._topo_h)
            self._topo_h = None
            raise (ValueError if rc == -1 else McbsError)(f"mcbs_batch_create failed ({rc}): {msg}")
        self._shapes = obs_field_shapes(topo, spec)
        self.mask_discrete_stride = 0
        self._tape = None
        self._raw_stream = None
        f32, u8, f64, i32 = torch.float32, torch.uint8, torch.float64, torch.int32
        dev = self.device
        self.reward = torch.zeros(self.E, dtype=f32, device=dev)
        self.terminated = torch.zeros(self.E, dtype=u8, device=dev)
        self.info = dict(network_availability=torch.zeros(self.E, dtype=f64, device=dev),
                         step_count=torch.zeros(self.E, dtype=i32, device=dev),
                         truncated=torch.zeros(self.E, dtype=u8, device=dev),
                         out_of_bound=torch.zeros(self.E, dtype=u8, device=dev),
                         raw_reward=torch.zeros(self.E, dtype=f32, device=dev))
        self._info_struct = InfoBuffers(**{k: v.data_ptr() for k, v in self.info.items()})

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.torch.cuda.synchronize(self.device)
            self.lib.mcbs_batch_destroy(self._h)
            self._h = None
        if getattr(self, "_topo_h", None):
            self.lib.mcbs_topology_destroy(self._topo_h)
            self._topo_h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers --
    def _stream(self) -> int:
        """The raw handle of torch's current stream on this engine's device (queried per call: the caller may switch streams)."""
        raw = self._raw_stream
        if raw is None:
            # torch's C binding answers in ~0.2 us; the Python-level current_stream() builds a Stream object first (~1.5 us of a ~10 us call)
            get = getattr(self.torch._C, "_cuda_getCurrentRawStream", None)
            idx = self.device.index if self.device.index is not None else self.torch.cuda.current_device()
            raw = self._raw_stream = (lambda: get(idx)) if get is not None else (lambda: self.torch.cuda.current_stream(self.device).cuda_stream)
        return raw()

    def alloc_obs(self, fields: Optional[Iterable[str]] = None) -> dict:
        t = self.torch
        out = {}
        for f in (fields or [k for k in self._shapes if k != "mask_discrete"]):
            shape, dt = self._shapes[f]
            if f == "mask_discrete" and self.mask_discrete_stride:
                shape = (self.mask_discrete_stride,)          # padded rows (set_mask_discrete_stride): [:, :discrete_action_count()] is the mask
            out[f] = t.zeros((self.E,) + shape, dtype=getattr(t, dt), device=self.device)
        return out

    def set_mask_discrete_stride(self, stride_bytes: int) -> None:
        """Rows of `mask_discrete` buffers handed to this engine are `stride_bytes` apart (0: dense).  A multiple of 128 puts every env's
        mask on cache lines of its own (mcbs_set_mask_discrete_stride); alloc_obs then allocates padded rows."""
        _check(self.lib, self.lib.mcbs_set_mask_discrete_stride(self._h, int(stride_bytes)), "mcbs_set_mask_discrete_stride")
        self.mask_discrete_stride = int(stride_bytes)

    @staticmethod
    def _obs_struct(obs: dict) -> ObsBuffers:
        return ObsBuffers(**{k: v.data_ptr() for k, v in obs.items()})

    def _actions(self, actions):
        t = self.torch
        a = actions if isinstance(actions, t.Tensor) else t.as_tensor(np.asarray(actions), device=self.device)
        a = a.to(device=self.device, dtype=t.int32).contiguous()
        if tuple(a.shape) != (self.E, 5):
            raise ValueError(f"actions must have shape ({self.E}, 5), got {tuple(a.shape)}")
        return a

    # -- the C ABI --
    def reset(self, env_mask=None) -> None:
        ptr = None
        if env_mask is not None:
            env_mask = env_mask.to(device=self.device, dtype=self.torch.uint8).contiguous()
            ptr = env_mask.data_ptr()
        _check(self.lib, self.lib.mcbs_reset(self._h, ptr, self._stream()), "mcbs_reset")

    def rewind(self) -> None:
        """Every env back to the state right after creation (episode counters 0): a recorded trajectory replays exactly, defender draws included."""
        _check(self.lib, self.lib.mcbs_rewind(self._h, self._stream()), "mcbs_rewind")

    def set_draw_tape(self, tape) -> None:
        t = self.torch
        if tape is None:
            self._tape = None
            _check(self.lib, self.lib.mcbs_set_draw_tape(self._h, None, 0), "mcbs_set_draw_tape")
            return
        tape = tape if isinstance(tape, t.Tensor) else t.as_tensor(np.asarray(tape, dtype=np.float64))
        self._tape = tape.to(device=self.device, dtype=t.float64).reshape(self.E, -1).contiguous()
        _check(self.lib, self.lib.mcbs_set_draw_tape(self._h, self._tape.data_ptr(), self._tape.shape[1]), "mcbs_set_draw_tape")

    def step(self, actions, with_info: bool = True):
        """One CyberBattleEnv.step for every env (observation excluded).  Returns (reward, terminated) device tensors,
        overwritten by the next call; self.info holds StepInfo tensors when with_info."""
        a = self._actions(actions)
        _check(self.lib, self.lib.mcbs_step(self._h, a.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(),
                                            C.byref(self._info_struct) if with_info else None, self._stream()), "mcbs_step")
        return self.reward, self.terminated

    def step_many(self, actions, rewards=None, terminated=None):
        """K consecutive steps in one launch: actions [K, E, 5] int32 on the device (a recorded / scripted / pre-sampled sequence).
        Returns (rewards [K, E] float32, terminated [K, E] uint8); same results as K calls of step()."""
        t = self.torch
        a = actions if (isinstance(actions, t.Tensor) and actions.dtype == t.int32 and actions.is_contiguous() and
                        actions.device == self.device) else t.as_tensor(actions, dtype=t.int32, device=self.device).contiguous()
        if a.dim() != 3 or a.shape[1] != self.E or a.shape[2] != 5:
            raise ValueError(f"actions must be [K, {self.E}, 5]")
        K = a.shape[0]
        rewards = t.empty((K, self.E), dtype=t.float32, device=self.device) if rewards is None else rewards
        terminated = t.empty((K, self.E), dtype=t.uint8, device=self.device) if terminated is None else terminated
        _check(self.lib, self.lib.mcbs_step_many(self._h, a.data_ptr(), rewards.data_ptr(), terminated.data_ptr(), K, self._stream()),
               "mcbs_step_many")
        return rewards, terminated

    def rollout_random(self, n_steps: int, valid: bool = True, seed: int = 0, first_step: int = 0, record_actions: bool = False):
        """Random agents for n_steps steps in one launch (actions sampled inside the kernel, the distribution of sample_valid_action
        or uniform).  Returns (rewards [K, E], terminated [K, E], actions [K, E, 5] or None)."""
        t = self.torch
        rewards = t.empty((n_steps, self.E), dtype=t.float32, device=self.device)
        terminated = t.empty((n_steps, self.E), dtype=t.uint8, device=self.device)
        actions = t.empty((n_steps, self.E, 5), dtype=t.int32, device=self.device) if record_actions else None
        _check(self.lib, self.lib.mcbs_rollout_random(self._h, int(bool(valid)), int(seed) & (2 ** 64 - 1), int(first_step), int(n_steps),
                                                      None if actions is None else actions.data_ptr(), rewards.data_ptr(),
                                                      terminated.data_ptr(), self._stream()), "mcbs_rollout_random")
        return rewards, terminated, actions

    def wrapper_post(self, bufs, modifier: float, max_timesteps: int) -> None:
        """AttackerEnvWrapper.step's bookkeeping for every env in one launch; `bufs` is a _abi.WrapperBuffers of device pointers."""
        _check(self.lib, self.lib.mcbs_attacker_wrapper_post(self._h, C.byref(bufs), float(modifier), int(max_timesteps), self._stream()),
               "mcbs_attacker_wrapper_post")

    def defender_wrapper_post(self, bufs, cfg) -> None:
        """DefenderEnvWrapper.step's reward shaping for every env in one launch (_abi.DefenderWrapperBuffers / DefenderWrapperCfg)."""
        _check(self.lib, self.lib.mcbs_defender_wrapper_post(self._h, C.byref(bufs), C.byref(cfg), self._stream()), "mcbs_defender_wrapper_post")

    def copy_rows_masked(self, pairs, env_mask) -> None:
        """dst[e] = src[e] for the envs whose byte in env_mask (uint8 [E], device) is set, for up to eight (src, dst) pairs of
        contiguous [E, ...] tensors in one launch."""
        from ._abi import RowCopies
        if not 0 < len(pairs) <= 8:
            raise ValueError("copy_rows_masked takes one to eight (src, dst) pairs")
        rc = RowCopies()
        rc.n = len(pairs)
        for i, (src, dst) in enumerate(pairs):
            if src.shape != dst.shape or src.dtype != dst.dtype or not (src.is_contiguous() and dst.is_contiguous()) or src.shape[0] != self.E:
                raise ValueError("copy_rows_masked needs contiguous [E, ...] tensors of the same shape and dtype")
            rc.src[i], rc.dst[i], rc.row_bytes[i] = src.data_ptr(), dst.data_ptr(), src[0].numel() * src.element_size()
        _check(self.lib, self.lib.mcbs_copy_rows_masked(self._h, C.byref(rc), env_mask.data_ptr(), self._stream()), "mcbs_copy_rows_masked")

    def _row_copies(self, pairs, one_row_src: bool):
        from ._abi import RowCopies
        if len(pairs) > 8:
            raise ValueError("at most eight (src, dst) pairs per list")
        rc = RowCopies()
        rc.n = len(pairs)
        for i, (src, dst) in enumerate(pairs):
            rows = 1 if one_row_src else self.E
            if src.shape[1:] != dst.shape[1:] or src.dtype != dst.dtype or not (src.is_contiguous() and dst.is_contiguous()) or \
                    src.shape[0] != rows or dst.shape[0] != self.E:
                raise ValueError("row copies need contiguous tensors of the same row shape and dtype ([E, ...]; reset rows: [1, ...])")
            rc.src[i], rc.dst[i], rc.row_bytes[i] = src.data_ptr(), dst.data_ptr(), dst[0].numel() * dst.element_size()
        return rc

    def wrapper_finish(self, bufs, modifier: float, max_timesteps: int, auto_reset: bool, keep=(), fresh=()) -> None:
        """mcbs_attacker_wrapper_finish: the wrapper's bookkeeping and — for the envs it flags done — terminal observation (`keep`:
        (obs, terminal) pairs), env reset, reset observation (`fresh`: ([1, ...] row of a freshly reset env, obs) pairs) and cleared
        counters, all in one launch.  `keep` / `fresh` may be argument blocks built once with `row_copies()`."""
        from ._abi import RowCopies
        k = keep if isinstance(keep, RowCopies) else self._row_copies(list(keep), False)
        f = fresh if isinstance(fresh, RowCopies) else self._row_copies(list(fresh), True)
        _check(self.lib, self.lib.mcbs_attacker_wrapper_finish(self._h, C.byref(bufs), float(modifier), int(max_timesteps), int(bool(auto_reset)),
                                                               C.byref(k), C.byref(f), self._stream()), "mcbs_attacker_wrapper_finish")

    def wrapper_step(self, actions, discrete: bool, decoded, obs_struct, bufs, modifier: float, max_timesteps: int, auto_reset: bool, keep, fresh) -> None:
        """mcbs_attacker_wrapper_step: decode, environment step + observation, bookkeeping / auto-reset — one call, five launches.
        `actions`: int64 device tensor ([E] Discrete or [E, 10] MultiDiscrete); `obs_struct`: obs_struct(obs) built once; `bufs` /
        `keep` / `fresh`: argument blocks built once (WrapperBuffers, row_copies())."""
        p = actions.data_ptr()
        _check(self.lib, self.lib.mcbs_attacker_wrapper_step(self._h, None if discrete else p, p if discrete else None, decoded.data_ptr(),
                                                             C.byref(self._info_struct), C.byref(obs_struct), C.byref(bufs), float(modifier),
                                                             int(max_timesteps), int(bool(auto_reset)), C.byref(keep), C.byref(fresh), self._stream()),
               "mcbs_attacker_wrapper_step")

    def wrapper_step_call(self, discrete: bool, decoded, obs_struct, bufs, modifier: float, max_timesteps: int, auto_reset: bool, keep, fresh):
        """callable(actions_ptr): wrapper_step with every argument but the actions (and the stream, queried per call) bound ONCE — the
        argument blocks' references and conversions are most of what a per-step call costs on the host next to the launch itself."""
        fn, eng, lib, stream = self.lib.mcbs_attacker_wrapper_step, self, self.lib, self._stream       # (eng._h is None once closed: refused)
        refs = (C.byref(self._info_struct), C.byref(obs_struct), C.byref(bufs), C.byref(keep), C.byref(fresh))
        rows, mod, mt, ar = decoded.data_ptr(), float(modifier), int(max_timesteps), int(bool(auto_reset))
        alive = (decoded, obs_struct, bufs, keep, fresh, self._info_struct)       # the blocks the references point into

        def call(p: int, _alive=alive) -> None:
            rc = fn(eng._h, None if discrete else p, p if discrete else None, rows, refs[0], refs[1], refs[2], mod, mt, ar, refs[3], refs[4], stream())
            if rc:
                _check(lib, rc, "mcbs_attacker_wrapper_step")
        return call

    def defender_wrapper_step_call(self, obs_struct, bufs, cfg):
        """callable(actions_ptr): defender_wrapper_step with the argument blocks bound once."""
        fn, eng, lib, stream = self.lib.mcbs_defender_wrapper_step, self, self.lib, self._stream
        refs = (C.byref(obs_struct) if obs_struct is not None else None, C.byref(bufs), C.byref(cfg))
        alive = (obs_struct, bufs, cfg)

        def call(p: int, _alive=alive) -> None:
            rc = fn(eng._h, p, refs[0], refs[1], refs[2], stream())
            if rc:
                _check(lib, rc, "mcbs_defender_wrapper_step")
        return call

    def wrapper_step_launches(self, with_masks: bool) -> int:
        """Kernel launches per mcbs_attacker_wrapper_step of this batch (1: the whole wrapper step is one launch; 3 otherwise)."""
        return int(self.lib.mcbs_attacker_wrapper_step_launches(self._h, int(bool(with_masks))))

    def obs_struct(self, obs: dict):
        """Argument block (_abi.ObsBuffers) for wrapper_step, built once for a set of persistent observation tensors."""
        return self._obs_struct(obs)

    def row_copies(self, pairs, one_row_src: bool = False):
        """Argument block (_abi.RowCopies) for wrapper_finish, built once and reused across steps."""
        return self._row_copies(list(pairs), one_row_src)

    def wrapper_clear(self, bufs) -> None:
        _check(self.lib, self.lib.mcbs_attacker_wrapper_clear(self._h, C.byref(bufs), self._stream()), "mcbs_attacker_wrapper_clear")

    def step_observe(self, actions, obs: dict):
        a = self._actions(actions)
        b = self._obs_struct(obs)
        _check(self.lib, self.lib.mcbs_step_observe(self._h, a.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(),
                                                    C.byref(self._info_struct), C.byref(b), self._stream()), "mcbs_step_observe")
        return self.reward, self.terminated

    def observe(self, obs: dict, env_mask=None) -> dict:
        b = self._obs_struct(obs)
        if env_mask is None:
            _check(self.lib, self.lib.mcbs_observe(self._h, C.byref(b), self._stream()), "mcbs_observe")
        else:
            env_mask = env_mask.to(device=self.device, dtype=self.torch.uint8).contiguous()
            _check(self.lib, self.lib.mcbs_observe_masked(self._h, C.byref(b), env_mask.data_ptr(), self._stream()), "mcbs_observe_masked")
        return obs

    def action_mask(self, masks: dict) -> dict:
        """compute_action_mask for every env: fills the mask_* tensors given (current state, never blank)."""
        b = self._obs_struct({k: v for k, v in masks.items() if k.startswith("mask_")})
        _check(self.lib, self.lib.mcbs_action_mask(self._h, C.byref(b), self._stream()), "mcbs_action_mask")
        return masks

    def step_info(self) -> dict:
        _check(self.lib, self.lib.mcbs_step_info(self._h, C.byref(self._info_struct), self._stream()), "mcbs_step_info")
        return self.info

    def sample_actions(self, valid: bool, seed: int, step: int, out=None):
        t = self.torch
        if out is None:
            out = t.empty((self.E, 5), dtype=t.int32, device=self.device)
        _check(self.lib, self.lib.mcbs_sample_actions(self._h, int(bool(valid)), int(seed), int(step), out.data_ptr(), self._stream()),
               "mcbs_sample_actions")
        return out

    def decode_attacker_actions(self, multidiscrete=None, discrete=None, actions_out=None, invalid_out=None):
        """marlon's MultiDiscrete(10) / Discrete attacker actions -> engine rows [E,5] + invalid flags [E] (device)."""
        t = self.torch
        if (multidiscrete is None) == (discrete is None):
            raise ValueError("give exactly one of multidiscrete / discrete")
        src = multidiscrete if multidiscrete is not None else discrete
        src = src if isinstance(src, t.Tensor) else t.as_tensor(np.asarray(src))
        src = src.to(device=self.device, dtype=t.int64).contiguous()
        want = (self.E, 10) if multidiscrete is not None else (self.E,)
        if tuple(src.shape) != want:
            raise ValueError(f"expected shape {want}, got {tuple(src.shape)}")
        if actions_out is None:
            actions_out = t.empty((self.E, 5), dtype=t.int32, device=self.device)
        if invalid_out is None:
            invalid_out = t.empty(self.E, dtype=t.uint8, device=self.device)
        _check(self.lib, self.lib.mcbs_decode_attacker_actions(
            self._h, src.data_ptr() if multidiscrete is not None else None, src.data_ptr() if discrete is not None else None,
            actions_out.data_ptr(), invalid_out.data_ptr(), self._stream()), "mcbs_decode_attacker_actions")
        return actions_out, invalid_out

    def discrete_action_count(self) -> int:
        """Size of MaskedDiscreteAttackerWrapper's Discrete space: N*N*P*C + N*L + N*N*R (action_masking.py:74-80)."""
        return int(self.lib.mcbs_discrete_action_count(self._h))

    def mask_logits(self, logits, fill: float = -1e8):
        """In place: logits[e, a] = mask(e, a) ? logits[e, a] : fill for the Discrete action mask of the LAST observation call
        (step_observe / observe / action_mask), rebuilt on the device from that call's per-env digest — the mask itself is never
        written.  logits: device float32 or bfloat16 [E, >= discrete_action_count()], rows contiguous."""
        t = self.torch
        if logits.dtype not in (t.float32, t.bfloat16):
            raise ValueError("logits must be float32 or bfloat16")
        if logits.dim() != 2 or logits.shape[0] != self.E or logits.stride(1) != 1 or logits.device != self.device:
            raise ValueError(f"logits must be a device tensor [{self.E}, A] with contiguous rows")
        _check(self.lib, self.lib.mcbs_mask_logits(self._h, logits.data_ptr(), 0 if logits.dtype == t.float32 else 1, logits.stride(0),
                                                   float(fill), self._stream()), "mcbs_mask_logits")
        return logits

    # -- learned defender (batches created with defender=("external",)) --
    def alloc_defender_obs(self) -> dict:
        t, N, S = self.torch, self.topo.n_nodes, int(self.topo.header()["n_services"])
        mk = lambda n: t.zeros((self.E, n), dtype=t.int8, device=self.device)
        return dict(infected_nodes=mk(N), incoming_firewall_status=mk(6 * N), outgoing_firewall_status=mk(6 * N), services_status=mk(S))

    def defender_step(self, actions12, obs: Optional[dict] = None, out=None):
        """DefenderEnvWrapper's validity check + LearningDefender.executeAction for every env.
        -> (valid u8 [E], availability f64 [E], evicted u8 [E]) device tensors (`out`: that triple, caller-owned; default: the engine's own
        buffers, overwritten by the next call)."""
        t = self.torch
        a = actions12 if isinstance(actions12, t.Tensor) else t.as_tensor(np.asarray(actions12))
        a = a.to(device=self.device, dtype=t.int64).contiguous()
        if tuple(a.shape) != (self.E, 12):
            raise ValueError(f"defender actions must have shape ({self.E}, 12), got {tuple(a.shape)}")
        if not hasattr(self, "_def_out"):
            self._def_out = (t.zeros(self.E, dtype=t.uint8, device=self.device), t.zeros(self.E, dtype=t.float64, device=self.device),
                             t.zeros(self.E, dtype=t.uint8, device=self.device))
        v, av, ev = self._def_out if out is None else out
        o = DefenderObs(**{k: x.data_ptr() for k, x in obs.items()}) if obs is not None else None
        _check(self.lib, self.lib.mcbs_defender_step(self._h, a.data_ptr(), v.data_ptr(), av.data_ptr(), ev.data_ptr(),
                                                     C.byref(o) if o is not None else None, self._stream()), "mcbs_defender_step")
        return v, av, ev

    def defender_wrapper_step(self, actions12, obs_struct, bufs, cfg) -> None:
        """mcbs_defender_wrapper_step: the defender's turn + the wrapper's reward shaping in one launch, then the observation.
        `actions12`: int64 device tensor [E, 12]; `obs_struct`: a DefenderObs block (or None); `bufs` / `cfg`: argument blocks built once."""
        _check(self.lib, self.lib.mcbs_defender_wrapper_step(self._h, actions12.data_ptr(), C.byref(obs_struct) if obs_struct is not None else None,
                                                             C.byref(bufs), C.byref(cfg), self._stream()), "mcbs_defender_wrapper_step")

    def defender_observe(self, obs: dict) -> dict:
        o = DefenderObs(**{k: x.data_ptr() for k, x in obs.items()})
        _check(self.lib, self.lib.mcbs_defender_observe(self._h, C.byref(o), self._stream()), "mcbs_defender_observe")
        return obs

    def get_state(self):
        rb = state_record_bytes(self.topo.n_nodes, self.spec.maximum_total_credentials)
        assert rb == self.lib.mcbs_state_record_bytes(self._h)
        buf = np.zeros(rb * self.E, np.uint8)
        _check(self.lib, self.lib.mcbs_get_state(self._h, buf.ctypes.data, buf.size), "mcbs_get_state")
        return split_state(buf, self.E, self.topo.n_nodes, self.spec.maximum_total_credentials)

    def set_state(self, hdr, nodes, order, cache) -> None:
        """Inverse of get_state (parity / debugging): upload canonical state records."""
        from ._abi import STATE_HEADER_DT, STATE_NODE_DT
        N, Cm = self.topo.n_nodes, self.spec.maximum_total_credentials
        rb = state_record_bytes(N, Cm)
        raw = np.zeros((self.E, rb), np.uint8)
        raw[:, :64] = np.ascontiguousarray(hdr.astype(STATE_HEADER_DT)).view(np.uint8).reshape(self.E, 64)
        raw[:, 64:64 + 32 * N] = np.ascontiguousarray(nodes.astype(STATE_NODE_DT)).view(np.uint8).reshape(self.E, 32 * N)
        o = 64 + 32 * N
        raw[:, o:o + 2 * N] = np.ascontiguousarray(order.astype("<u2")).view(np.uint8).reshape(self.E, 2 * N)
        raw[:, o + 2 * N:o + 2 * N + 2 * Cm] = np.ascontiguousarray(cache.astype("<u2")).view(np.uint8).reshape(self.E, 2 * Cm)
        _check(self.lib, self.lib.mcbs_set_state(self._h, raw.ctypes.data, raw.size), "mcbs_set_state")

    def timing_enable(self, on: bool = True) -> None:
        _check(self.lib, self.lib.mcbs_timing_enable(self._h, int(on)), "mcbs_timing_enable")

    def timing_read(self):
        ms, n = C.c_double(), C.c_uint64()
        _check(self.lib, self.lib.mcbs_timing_read(self._h, C.byref(ms), C.byref(n)), "mcbs_timing_read")
        return ms.value, n.value
