"""CPU-side checks of the product's native library: it loads, exports every entry point include/mcbs.h declares,
rejects malformed input through the C ABI without touching a GPU, and the Python layer refuses to run without it
(no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from marlon_amd import engine, flatten
from marlon_amd._abi import BatchCfg
from marlon_amd.samples import chainpattern

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_functions():
    text = open(os.path.join(REPO, "include", "mcbs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcbs_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = engine.load_library()
    names = declared_functions()
    assert len(names) >= 18 and set(names) == set(engine.EXPORTS)
    for n in names:
        assert hasattr(lib, n), f"libmcbs.so does not export {n}"
    assert lib.mcbs_abi_version() == 1


def test_abi_struct_sizes_match_header():
    # compiled against include/mcbs.h by gcc in tests/test_oracle_*: here only the Python mirrors
    assert C.sizeof(BatchCfg) == 136
    assert flatten.HEADER_DT.itemsize == 192 and flatten.NODE_DT.itemsize == 64 and flatten.SLOT_DT.itemsize == 32


def test_topology_create_rejects_malformed_blobs_without_gpu():
    lib = engine.load_library()
    out = C.c_void_p()
    junk = np.zeros(64, np.uint8)
    assert lib.mcbs_topology_create(junk.ctypes.data, junk.size, 0, C.byref(out)) == -1
    assert b"too small" in lib.mcbs_last_error()
    blob = np.frombuffer(flatten.flatten(chainpattern.new_environment(4)).blob, np.uint8).copy()
    bad = blob.copy()
    bad[0] ^= 0xFF
    assert lib.mcbs_topology_create(bad.ctypes.data, bad.size, 0, C.byref(out)) == -1
    assert b"magic" in lib.mcbs_last_error()
    bad = blob.copy()
    bad[:192].view(flatten.HEADER_DT)[0]["n_nodes"] = 300
    assert lib.mcbs_topology_create(bad.ctypes.data, bad.size, 0, C.byref(out)) == -2       # MCBS_ELIMIT
    bad = blob.copy()
    hdr = bad[:192].view(flatten.HEADER_DT)[0]
    bad[int(hdr["off_slot_of"])] = 200                                                     # slot index out of range
    assert lib.mcbs_topology_create(bad.ctypes.data, bad.size, 0, C.byref(out)) == -1
    assert lib.mcbs_step(None, None, None, None, None, None) == -1                          # null arguments
    assert lib.mcbs_rewind(None, None) == -1 and lib.mcbs_set_mask_discrete_stride(None, 128) == -1
    assert lib.mcbs_attacker_wrapper_step_launches(None, 0) == 0
    # firewall rule sections (read on the host at batch creation and by the random-events kernels): truncated / inconsistent blobs
    from marlon_amd.samples import toy_ctf
    blob = np.frombuffer(flatten.flatten(toy_ctf.new_environment()).blob, np.uint8).copy()
    hdr = blob[:192].view(flatten.HEADER_DT)[0]
    assert hdr["n_fw_lists"] > 0 and hdr["n_fw_rules"] > 0
    bad = blob.copy()
    bad[:192].view(flatten.HEADER_DT)[0]["off_fw_range"] = (int(hdr["total_bytes"]) // 16) * 16          # section starts at the end of the blob
    assert lib.mcbs_topology_create(bad.ctypes.data, bad.size, 0, C.byref(out)) == -1 and b"fw_range" in lib.mcbs_last_error()
    bad = blob.copy()
    bad[:192].view(flatten.HEADER_DT)[0]["n_fw_rules"] = 1 << 30                                           # rule array far beyond the blob
    assert lib.mcbs_topology_create(bad.ctypes.data, bad.size, 0, C.byref(out)) == -1 and b"fw_rule" in lib.mcbs_last_error()
    bad = blob.copy()
    fr = bad[int(hdr["off_fw_range"]):int(hdr["off_fw_range"]) + 4].view("<u2")
    fr[1] = 60000                                                                                          # list 0 claims 60 000 rules
    assert lib.mcbs_topology_create(bad.ctypes.data, bad.size, 0, C.byref(out)) == -1 and b"rule list" in lib.mcbs_last_error()
    bad = blob.copy()
    bad[int(hdr["off_fw_rule"])] = 250                                                                     # rule 0 names port 250 of n_names
    assert lib.mcbs_topology_create(bad.ctypes.data, bad.size, 0, C.byref(out)) == -1 and b"port name" in lib.mcbs_last_error()


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(engine.NativeLibraryMissing, match="no CPU fallback"):
        engine.load_library(str(tmp_path / "libmcbs.so"))


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "marlon_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "from oracle" not in src and "import oracle" not in src and "cbs_oracle" not in src, f


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="only meaningful on a GPU-less host")
def test_engine_refuses_to_run_without_gpu():
    from marlon_amd._abi import EnvSpec
    topo = flatten.flatten(chainpattern.new_environment(4))
    with pytest.raises(engine.McbsError, match="no CPU fallback"):
        engine.BatchEngine(topo, EnvSpec(n_envs=4, maximum_node_count=6, maximum_total_credentials=6))


def test_vecenv_surface_is_importable_without_gpu_or_sb3():
    """The SB3-VecEnv adapter names what baseline_marlon_agent.py:100-167 calls; importing it needs neither a GPU, SB3 nor gymnasium."""
    from marlon_amd import vecenv
    for cls in (vecenv.MarlonVecEnv, vecenv.DefenderVecEnvAdapter):
        for m in ("reset", "step_async", "step_wait", "step", "env_method", "get_attr", "env_is_wrapped", "seed", "close"):
            assert callable(getattr(cls, m)), (cls.__name__, m)


def test_every_launching_entry_point_selects_the_batch_device():
    """A process whose current device is not the batch's (PyTorch driving several GPUs) must not launch on the wrong GPU: every
    extern "C" entry point of mcbs_api.hip that launches a kernel, enqueues a copy or synchronises takes MCBS_ON_DEVICE(b) (or a
    DeviceGuard of its own) BEFORE the first such statement.  The one-GPU box cannot exercise two devices, so this is checked on
    the source."""
    src = open(os.path.join(REPO, "marlon_amd", "csrc", "mcbs_api.hip")).read()
    heads = list(re.finditer(r'extern "C" (?:int|void|size_t|uint64_t|uint32_t|const char\*) (mcbs_[a-z_]+)\(', src))
    assert len(heads) >= 30
    device_work = re.compile(r"hipLaunchKernelGGL|launch_step|launch_obs|launch_masks|launch_defender_obs|launch_decode_step1|launch_step2_finish|"
                             r"hipMemcpy|hipMemset|hipDeviceSynchronize|hipEventSynchronize|hipFree|hipMalloc|launch_wrapper")
    checked = 0
    for i, m in enumerate(heads):
        body = src[m.end():heads[i + 1].start() if i + 1 < len(heads) else len(src)]
        # cut at the end of the function: the first line that is just "}"
        end = re.search(r"^}\s*$", body, flags=re.M)
        body = body[:end.start()] if end else body
        w = device_work.search(body)
        if not w:
            continue
        g = re.search(r"MCBS_ON_DEVICE\(b\)|DeviceGuard guard\(", body)
        assert g and g.start() < w.start(), f"{m.group(1)} touches the device before selecting the batch's device"
        checked += 1
    assert checked >= 25


def test_library_is_loaded_after_torch():
    """libmcbs.so and PyTorch-ROCm both need libamdhip64, and the copy a process maps first serves both: load_library must bring torch's
    in before the library's own dependency resolves to /opt/rocm's (INTEGRATION.md "One HIP runtime per process").  Fresh interpreter."""
    import subprocess
    import sys
    code = ("import sys; from marlon_amd import engine; assert 'torch' not in sys.modules, 'importing the package must stay light'; "
            "engine.load_library(); assert 'torch' in sys.modules; print('ok')")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=repo, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
