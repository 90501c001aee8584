"""The C-ABI library loads on a GPU-less box and exports every symbol include/viszephyr.h declares."""
import os
import re
import subprocess

import pytest

from util import REPO


def _ensure_built():
    lib = os.path.join(REPO, "vision-zephyr_amd", "lib", "libviszephyr_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "vision-zephyr_amd", "csrc"), "-j4"])
    return lib


def test_header_symbols_are_exported_and_bound():
    lib = _ensure_built()
    hdr = open(os.path.join(REPO, "include", "viszephyr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vz_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    from vz_hip import binding
    assert declared == set(binding.SYMBOLS), declared ^ set(binding.SYMBOLS)
    handle = binding.load_library(lib)          # raises AttributeError on a missing export
    assert handle.vz_abi_version() == binding.ABI_VERSION == 10
    assert handle.vz_target_arch() == b"gfx950"
    assert handle.vz_last_error() == b""


def test_ops_refuse_cpu_tensors():
    import torch
    from vz_hip import binding
    _ensure_built()
    with pytest.raises(ValueError):
        binding.linear(torch.zeros(2, 64).bfloat16(), torch.zeros(4, 64).bfloat16())


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vz_hip.engine import Engine
    from vz_hip.synth import ArchConfig
    with pytest.raises(RuntimeError):
        Engine(ArchConfig(n_layers=1))


def test_missing_library_is_an_error(tmp_path):
    from vz_hip import binding
    saved = binding._lib
    binding._lib = None
    try:
        with pytest.raises(RuntimeError):
            binding.load_library(str(tmp_path / "nope.so"))
    finally:
        binding._lib = saved


def test_graft_entry_abi_assertion_matches_library():
    """__graft_entry__.build() asserts the ABI version: keep it in step with the library (the driver's build check runs it)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "__graft_entry__.py")).read()
    assert "vz_abi_version() == binding.ABI_VERSION" in src
    from vz_hip import binding
    want = binding.ABI_VERSION
    eng = open(os.path.join(root, "vision-zephyr_amd", "csrc", "engine.hip")).read()
    have = int(re.search(r"vz_abi_version\(void\) \{ return (\d+); \}", eng).group(1))
    assert want == have
