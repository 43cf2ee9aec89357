"""CPU tier: the gfx950 library builds, loads, and exports every symbol include/digiham_amd.h declares.
No compute call is made here (no GPU in this tier)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "digiham_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dh_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    import __graft_entry__ as g
    return g.build_hip()


def test_header_and_binding_agree():
    from digiham_amd import _capi
    assert declared_symbols() == sorted(_capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.dh_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.dh_version()


def test_code_object_is_gfx950_only(lib_path):
    data = open(lib_path, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"gfx1100"):
        assert other not in data


def test_exact_kernels_have_no_fma_contraction():
    """-ffp-contract=off is part of the numerical contract: the build flags must carry it."""
    import __graft_entry__ as g
    assert "-ffp-contract=off" in g.HIP_FLAGS and "-fhip-fp32-correctly-rounded-divide-sqrt" in g.HIP_FLAGS


def test_package_has_no_cpu_fallback(monkeypatch, tmp_path):
    """If the gfx950 library is missing the package must fail loudly, not compute on the CPU."""
    from digiham_amd import _capi
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_capi, "_LIB", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _capi.load()


def test_engine_config_validation_in_emulation(emu_ctx):
    """Host-side argument checking (same code in both builds): bad configs are rejected with DH_EINVAL."""
    from digiham_amd import api
    from digiham_amd._capi import DhError
    for kw in (dict(sps=2), dict(sps=41), dict(rrc="none", demod="none", proto="none")):
        with pytest.raises(DhError):
            api.Engine(1, 100, ctx=emu_ctx, **kw)
    eng = api.Engine(2, 100, ctx=emu_ctx)
    import numpy as np
    with pytest.raises(DhError):
        eng.push(np.zeros((2, 101), np.float32))          # n > max_samples


@pytest.mark.gpu
def test_engine_refuses_a_device_without_unaligned_lds_reads(gpu_ctx, monkeypatch):
    """The slicer's window phases read 8 / 16 bytes from four-byte aligned LDS addresses (dsp_core.hpp, P3); the backend probes once
    per device that such reads come back whole (engine.hip: k_lds_unaligned_probe).  A failed probe -- forced here -- must refuse the
    engine with an error that says why, not run kernels that would slice wrong symbols after every timing step."""
    from digiham_amd import api
    api.Engine(2, 4096, proto="dmr", ctx=gpu_ctx).close()              # the real probe passes on an MI355X
    monkeypatch.setenv("DH_LDS_PROBE_FORCE_FAIL", "1")
    with pytest.raises(Exception) as e:
        api.Engine(2, 4096, proto="dmr", ctx=gpu_ctx)
    assert "LDS" in str(e.value)
