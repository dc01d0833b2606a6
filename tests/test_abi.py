"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/servicegraph.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from alaz_amd import build, engine, replay

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, "include", "servicegraph.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", h)))


def test_header_and_binding_declare_the_same_symbols():
    assert _declared() == sorted(engine.EXPORTS)


def test_library_builds_loads_and_exports_every_symbol(engine_lib):
    for name in _declared():
        assert hasattr(engine_lib, name), name
    assert engine_lib.sg_abi_version() == engine.ABI_VERSION == 6
    from alaz_amd import weights
    assert engine_lib.sg_weights_count(1) == weights.weights_count(1) == 12993
    assert engine_lib.sg_weights_count(2) == weights.weights_count(2)
    assert engine_lib.sg_hash32(12345) == int(replay.hash32(np.array([12345], dtype=np.uint32))[0])


def test_struct_layouts_match_the_header():
    assert C.sizeof(engine.SgConfig) == 88 and C.sizeof(engine.SgStats) == 184 and C.sizeof(engine.SgGeometry) == 52
    assert engine.SgConfig.struct_size.offset == 0 and engine.SgConfig.abi_version.offset == 4 and engine.SgConfig.max_edges.offset == 32
    assert replay.EVENT_DTYPE.itemsize == 32 and replay.EDGE_OUT_DTYPE.itemsize == 64
    assert replay.EVENT_DTYPE.fields["duration_ns"][1] == 16 and replay.EVENT_DTYPE.fields["status"][1] == 12
    assert replay.EDGE_OUT_DTYPE.fields["from_ref"][1] == 24 and replay.EDGE_OUT_DTYPE.fields["score"][1] == 40


def test_code_object_is_gfx950_only():
    lib = build.build_engine()
    blob = open(lib, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful without a GPU")
def test_no_cpu_fallback_create_fails_loudly(engine_lib):
    with pytest.raises(engine.ServiceGraphError) as ei:
        engine.ServiceGraph(max_known_nodes=16, max_edges=16)
    assert ei.value.rc == engine.SG_ENODEV


def test_bad_config_rejected(engine_lib):
    cfg = engine.make_config(max_known_nodes=16, max_edges=16)
    h = C.c_void_p()
    cfg.abi_version = 999
    assert engine_lib.sg_create(C.byref(cfg), C.byref(h)) == engine.SG_EINVAL
    cfg.abi_version = engine.ABI_VERSION; cfg.struct_size = 80          # an ABI-2 caller's sg_config: shorter than ABI 3's first layout
    assert engine_lib.sg_create(C.byref(cfg), C.byref(h)) == engine.SG_EINVAL
    cfg.struct_size = 0
    assert engine_lib.sg_create(C.byref(cfg), C.byref(h)) == engine.SG_EINVAL
    assert engine_lib.sg_create(None, C.byref(h)) == engine.SG_EINVAL
    assert engine_lib.sg_destroy(None) == engine.SG_EINVAL


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under alaz_amd/ or include/ may reference it."""
    bad = []
    for base in ("alaz_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"pyoracle|sg_oracle|libsgoracle|from oracle|import oracle", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_hand_issued_loads_are_never_touched_before_their_wait():
    """tools/check_asm_loads.py on the gfx950 assembly of the engine: no instruction reads or writes a VGPR that
    is the destination of an inline-asm global load before an inline-asm s_waitcnt has covered it (a compiler
    copy there reads stale data; it happened once and only showed up as wrong joins on the GPU)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_loads.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
