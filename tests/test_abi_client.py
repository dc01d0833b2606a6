"""A plain C11 program (tests/micro/abi_client.c, gcc -std=c11 -Wall -Werror) against include/servicegraph.h and
libservicegraph.so — what a cgo / FFI binding of the reference would be compiled against.  On CPU it must build, link and be
told SG_ENODEV by sg_create (exit code 77: no CPU fallback); on the GPU it drives create -> upsert -> ingest -> flush ->
destroy and checks the rows itself."""
import os
import subprocess

import pytest

from alaz_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    lib = build.build_engine()
    exe = tmp_path / "abi_client"
    libdir = os.path.dirname(lib)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "micro", "abi_client.c"),
                           "-L", libdir, "-lservicegraph", "-L/opt/rocm/lib", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    return str(exe)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful without a GPU")
def test_c11_client_builds_links_and_is_refused_without_a_gpu(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77, r.stdout + r.stderr


@pytest.mark.gpu
def test_c11_client_drives_the_engine_through_the_c_abi(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "abi_client ok" in r.stdout, r.stdout + r.stderr
