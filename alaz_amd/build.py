"""Build the HIP library in-tree: alaz_amd/lib/libservicegraph.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as well as on the
GPU box.  The built .so is git-ignored but travels with the working tree."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "servicegraph.hip")
DEPS = [SRC] + sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".h", ".hpp"))) + \
       [os.path.join(ROOT, "include", "servicegraph.h")]
LIB = os.path.join(HERE, "lib", "libservicegraph.so")
# the development build of the same sources (-DSG_DEV_KNOBS: SG_* tuning knobs read from the environment, SG_ABLATE bits and phase
# stamps compiled into the kernels) — what tools/ and the A/B tests of alternative kernel paths load; the shipped library has none of it
LIB_DEV = os.path.join(HERE, "lib", "libservicegraph_dev.so")
HOST_SRC = os.path.join(HERE, "csrc", "host")
HOST_LIB = os.path.join(HERE, "lib", "libsgdatastore.so")


def hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the ServiceGraph engine cannot be built (there is no CPU fallback)")


def _stale(target: str, deps) -> bool:
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps if os.path.exists(d))


def build_engine(force: bool = False, dev: bool = False) -> str:
    lib = LIB_DEV if dev else LIB
    if force or _stale(lib, DEPS):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wl,-Bsymbolic",
               "-Wno-unused-function", "-Wno-unused-value"] + (["-DSG_DEV_KNOBS"] if dev else []) + ["-o", lib, SRC]
        subprocess.check_call(cmd)
    return lib


def build_host(force: bool = False) -> str | None:
    """C++ host side (datastore mirror + L7 packer); plain g++, no HIP."""
    if not os.path.isdir(HOST_SRC):
        return None
    srcs = sorted(os.path.join(HOST_SRC, f) for f in os.listdir(HOST_SRC) if f.endswith(".cpp"))
    deps = srcs + [os.path.join(HOST_SRC, f) for f in os.listdir(HOST_SRC) if f.endswith(".hpp")] + [DEPS[-1]]
    if srcs and (force or _stale(HOST_LIB, deps)):
        os.makedirs(os.path.dirname(HOST_LIB), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-pthread",
                               "-I", os.path.join(ROOT, "include"), "-o", HOST_LIB] + srcs + ["-ldl", "-lz"])
    return HOST_LIB if srcs else None


def build_all(force: bool = False):
    """both engine builds side by side (two hipcc runs of the same translation unit) and the host library"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as ex:
        a, b = ex.submit(build_engine, force, False), ex.submit(build_engine, force, True)
        return a.result(), build_host(force), b.result()


if __name__ == "__main__":
    print(build_all(force=True))
