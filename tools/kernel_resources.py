#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in the engine library, from the compiler's own remarks
(`hipcc -Rpass-analysis=kernel-resource-usage`, the same flags as alaz_amd/build.py; runs without a GPU).

    python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt

One line per kernel instantiation: VGPRs, AGPRs, SGPRs, scratch bytes per lane, spilled VGPRs / SGPRs, static LDS bytes per
workgroup (dynamic LDS is the launch's: DESIGN.md §3), waves per SIMD the registers allow."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "alaz_amd", "csrc", "servicegraph.hip")
FILT = "c++filt"


def main():
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-everything",
               "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(td, "x.so"), SRC]
        log = subprocess.run(cmd, capture_output=True, text=True, cwd=td).stderr
    rows, cur = [], None
    for ln in log.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?:\[[^\]]+\])?):\s*(\S+)", ln)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}; rows.append(cur)
        elif cur is not None:
            cur[k] = v
    names = subprocess.run([FILT], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage alaz_amd/csrc/servicegraph.hip")
    print(f"# {'VGPR':>4} {'AGPR':>4} {'SGPR':>4} {'scratch':>7} {'vspill':>6} {'sspill':>6} {'LDS':>6} {'occ':>3}  kernel")
    seen = set()
    for r, n in zip(rows, names):
        n = re.sub(r"\(Dev.*$|\(.*$", "", n).replace("void ", "")
        key = (n, r.get("VGPRs"), r.get("ScratchSize [bytes/lane]"))
        if key in seen:
            continue
        seen.add(key)
        print(f"  {r.get('VGPRs', '?'):>4} {r.get('AGPRs', '?'):>4} {r.get('TotalSGPRs', '?'):>4} {r.get('ScratchSize [bytes/lane]', '?'):>7} "
              f"{r.get('VGPRs Spill', '?'):>6} {r.get('SGPRs Spill', '?'):>6} {r.get('LDS Size [bytes/block]', '?'):>6} {r.get('Occupancy [waves/SIMD]', '?'):>3}  {n}")


if __name__ == "__main__":
    sys.exit(main())
