#!/usr/bin/env python3
"""Static check of the hand-issued global loads (gload16_issue / gload8_issue in sg_kernels.h).

A load issued through inline asm leaves its destination registers "in flight" until the matching
inline-asm s_waitcnt; the compiler believes they are valid immediately.  If it inserts a register copy
(phi resolution at a branch merge, a loop back-edge, spilling) between the issue and the wait, the copy
reads stale data — results silently go wrong (this happened once: r02c).  This script compiles the engine
to gfx950 assembly and, for every kernel, flags any instruction that READS a VGPR that is the destination
of an asm-issued load before an asm-issued s_waitcnt vmcnt(...) has covered it.

Conservative model: every asm wait `vmcnt(N)` retires all asm loads except the N youngest (loads return in
order).  Compiler-generated waits are ignored (they can only retire more).  Exit code 1 on a finding."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "alaz_amd", "csrc", "servicegraph.hip")


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(asm_text):
    findings, kernel = [], None
    inflight = []                      # list of (set(regs), line) in issue order
    lines = asm_text.split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel, inflight = m.group(1), []
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            j = i + 1                    # an asm statement may hold several instructions (issues + their wait)
            while j < len(lines) and not lines[j].strip().startswith(";;#ASMEND"):
                body = lines[j].strip()
                if body.startswith(("global_load", "buffer_load", "global_atomic")):
                    ops = [o.strip() for o in body.split(None, 1)[1].split(",")]
                    # the address operand must not be an in-flight register either
                    for o in ops[1:]:
                        bad = regs(o.split()[0]) & set().union(*[r for r, _ in inflight]) if inflight else set()
                        if bad:
                            findings.append((kernel, j + 1, body, sorted(bad)))
                    inflight.append((regs(ops[0]), j + 1))
                elif body.startswith("s_waitcnt"):
                    m2 = re.search(r"vmcnt\((\d+)\)", body)
                    keep = int(m2.group(1)) if m2 else 0
                    inflight = inflight[len(inflight) - keep:] if keep else []
                j += 1
            i = j + 1
            continue
        if inflight and t and not t.startswith((";", ".")) and re.match(r"^[a-z]", t):
            parts = t.split(None, 1)
            if len(parts) == 2:
                ops = [o.strip() for o in parts[1].split(",")]
                srcs = ops[1:] if not parts[0].startswith(("global_store", "ds_write", "ds_store", "buffer_store", "scratch_store")) else ops
                live = set().union(*[r for r, _ in inflight])
                for o in srcs:
                    o = o.split()[0] if o else o
                    bad = regs(o) & live
                    if bad:
                        findings.append((kernel, i + 1, t, sorted(bad)))
                # a compiler write to an in-flight register is just as wrong
                if not parts[0].startswith(("global_store", "ds_write", "ds_store", "s_", "buffer_store", "scratch_store")):
                    bad = regs(ops[0].split()[0]) & live
                    if bad:
                        findings.append((kernel, i + 1, t + "   ; WRITES in-flight", sorted(bad)))
        if t.startswith("s_endpgm"):
            inflight = []
        i += 1
    return findings


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "sg.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-Wno-unused-value", "-Wno-unused-command-line-argument", "-o", out, SRC], stderr=subprocess.DEVNULL)
        f = check(open(out).read())
    for k, ln, ins, bad in f:
        print(f"{k}: line {ln}: `{ins}` touches in-flight v{bad}")
    print("asm-issued loads:", "OK" if not f else f"{len(f)} finding(s)")
    return 1 if f else 0


if __name__ == "__main__":
    sys.exit(main())
