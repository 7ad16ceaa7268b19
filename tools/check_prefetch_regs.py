#!/usr/bin/env python3
"""Check an invariant of gm::k_sweep that the compiler does not know about (sweep.hip prefetch).

The genotype prefetch issues `global_load_dwordx4` through inline asm into AGPRs and waits for the data
a whole round later (`s_waitcnt vmcnt(0)` in prefetch_commit).  hipcc does not count asm loads, so if it
ever copied, spilled or reused one of those registers between the load and the wait, it would move a value
that has not landed yet -- silent, timing-dependent corruption of genotype slices.  This script
disassembles every k_sweep instantiation, rebuilds its basic blocks and proves by forward data-flow
that every instruction naming one of those AGPRs (other than the loads themselves) is reached only
through an `s_waitcnt vmcnt(0)` issued after the last such load, on every path.

  python tools/check_prefetch_regs.py            # compiles gmrm_amd/csrc/sweep.hip to assembly (hipcc -S)
  python tools/check_prefetch_regs.py file.s     # checks an existing assembly file
Exit code 0 = invariant holds for every instantiation.
"""
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
RANGE = re.compile(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b")


def agprs(text):
    out = set()
    for m in RANGE.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernels(asm):
    cur, name = None, None
    for ln in asm.splitlines():
        m = re.match(r"^(_ZN2gm7k_sweepILi\d+ELi\d+ELb[01]EEEvNS_9SweepArgsE):", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):           # (a kernel may contain several s_endpgm)
                yield name, cur
                cur = None
            else:
                cur.append(ln)


def check_kernel(name, lines):
    """Forward data-flow over the kernel's basic blocks.  State = "every asm prefetch load issued so far has
    landed": cleared by an asm `global_load_dwordx4 a[..]`, set by any `s_waitcnt` with vmcnt(0) (the
    hardware counter covers asm loads too).  Every other instruction that names a prefetch AGPR must see
    the state set on EVERY path that reaches it."""
    pf = set()
    for ln in lines:
        s = ln.strip()
        if s.startswith("global_load_dwordx4 a["):
            pf |= agprs(s.split(",")[0])
    if not pf:
        return [f"{name}: no AGPR prefetch loads found (did the kernel change?)"]
    # ---- basic blocks ---------------------------------------------------------------------------
    blocks, cur, label_of = [], {"label": None, "ins": []}, {}
    for ln in lines:
        s = ln.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            if cur["ins"] or cur["label"] is not None:
                blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        cur["ins"].append(s)
        if s.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    if cur["ins"] or cur["label"] is not None:
        blocks.append(cur)
    for i, b in enumerate(blocks):
        if b["label"]:
            label_of[b["label"]] = i
    succ = []
    for i, b in enumerate(blocks):
        out = []
        last = b["ins"][-1] if b["ins"] else ""
        if last.startswith("s_branch"):
            out.append(label_of[last.split()[-1]])
        elif last.startswith("s_cbranch"):
            out.append(label_of[last.split()[-1]])
            if i + 1 < len(blocks):
                out.append(i + 1)
        elif last.startswith(("s_endpgm", "s_setpc")):
            pass
        elif i + 1 < len(blocks):
            out.append(i + 1)
        succ.append(out)
    pred = [[] for _ in blocks]
    for i, out in enumerate(succ):
        for j in out:
            pred[j].append(i)

    def transfer(state, ins, report):
        for s in ins:
            if s.startswith("global_load_dwordx4 a["):
                state = False
                continue
            if s.startswith("s_waitcnt") and "vmcnt(0)" in s:
                state = True
                continue
            touched = agprs(s) & pf
            if touched and not state and report is not None:
                report.append(f"{name}: `{s}` touches prefetch register(s) {sorted(touched)[:4]}.. while a prefetch load may still be in flight")
        return state

    n = len(blocks)
    inn = [True] * n                      # optimistic start, meet = AND
    changed = True
    while changed:
        changed = False
        for i in range(n):
            st = all(transfer(inn[j], blocks[j]["ins"], None) for j in pred[i]) if pred[i] else True
            if st != inn[i]:
                inn[i], changed = st, True
    problems = []
    for i in range(n):
        transfer(inn[i], blocks[i]["ins"], problems)
    if not problems:
        reads = sum(1 for b in blocks for s in b["ins"] if s.startswith("v_accvgpr_read_b32") and agprs(s) & pf)
        print(f"ok  {name}: {len(pf)} prefetch AGPRs in {n} basic blocks; {reads} reads, all behind a vmcnt(0) wait on every path")
    return problems


def main():
    if len(sys.argv) > 1:
        asm = Path(sys.argv[1]).read_text()
    else:
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        with tempfile.TemporaryDirectory() as td:
            out = Path(td) / "sweep.s"
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only",
                   str(ROOT / "gmrm_amd" / "csrc" / "sweep.hip"), "-o", str(out)]
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            asm = out.read_text()
    bad = []
    n = 0
    for name, lines in kernels(asm):
        n += 1
        bad += check_kernel(name, lines)
    if n == 0:
        bad.append("no gm::k_sweep instantiation found in the assembly")
    for b in bad:
        print("FAIL", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
