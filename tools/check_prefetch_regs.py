#!/usr/bin/env python3
"""Check an invariant of gm::k_sweep that the compiler does not know about (sweep.hip prefetch).

The genotype prefetch issues `global_load_dwordx4` through inline asm into AGPRs and waits for the data
a whole round later (`s_waitcnt vmcnt(0)` in prefetch_commit).  hipcc does not count asm loads, so if it
ever copied, spilled or reused one of those registers between the load and the wait, it would move a value
that has not landed yet -- silent, timing-dependent corruption of genotype slices.  This script
disassembles every k_sweep instantiation, rebuilds its basic blocks and proves by forward data-flow
that every instruction naming one of those AGPRs (other than the loads themselves) is reached only
through an `s_waitcnt vmcnt(0)` issued after the last such load, on every path.

The same hazard one level down: phase A stages its MFMA operands with `ds_read_b128` issued through inline
asm (stage_read) one super-step ahead of their use and waits with a hand-counted `s_waitcnt lgkmcnt(n)` (stage_wait).
check_stage() replays every basic block: it keeps the in-order queue of LDS operations in flight (LDS returns in
order, so `lgkmcnt(n)` retires all but the last n), marks the destination VGPRs of the asm reads as pending until a
wait retires them, and fails if any instruction names a pending register, if a scalar memory load (out of order
with respect to LDS) is in flight at a counted wait, or if a block ends with such a register pending.

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
        m = re.match(r"^(_ZN2gm7k_sweepILi\d+ELi\d+ELb[01]ELb[01]EEEvNS_9SweepArgsE):", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):           # (a kernel may contain several s_endpgm)
                yield name, cur
                cur = None
            else:
                cur.append(ln)


VRANGE = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def vgprs(text):
    out = set()
    for m in VRANGE.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check_stage(name, lines):
    """The staged LDS reads of phase A (see the module docstring).  Straight-line replay per basic block."""
    problems, in_asm = [], False
    queue = []                 # LDS operations in flight, oldest first: the set of asm-read destination VGPRs (or an empty set)
    smem = 0                   # scalar loads in flight (they share lgkmcnt and return out of order)
    n_reads = n_waits = 0

    def pending():
        out = set()
        for q in queue:
            out |= q
        return out

    for ln in lines:
        raw = ln.strip()
        if raw.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.startswith(";;#ASMEND"):
            in_asm = False
            continue
        s = ln.split(";")[0].strip()
        if not s:
            continue
        if re.match(r"^\.LBB\d+_\d+:", s) or s.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            if pending():
                problems.append(f"{name}: staged LDS read register(s) {sorted(pending())[:4]}.. still pending at a basic-block boundary (`{s}`)")
            queue, smem = [], 0                                   # (compiler-issued operations are the compiler's business)
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        if s.startswith("s_waitcnt"):
            m = LGKM.search(s)
            if m is None and "lgkmcnt" not in s and not re.match(r"^s_waitcnt\s+(0x[0-9a-f]+|\d+)$", s):
                continue                                          # vmcnt / expcnt only
            keep = int(m.group(1)) if m else 0                    # a raw immediate: treated as lgkmcnt(0) only if it says so below
            if m is None:
                imm = int(s.split()[1], 0)
                keep = (imm >> 8) & 0xF
            if pending():
                n_waits += 1
                if smem and keep:
                    problems.append(f"{name}: `{s}` counts LDS reads while {smem} scalar load(s) are in flight (they return out of order)")
            while len(queue) > keep:
                queue.pop(0)
            if keep == 0:
                smem = 0
            continue
        if s.startswith(("s_load_", "s_buffer_load_", "s_scratch_load_")):
            smem += 1
            continue
        if in_asm and s.startswith("ds_read_b128"):
            dst = vgprs(s.split(",")[0])
            touched = vgprs(s.split(",", 1)[1]) & pending()
            if touched:
                problems.append(f"{name}: `{s}` addresses through pending staged register(s) {sorted(touched)}")
            queue.append(dst)
            n_reads += 1
            continue
        touched = vgprs(s) & pending()
        if touched:
            problems.append(f"{name}: `{s}` touches staged LDS read register(s) {sorted(touched)[:4]}.. before the wait that retires them")
        if s.startswith("ds_") or s.startswith("buffer_load") and " lds" in s:
            queue.append(set())                                   # any other LDS operation takes a place in the in-order queue
    if n_reads == 0:
        problems.append(f"{name}: no staged ds_read_b128 found (did phase A change?)")
    if not problems:
        print(f"ok  {name}: {n_reads} staged LDS reads, retired by {n_waits} counted waits before any use")
    return problems


def check_kernel(name, lines):
    """Forward data-flow over the kernel's basic blocks, per register (round 4: the register-home tiles are loaded slot by
    slot while the other slots are in use).  State = the AGPRs named by an asm `global_load_dwordx4 a[..]` since the last
    `s_waitcnt` with vmcnt(0) on ANY path (the hardware counter covers asm loads too).  No other instruction may name a
    register in that set -- not a read, not a copy, not a spill, not a second load."""
    pf = set()
    for ln in lines:
        s = ln.strip()
        if s.startswith("global_load_dwordx4 a["):
            pf |= agprs(s.split(",")[0])
    if not pf:
        # register-home tiles exist only in the long-batch kernels (the layout without missing genotypes; last template argument) at R = 2 and R = 4
        if re.search(r"k_sweepILi[24]ELi0ELb[01]ELb1E", name):
            return [f"{name}: no AGPR tile loads found (did the kernel change?)"]
        print(f"ok  {name}: no register-home tiles in this kernel (every tile lives in LDS)")
        return []
    # ---- basic blocks ---------------------------------------------------------------------------
    blocks, cur, label_of = [], {"label": None, "ins": []}, {}
    guarded = False
    n_guarded = 0
    for ln in lines:
        raw = ln.strip()
        if raw.startswith(";;#ASMSTART") or raw.startswith(";;#ASMEND"):
            guarded = False
            continue
        if raw.startswith("; tile-window guarded read"):
            guarded = True                                        # sweep.hip, stage_tile_regs_guarded: the one exemption, inside ITS asm block
            continue
        s = ln.split(";")[0].strip()
        if not s:
            continue
        if guarded:
            if s.startswith("ds_write_b128") and agprs(s.split(",", 1)[1]):
                n_guarded += 1
                continue
            cur["ins"].append("s_nop 0 ; BAD-GUARDED " + s)
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
        # state: the AGPRs named by an asm tile load since the last vmcnt(0) on some path (their data may not have landed)
        state = set(state)
        for s in ins:
            if s.startswith("global_load_dwordx4 a["):
                dst = agprs(s.split(",")[0])
                again = dst & state
                if again and report is not None:
                    report.append(f"{name}: `{s}` loads into register(s) {sorted(again)[:4]}.. whose previous load may still be in flight")
                state |= dst
                continue
            if s.startswith("s_waitcnt") and "vmcnt(0)" in s:
                state = set()
                continue
            if "BAD-GUARDED" in s and report is not None:
                report.append(f"{name}: a guarded asm block holds something else than v_accvgpr_read_b32: `{s}`")
            touched = agprs(s) & state
            if touched and report is not None:
                report.append(f"{name}: `{s}` touches tile register(s) {sorted(touched)[:4]}.. while their load may still be in flight")
        return state

    n = len(blocks)
    inn = [set() for _ in range(n)]        # start empty, meet = union (a register is pending if it is on ANY path)
    changed = True
    while changed:
        changed = False
        for i in range(n):
            st = set()
            for j in pred[i]:
                st |= transfer(inn[j], blocks[j]["ins"], None)
            if st != inn[i]:
                inn[i], changed = st, True
    problems = []
    for i in range(n):
        transfer(inn[i], blocks[i]["ins"], problems)
    if not problems:
        reads = sum(1 for b in blocks for s in b["ins"] if s.startswith("v_accvgpr_read_b32") and agprs(s) & pf)
        print(f"ok  {name}: {len(pf)} tile AGPRs in {n} basic blocks; {reads} reads, none of a register whose load may be in flight"
              f" ({n_guarded} reads in asm blocks guarded by the tile window)")
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
        bad += check_stage(name, lines)
    if n == 0:
        bad.append("no gm::k_sweep instantiation found in the assembly")
    for b in bad:
        print("FAIL", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
