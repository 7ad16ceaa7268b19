#!/usr/bin/env python3
"""Resource numbers of the hand-written kernels, read from the built library.

The sweep kernel sits at the edge of the register file: round 3 lost and regained 20 % of sweep time through
register allocation alone when unexecuted code was added.  Nothing in the numerics tests sees that.  This tool reads
what the compiler recorded for every kernel of libgmrm_hip.so -- `.AMDGPU.metadata` of the gfx950 code objects
(vgpr / agpr / sgpr counts, spill counts, scratch and LDS bytes) and the kernels' code sizes from the symbol tables --
and compares the guarded kernels (k_sweep, k_pg_mfma, k_assoc_mfma instantiations) with the committed
profiles/kernel_resources.json:

  * any increase of VGPR spills or scratch bytes,
  * SGPR spills more than 10 % above the committed number,
  * code size more than 15 % above it,
  * a guarded kernel that appears or disappears

fail.  tests/test_kernel_resources.py runs it on the CPU (hipcc cross-compiles; no GPU needed).

  python tools/kernel_resources.py            # compare, exit code 1 on a violation
  python tools/kernel_resources.py --write    # accept the current numbers (commit the json with the change that moved them)
"""
import json
import re
import shutil
import subprocess
import sys
import tempfile

import yaml
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "gmrm_amd" / "libgmrm_hip.so"
BASE = ROOT / "profiles" / "kernel_resources.json"
LLVM = Path("/opt/rocm/lib/llvm/bin")
GUARDED = re.compile(r"k_sweep|k_pg_mfma|k_assoc_mfma")
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count",
          "private_segment_fixed_size", "group_segment_fixed_size")


def demangle(names):
    filt = shutil.which("c++filt") or str(LLVM / "llvm-cxxfilt")
    r = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, r.stdout.splitlines()))


def read_library(lib=LIB):
    """{demangled kernel name: {field: int, 'code_bytes': int}} for every kernel of every gfx950 code object in `lib`."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        so = Path(td) / "lib.so"
        shutil.copy(lib, so)
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(so)], capture_output=True, text=True, check=True)
        cos = sorted(Path(td).glob("lib.so.*gfx950*"))
        if not cos:
            raise RuntimeError("no gfx950 code object found in " + str(lib))
        for co in cos:
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True, check=True).stdout
            syms = subprocess.run([str(LLVM / "llvm-readelf"), "-s", "-W", str(co)], capture_output=True, text=True, check=True).stdout
            size = {}
            for ln in syms.splitlines():
                f = ln.split()
                if len(f) >= 8 and f[3] == "FUNC":
                    size[f[7]] = int(f[2])
            doc = notes[notes.index("---"):]
            if "\n..." in doc:
                doc = doc[:doc.index("\n...")]
            meta = yaml.safe_load(doc)
            for k in meta.get("amdhsa.kernels", []):
                name = k[".name"]
                rec = {f: int(k.get("." + f, 0)) for f in FIELDS}
                rec["code_bytes"] = size.get(name, 0)
                out[name] = rec
    dm = demangle(list(out))
    return {dm[k]: v for k, v in out.items()}


def compare(now, base):
    problems = []
    g_now = {k: v for k, v in now.items() if GUARDED.search(k)}
    g_base = {k: v for k, v in base.items() if GUARDED.search(k)}
    for k in sorted(set(g_now) | set(g_base)):
        if k not in g_base:
            problems.append(f"{k}: not in profiles/kernel_resources.json (new instantiation: run --write and commit)")
            continue
        if k not in g_now:
            problems.append(f"{k}: in profiles/kernel_resources.json but not in the library")
            continue
        a, b = g_now[k], g_base[k]
        if a.get("vgpr_spill_count", 0) > b.get("vgpr_spill_count", 0):
            problems.append(f"{k}: VGPR spills {b.get('vgpr_spill_count', 0)} -> {a['vgpr_spill_count']}")
        if a.get("private_segment_fixed_size", 0) > b.get("private_segment_fixed_size", 0):
            problems.append(f"{k}: scratch bytes {b.get('private_segment_fixed_size', 0)} -> {a['private_segment_fixed_size']}")
        if a.get("sgpr_spill_count", 0) > 1.10 * b.get("sgpr_spill_count", 0) + 8:
            problems.append(f"{k}: SGPR spills {b.get('sgpr_spill_count', 0)} -> {a['sgpr_spill_count']} (> +10 %)")
        if a.get("code_bytes", 0) > 1.15 * b.get("code_bytes", 1):
            problems.append(f"{k}: code bytes {b.get('code_bytes', 0)} -> {a['code_bytes']} (> +15 %)")
        if a.get("group_segment_fixed_size", 0) != b.get("group_segment_fixed_size", 0):
            problems.append(f"{k}: static LDS bytes {b.get('group_segment_fixed_size', 0)} -> {a['group_segment_fixed_size']}")
    return problems


def main(argv):
    now = read_library()
    if "--write" in argv:
        BASE.write_text(json.dumps({k: now[k] for k in sorted(now) if GUARDED.search(k)}, indent=1) + "\n")
        print(f"wrote {BASE} ({sum(1 for k in now if GUARDED.search(k))} guarded kernels of {len(now)})")
        return 0
    if "--print" in argv:
        for k in sorted(now):
            if GUARDED.search(k):
                print(k, now[k])
    base = json.loads(BASE.read_text())
    problems = compare(now, base)
    for p in problems:
        print("RESOURCE  " + p)
    print(f"{sum(1 for k in now if GUARDED.search(k))} guarded kernels checked, {len(problems)} problem(s)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
