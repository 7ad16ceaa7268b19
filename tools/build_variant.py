#!/usr/bin/env python3
"""Build a VARIANT of the library for an A/B on one box (tools/ab_bench.sh, ab_mixed.sh, ab_sparse.sh):
sweep.hip taken from a git revision and/or compiled with extra -D flags, linked with the current objects of
everything else, as gmrm_amd/libgmrm_hip_<name>.so (git-ignored; select it with GMRM_HIP_LIB).

  tools/build_variant.py pre HEAD                 # the committed kernel next to a modified working tree
  tools/build_variant.py z0 - -DGM_SPARSE_ZMAX=0  # the working tree's kernel with a knob changed
"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from gmrm_amd import build  # noqa: E402


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    name, rev, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    build.build()                                           # the other objects (gmrm_amd/_build)
    csrc = ROOT / "gmrm_amd" / "csrc"
    src = csrc / "sweep.hip"
    tmp = None
    if rev != "-":
        tmp = csrc / f"_sweep_{name}.hip"                   # next to the headers it includes
        tmp.write_bytes(subprocess.run(["git", "show", f"{rev}:gmrm_amd/csrc/sweep.hip"], cwd=ROOT, check=True,
                                       capture_output=True).stdout)
        src = tmp
    objdir = ROOT / "gmrm_amd" / "_build_var"
    objdir.mkdir(exist_ok=True)
    obj = objdir / f"sweep_{name}.o"
    try:
        subprocess.run([build.hipcc(), *build.FLAGS, *flags, "-c", str(src), "-o", str(obj)], check=True)
    finally:
        if tmp:
            tmp.unlink()
    others = [str(o) for o in sorted((ROOT / "gmrm_amd" / "_build").glob("*.o")) if o.name != "sweep_hip.o"]
    lib = ROOT / "gmrm_amd" / f"libgmrm_hip_{name}.so"
    subprocess.run([build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *others, str(obj)], check=True)
    print(lib)


if __name__ == "__main__":
    main()
