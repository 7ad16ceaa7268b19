"""Build libgmrm_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract
(gm_common.h): an a*b+c written as two operations must stay two roundings on the device.
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libgmrm_hip.so"
SOURCES = ["ops.hip", "sweep.hip", "capi.cpp", "sampler.cpp", "ingest.cpp", "shard_group.cpp"]
HEADERS = ["gm_common.h", "gm_rng.h", "gm_internal.h", "gm_host.h", "zig_tables.h", "../../include/gmrm_hip.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall",
         "-Wno-unused-function"]
# host-only translation units: plain C++ against the HIP runtime API (no device pass)
HOST_FLAGS = ["-O2", "-pthread", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-fPIC", "-std=c++17", "-Wall",
              "-Wno-unused-function", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: libgmrm_hip.so cannot be built")
    return exe


def stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [CSRC / h for h in HEADERS] + [Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False, prof: bool = False) -> Path:
    """prof=True builds the diagnostic variant libgmrm_hip_prof.so (-DGM_SWEEP_PROF: in-kernel
    phase stamps); select it with GMRM_HIP_LIB=.../libgmrm_hip_prof.so GMRM_SWEEP_PROF=1."""
    global LIB, FLAGS
    if prof:
        lib_prof = HERE / "libgmrm_hip_prof.so"
        saved = (LIB, FLAGS)
        LIB, FLAGS = lib_prof, FLAGS + ["-DGM_SWEEP_PROF"] + (["-DGM_PROF_TID=" + os.environ["GM_PROF_TID"]] if os.environ.get("GM_PROF_TID") else [])
        try:
            return _build(True, verbose, "_build_prof")
        finally:
            LIB, FLAGS = saved
    if not force and not stale():
        return LIB
    return _build(force, verbose, "_build")


def _build(force: bool, verbose: bool, objname: str) -> Path:
    objdir = HERE / objname
    objdir.mkdir(exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        o = objdir / (s.replace(".", "_") + ".o")
        if s.endswith(".hip"):
            cmd = [hipcc(), *FLAGS, "-c", str(CSRC / s), "-o", str(o)]
        else:
            cmd = [shutil.which("g++") or "g++", *HOST_FLAGS, "-c", str(CSRC / s), "-o", str(o)]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(str(o))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", str(LIB), *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, prof="--prof" in sys.argv))
