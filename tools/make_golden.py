#!/usr/bin/env python3
"""Regenerate tests/golden/ (run in the build container, where /root/reference exists).

  ref_luts.bin         reference lookup tables, dumped by oracle/_ref/ref_lut_dump
                       (reference headers compiled in place).
  ref_xfiles.spec.txt  inputs fed to the reference's own writers ...
  ref_xfiles.{csv,bet,cpn}   ... and the records oracle/_ref/ref_xfiles produced
                       (reference src/xfiles.cpp + utilities.cpp compiled in place).
  chain_*.npz          seeded small-case inputs + the oracle's outputs (both summation
                       modes), so the GPU box can check the oracle it rebuilt and the HIP
                       path against committed numbers.
  shards_k3_3ranks.npz case k3 on 3 ranks under the oracle's two multi-rank schedules (one exchange per sweep;
                       the reference's exchange per marker step).
  stl_shuffle.txt.gz   permutations of libstdc++'s std::random_shuffle driven by std::mt19937 (the library code the
                       reference's boost::range::random_shuffle ends in) for n in {2, 3, 777, 20000} x 3 seeds.
Fixtures are data only: no reference source text is stored.
"""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"


def ref_parts():
    subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "ref"], check=True)
    subprocess.run([str(ROOT / "oracle/_ref/ref_lut_dump"), str(GOLD / "ref_luts.bin")], check=True)
    rng = np.random.default_rng(20240611)
    n_it, G, K, Mtot = 3, 2, 4, 37
    lines = [f"{n_it} {G} {K} {Mtot}"]
    for n in range(n_it):
        it = (n + 1) * 5            # thin rate 5
        sg = rng.uniform(0.0, 0.01, G)
        se = rng.uniform(0.3, 0.9)
        m0 = int(rng.integers(0, 1000000))
        pi = rng.dirichlet(np.ones(K), G).ravel()
        betas = rng.normal(0, 0.01, Mtot) * (rng.random(Mtot) < 0.3)
        comp = rng.integers(0, K, Mtot)
        vals = [str(it)] + [repr(float(v)) for v in sg] + [repr(float(se)), str(m0)] \
            + [repr(float(v)) for v in pi] + [repr(float(v)) for v in betas] + [str(int(v)) for v in comp]
        lines.append(" ".join(vals))
    spec = GOLD / "ref_xfiles.spec.txt"
    spec.write_text("\n".join(lines) + "\n")
    for ext in ("csv", "bet", "cpn"):
        p = GOLD / f"ref_xfiles.{ext}"
        if p.exists():
            p.unlink()
    env = dict(os.environ)
    subprocess.run([str(ROOT / "oracle/_ref/ref_xfiles"), str(spec), str(GOLD / "ref_xfiles")], check=True, env=env)
    for ext in ("csv", "bet", "cpn"):
        print(ext, (GOLD / f"ref_xfiles.{ext}").stat().st_size, "bytes")


def stl_parts():
    """tests/golden/stl_shuffle.txt.gz: permutations from libstdc++'s std::random_shuffle + std::mt19937 (oracle/ref_harness/stl_shuffle.cpp)."""
    subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "stl"], check=True)
    args = []
    for n in (2, 3, 777, 20000):
        for seed in (0, 171014, 4294967295):
            args += [str(n), str(seed)]
    out = subprocess.run([str(ROOT / "oracle/_build/stl_shuffle"), *args], check=True, capture_output=True, text=True).stdout
    import gzip
    with gzip.GzipFile(GOLD / "stl_shuffle.txt.gz", "wb", mtime=0) as f:
        f.write(out.encode())
    print("stl_shuffle.txt.gz", len(out), "bytes of text")


def chain_parts():
    from tests import cases
    cases.write_golden(GOLD)
    cases.write_golden_shards(GOLD)


if __name__ == "__main__":
    GOLD.mkdir(parents=True, exist_ok=True)
    what = sys.argv[1:] or ["ref", "chains", "stl"]
    if "ref" in what:
        ref_parts()
    if "chains" in what:
        chain_parts()
    if "stl" in what:
        stl_parts()
