#!/usr/bin/env python3
"""Pin bin/extract_non_zero_betaAll to the reference's tool.  Upstream ships example/extract_non_zero_betaAll
as a BINARY only; this script (run where /root/reference exists) writes a small seeded .bet history, runs that
binary on it and commits its stdout as fixtures: tests/golden/ref_extract.bet, ref_extract_<min>_<max>.txt.
The .bet layout is src/xfiles.hpp:14-38 (uint32 Mtot | per record: uint32 iteration + Mtot float64)."""
import struct
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
REF = Path("/root/reference/example/extract_non_zero_betaAll")


def main():
    rng = np.random.default_rng(20211)
    M, recs = 37, 6
    blob = [struct.pack("<I", M)]
    for r in range(recs):
        beta = np.zeros(M)
        idx = rng.choice(M, size=rng.integers(0, 9), replace=False)
        beta[idx] = rng.normal(0.0, 0.05, size=len(idx)) * rng.choice([1.0, 1e-6, 30.0], size=len(idx))
        blob.append(struct.pack("<I", 10 * (r + 1)) + beta.tobytes())          # stored iteration numbers 10, 20, ... (thinned run)
    (GOLD / "ref_extract.bet").write_bytes(b"".join(blob))
    for lo, hi in ((0, 5), (2, 3), (4, 4)):
        out = subprocess.run([str(REF), str(GOLD / "ref_extract.bet"), str(lo), str(hi)], capture_output=True, check=True).stdout
        (GOLD / f"ref_extract_{lo}_{hi}.txt").write_bytes(out)
        print(lo, hi, len(out.splitlines()), "lines")
    usage = subprocess.run([str(REF)], capture_output=True).stdout
    (GOLD / "ref_extract_usage.txt").write_bytes(usage)


if __name__ == "__main__":
    main()
