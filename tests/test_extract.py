"""sparse extraction of non-zero effects from a .bet history.  Upstream ships the tool as a
binary only (example/extract_non_zero_betaAll); the fixtures are ITS output on a small seeded history
(tools/make_golden_extract.py), so this pins format and record numbering to the reference itself."""
import struct
import subprocess
from pathlib import Path

import numpy as np

from gmrm_amd import io

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
BIN = ROOT / "bin" / "extract_non_zero_betaAll"


def _run(*args):
    return subprocess.run([str(BIN), *map(str, args)], capture_output=True, timeout=60)


def test_extractor_matches_the_reference_binary_output():
    assert BIN.exists(), "bin/extract_non_zero_betaAll not built (python __graft_entry__.py)"
    for lo, hi in ((0, 5), (2, 3), (4, 4)):
        r = _run(GOLD / "ref_extract.bet", lo, hi)
        assert r.returncode == 0
        want = (GOLD / f"ref_extract_{lo}_{hi}.txt").read_bytes()
        assert r.stdout == want
        # the Python mirror used by the other tests prints the same records
        txt = "".join("%7d %7d %20.12f\n" % rec for rec in io.extract_non_zero(GOLD / "ref_extract.bet", lo, hi))
        assert txt.encode() == want


def test_extractor_usage_and_end_of_file(tmp_path):
    r = _run()
    assert r.returncode != 0 and r.stdout == (GOLD / "ref_extract_usage.txt").read_bytes()
    r = _run(tmp_path / "missing.bet", 0, 1)
    assert r.returncode != 0 and b"Error opening file" in r.stdout
    # records past the end of the file: upstream prints its stale read buffer; this build stops (documented difference)
    M = 5
    blob = struct.pack("<I", M) + struct.pack("<I", 1) + np.array([0, 1.5, 0, 0, -2.25]).tobytes()
    p = tmp_path / "one.bet"
    p.write_bytes(blob)
    r = _run(p, 0, 9)
    assert r.stdout.decode() == "%7d %7d %20.12f\n%7d %7d %20.12f\n" % (0, 1, 1.5, 0, 4, -2.25)


def test_extractor_reads_what_the_history_writer_writes(tmp_path):
    """Round trip through the product's own .bet layout (gmrm_amd.io.read_history reads it back)."""
    M, its = 11, [2, 4, 6]
    rng = np.random.default_rng(3)
    betas = [np.where(rng.random(M) < 0.3, rng.normal(size=M), 0.0) for _ in its]
    p = tmp_path / "h.bet"
    p.write_bytes(struct.pack("<I", M) + b"".join(struct.pack("<I", it) + b.tobytes() for it, b in zip(its, betas)))
    h_m, h_it, h_val = io.read_history(p, np.float64)
    assert h_m == M and list(h_it) == its and all(np.array_equal(a, b) for a, b in zip(h_val, betas))
    got = io.extract_non_zero(p, 0, len(its) - 1)
    want = [(r, int(m), float(b[m])) for r, b in enumerate(betas) for m in np.flatnonzero(b)]
    assert got == want
    assert _run(p, 0, 2).stdout.decode() == "".join("%7d %7d %20.12f\n" % rec for rec in want)
