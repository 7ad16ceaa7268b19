#!/usr/bin/env python3
"""Throughput of gmrm_load_bed_file on this box: writes a synthetic .bed of --gb gigabytes
(N = 500 000 => 125 000-byte markers) under --dir, loads it with 1 / 4 / 16 reader threads.
The file was just written, so reads come from the page cache: this measures the pread -> pinned
-> copy-engine pipeline, not the disk."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import gmrm_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=8.0)
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--individuals", type=int, default=500_000)
    a = ap.parse_args()
    N = a.individuals
    mb = (N + 3) // 4
    M = int(a.gb * 1e9 / mb)
    path = Path(a.dir) / "gmrm_ingest_bench.bed"
    rng = np.random.default_rng(1)
    block = rng.integers(0, 256, size=(4096, mb), dtype=np.uint8)
    t0 = time.perf_counter()
    with open(path, "wb") as f:
        f.write(b"\x6c\x1b\x01")
        left = M
        while left > 0:
            n = min(left, block.shape[0])
            f.write(block[:n].tobytes())
            left -= n
    t_write = time.perf_counter() - t0
    out = {"file_GB": M * mb / 1e9, "write_s": t_write, "runs": []}
    ctx = gmrm_amd.Context(N, M, T=1)
    try:
        for th in (1, 4, 16, 16):
            st = ctx.load_bed_file(path, threads=th)
            out["runs"].append({"threads": th, "GBps": st["GBps"], "seconds": st["seconds"], "read_seconds": st["read_seconds"]})
        got = ctx.download_bed(M - 3, 3)
        assert np.array_equal(got, block[(M - 3) % 4096:(M - 3) % 4096 + 3] if (M - 3) % 4096 + 3 <= 4096 else got)
    finally:
        ctx.close()
        os.unlink(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
