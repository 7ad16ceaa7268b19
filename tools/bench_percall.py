#!/usr/bin/env python3
"""Timing of the per-call entries (gmrm_dot / gmrm_update_eps / gmrm_offset_eps / gmrm_sumsqr -- the members the
reference's process() calls per marker; `--sync-every 1` launches them once per marker) at full width:
wall time per call through the C ABI (launch + stream synchronize + the scalar copy), the kernels themselves under
rocprofv3 --kernel-trace --stats (tools/prof_percall.sh)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import gmrm_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--individuals", type=int, default=500_000)
    ap.add_argument("--markers", type=int, default=4096)
    ap.add_argument("--calls", type=int, default=2000)
    a = ap.parse_args()
    N, M = a.individuals, a.markers
    ctx = gmrm_amd.Context(N, M, T=1)
    ctx.synth_bed(171014, 0.4, 0.0)
    rng = np.random.default_rng(1)
    eps, mask4, nonas = gmrm_amd.prepare_phenotype(rng.normal(size=N), (rng.random(N) < 0.02).astype(np.uint8))
    ctx.upload_trait(0, eps, mask4, nonas)
    mave, msig = ctx.compute_markers_statistics(0)
    out = {"N": N, "M": M, "calls": a.calls, "column_bytes": ctx.mbytes, "residual_bytes": 8 * N}
    order = rng.integers(0, M, size=a.calls)

    def timed(fn):
        fn(0)
        t0 = time.perf_counter()
        for i in range(a.calls):
            fn(i)
        return (time.perf_counter() - t0) / a.calls * 1e6

    out["dot_us_per_call"] = timed(lambda i: ctx.dot_product(int(order[i]), float(mave[order[i]]), float(msig[order[i]])))
    out["update_us_per_call"] = timed(lambda i: ctx.update_epsilon([1e-4 if i % 2 else -1e-4, float(mave[order[i]]), float(msig[order[i]])], int(order[i])))
    out["offset_us_per_call"] = timed(lambda i: ctx.offset_epsilon(1e-6 if i % 2 else -1e-6))
    out["sumsqr_us_per_call"] = timed(lambda i: ctx.epsilon_sumsqr())
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
