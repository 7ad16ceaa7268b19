#!/usr/bin/env python3
"""Full-size timing of the two Bayes::predict kernels (k_predict_g: g = Z beta, k_assoc: per-marker
xtx / xty) on synthetic genotypes; both stream the whole .bed block once (algorithmic bytes =
M * ceil(N/4))."""
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
    ap.add_argument("--markers", type=int, default=1_000_000)
    ap.add_argument("--nonzero", type=float, default=1.0, help="fraction of non-zero mean effects")
    ap.add_argument("--missing", type=float, default=0.0, help="rate of missing genotypes (the block then takes the indicator variant of k_pg_mfma)")
    a = ap.parse_args()
    N, M = a.individuals, a.markers
    ctx = gmrm_amd.Context(N, M, T=1)
    ctx.synth_bed(171014, 0.4, a.missing)
    rng = np.random.default_rng(1)
    eps, mask4, nonas = gmrm_amd.prepare_phenotype(rng.normal(size=N), np.zeros(N, dtype=np.uint8))
    ctx.upload_trait(0, eps, mask4, nonas)
    ctx.compute_markers_statistics(0)
    beta = rng.normal(0.0, 1e-3, size=M)
    beta[rng.random(M) >= a.nonzero] = 0.0
    out = {"N": N, "M": M, "bytes": M * ctx.mbytes, "nonzero": a.nonzero, "missing": a.missing}
    for name, fn in (("predict_g", lambda: ctx.predict_g(0, beta)), ("assoc", lambda: ctx.assoc(0, eps[:N]))):
        fn()
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        out[name] = {"seconds_incl_copies": dt, "GBps": M * ctx.mbytes / dt / 1e9}
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
