#!/usr/bin/env python3
"""Debug helper: run one small case on the GPU and in the oracle, report the first
divergence in visit order (per iteration: comp, betas, acum, hyper-parameters)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import gmrm_amd
from oracle import orc
from tests import cases

name = sys.argv[1] if len(sys.argv) > 1 else "small"
if name == "tiny":
    case = cases.Case("tiny", 403, 13, 3, 4, 1, 0.1, 9, 5, 6, 3)
    inp = cases.make_inputs(case)
    inp["group_index"] = np.array([0, 0, 2, 0, 2, 0, 0, 2, 0, 0, 2, 0, 0], dtype=np.int32)
else:
    case = cases.CASE_BY_NAME[name]
    inp = cases.make_inputs(case)
eps, mask4, nonas = cases.prepare_traits(inp)[0]
ctx = gmrm_amd.Context(case.N, case.M, T=1)
print("R/W: stride", ctx.mbytes)
ctx.upload_bed(inp["bed"]); ctx.upload_trait(0, eps, mask4, nonas)
ctx.compute_markers_statistics(0)
smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"])
ch = orc.Chain(case.N, inp["bed"], eps, mask4, nonas, inp["group_index"], inp["cva"], case.seed, canon=True)
for it in range(1, case.iters + 1):
    smp.iterate(it); ch.iterate(it)
    hy = smp.hyper(0)
    order = ch.midx
    gc, oc = ctx.comp(0), ch.comp
    gb, ob = ctx.betas(0), ch.betas
    ga, oa = ctx.acum(0), ch.acum
    print(f"it {it}: sigmae gpu {hy.sigmae!r} orc {ch.sigmae!r}  mu {hy.mu!r} {ch.mu!r} nupd {hy.n_updates} batches {hy.n_batches}")
    print("   sigmag", hy.sigmag, ch.sigmag)
    bad = [k for k, m in enumerate(order) if gc[m] != oc[m] or gb[m] != ob[m] or ga[m] != oa[m]]
    if bad:
        k = bad[0]; m = order[k]
        print(f"   first divergence at visit {k} marker {m}: comp {gc[m]} vs {oc[m]}  beta {gb[m]!r} vs {ob[m]!r}  acum {ga[m]!r} vs {oa[m]!r}")
        print("   n divergent:", len(bad), "of", case.M, " next few:", bad[:10])
        ge, oe = ctx.get_epsilon(0), ch.eps
        print("   eps max abs diff", np.abs(ge - oe).max())
        break
    else:
        ge, oe = ctx.get_epsilon(0), ch.eps
        print("   all equal; eps equal:", np.array_equal(ge, oe))
