"""Does the sweep-synchronous multi-shard schedule sample the same POSTERIOR as the sequential chain?
("nothing checks the statistics ... against a 1-shard chain on data that carries signal".)

With n shards every shard sweeps its marker block against a residual replica that is reconciled once per sweep
(DESIGN.md section 6): an approximation of the sequential scan -- correlated markers in different shards see each
other's updates one sweep late -- and a different Markov chain from both the 1-shard run and the reference's
n-rank run.  What CAN be required of it: on a phenotype with real signal the quantities a user reads off the
chain (sigmaG, sigmaE, h2, posterior-mean effects, inclusion of the causal markers) agree with the 1-shard chain
within Monte-Carlo error.  Independent genotype columns here (the synthetic recipe of example/data_sim.R); with
strong LD across shard boundaries the approximation is looser -- said so in the CLI's warning."""
import ctypes as C

import numpy as np
import pytest

import gmrm_amd
from gmrm_amd import _lib
from gmrm_amd._lib import check
from tests import cases

pytestmark = pytest.mark.gpu


def _run(n_shards, case, inp, sweeps, burn):
    lib = gmrm_amd.load_library()
    traits = cases.prepare_traits(inp)
    eps, mask4, nonas = traits[0]
    ctxs, smps = [], []
    for r in range(n_shards):
        S, Ml, _ = gmrm_amd.block_of_markers(case.M, n_shards, r)
        ctx = gmrm_amd.Context(case.N, Ml, Mt=case.M, S=S, T=1)
        ctx.upload_bed(inp["bed"][S:S + Ml])
        ctx.upload_trait(0, eps, mask4, nonas)
        smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"], rank=r, nranks=n_shards)
        ctxs.append(ctx); smps.append(smp)
    grp = C.c_void_p()
    if n_shards > 1:
        ca = (C.c_void_p * n_shards)(*[c.h for c in ctxs])
        sa = (C.c_void_p * n_shards)(*[s.h for s in smps])
        check(lib.gmrm_group_create(C.byref(grp), n_shards, ca, sa, inp["cva"].shape[0], inp["cva"].shape[1], 0))
    sg, se, bsum, incl = [], [], np.zeros(case.M), np.zeros(case.M)
    for it in range(1, sweeps + 1):
        if n_shards > 1:
            check(lib.gmrm_group_iterate(grp, it))
        else:
            smps[0].iterate(it)
        hy = smps[0].hyper(0)
        sg.append(float(np.sum(hy.sigmag))); se.append(float(hy.sigmae))
        if it > burn:
            b = np.concatenate([c.betas(0) for c in ctxs])
            bsum += b
            incl += b != 0.0
    if n_shards > 1:
        check(lib.gmrm_group_destroy(grp))
    for s_ in smps:
        s_.close()
    for c in ctxs:
        c.close()
    n = sweeps - burn
    return np.array(sg[burn:]), np.array(se[burn:]), bsum / n, incl / n


def _mc_se(x, nb=10):
    """Monte-Carlo standard error of the mean of a correlated series: batch means."""
    k = len(x) // nb
    means = np.array([x[i * k:(i + 1) * k].mean() for i in range(nb)])
    return means.std(ddof=1) / np.sqrt(nb)


def test_sharded_schedule_samples_the_same_posterior_within_mc_error(gpu):
    case = cases.Case("stat", 4000, 2000, 1, 4, 1, 0.0, 0, 31337, 0, 25)    # 25 causal markers, h2 = 0.5
    inp = cases.make_inputs(case)
    inp["cva"] = np.array([[0.0, 0.0001, 0.001, 0.01]])
    sweeps, burn = 700, 200
    ref = _run(1, case, inp, sweeps, burn)
    h2_ref = ref[0] / (ref[0] + ref[1])
    assert 0.3 < h2_ref.mean() < 0.7, "the 1-shard chain does not find the simulated signal (h2 = 0.5)"
    top = np.argsort(-np.abs(ref[2]))[:25]                                # the markers the sequential chain is surest of
    for n in (2, 4):
        got = _run(n, case, inp, sweeps, burn)
        h2 = got[0] / (got[0] + got[1])
        for name, a, b in (("sigmaG", ref[0], got[0]), ("sigmaE", ref[1], got[1]), ("h2", h2_ref, h2)):
            tol = 4.0 * np.hypot(_mc_se(a), _mc_se(b)) + 0.02 * abs(a.mean())
            assert abs(a.mean() - b.mean()) < tol, f"{n} shards: posterior mean of {name} {b.mean():.4f} vs {a.mean():.4f} (1 shard), tol {tol:.4f}"
        # posterior-mean effects: same picture of the genome
        r = np.corrcoef(ref[2], got[2])[0, 1]
        assert r > 0.97, f"{n} shards: posterior-mean effects correlate only {r:.3f} with the 1-shard chain"
        assert np.max(np.abs(ref[3][top] - got[3][top])) < 0.15, f"{n} shards: inclusion probabilities of the top markers moved"
