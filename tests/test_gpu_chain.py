"""GPU parity tests of the whole path: the persistent sweep kernel + host sampler against
the oracle chain.  Bar (BASELINE.json north_star): component (inclusion) indices bit-exact
under a fixed seed, effect sizes within 1e-6 relative.  Against the oracle's
order-independent mode the HIP path is held to bit-exact EVERYTHING (betas, residual,
hyper-parameters, .csv records); against its reference-order mode to the north_star bar."""
import numpy as np
import pytest

import gmrm_amd
from tests import cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", [c.name for c in cases.CASES])
def test_chain_matches_oracle_and_golden(gpu, name):
    case = cases.CASE_BY_NAME[name]
    inp, z = cases.load_golden(name)
    got = cases.run_gpu(case, inp)
    want = cases.run_oracle(case, inp, canon=True)
    cases.assert_same_history(got, want, exact=True)
    for t, h in enumerate(got):
        assert np.array_equal(np.array(h["comp"], dtype=np.int8), z[f"t{t}_comp"])
        assert np.array_equal(np.array(h["betas"]), z[f"t{t}_betas"])
        assert b"".join(h["csv"]) == z[f"t{t}_csv"].tobytes()
        assert np.array_equal(h["mave"], z[f"t{t}_mave"]) and np.array_equal(h["msig"], z[f"t{t}_msig"])
        # north_star bar against the reference-order oracle
        assert np.array_equal(np.array(h["comp"], dtype=np.int8), z[f"t{t}_ref_comp"])
        np.testing.assert_allclose(np.array(h["betas"]), z[f"t{t}_ref_betas"], rtol=1e-6, atol=1e-300)
        assert list(np.cumsum(np.array(h["nupd"]))) == list(z[f"t{t}_nupd"])
        # the walk crosses markers that were in the model (sweep.hip, "continuation": the fast layout and the all-dirty layout
        # of these cases), so a round may hold several residual updates -- and the chain above is still the oracle's, bit for bit
        assert sum(h["ncross"]) > 0 and h["nbatch"][-1] < h["nupd"][-1], (h["ncross"], h["nbatch"], h["nupd"])


def _null_case(N=20_000, M=6_000, iters=8, seed=11):
    """A phenotype without signal (y ~ N(0,1), what bench.py's headline workload uses): after the first sweeps well under
    1 % of the visits change an effect, runs exceed 64 markers and whole passes of the sampling wavefront are decided by
    the screen (sweep.hip, walk_piece) instead of the exact probabilities."""
    case = cases.Case("null", N, M, 1, 4, 1, 0.0, 60, seed, iters, 5)
    inp = cases.make_inputs(case)
    inp["y"][0] = np.random.default_rng(seed).normal(size=N)
    return case, inp


@pytest.mark.parametrize("name", [c.name for c in cases.CASES] + ["geo_fast", "geo_general", "null"])
def test_screened_sampling_path_is_the_oracle_chain(gpu, monkeypatch, name):
    """the branch that produces the headline number.  GMRM_SCREEN_MIN_RUN16=0 tries the cheap certain
    bound in front of the exact probabilities in EVERY pass of 64 markers (default: only when the recent run length is
    >= 48 markers), GMRM_NO_CROSS=1 keeps the launch on the kernels the screen is compiled into (the continuation
    kernels of the dense sweeps have none).  The chain must be the oracle's bit for bit -- components, effects, residual,
    hyper-parameters, .csv bytes -- with passes that the screen decided (n_screened_passes > 0 where runs are long) AND
    passes in which an uncertain lane sent everybody through the exact code (tries > screened)."""
    monkeypatch.setenv("GMRM_SCREEN_MIN_RUN16", "0")
    monkeypatch.setenv("GMRM_NO_CROSS", "1")
    if name == "null":
        case, inp = _null_case()
    elif name == "geo_fast":
        case = cases.Case("geo_fast", 50_000, 900, 1, 4, 1, 0.0, 300, 171014, 3, 20); inp = cases.make_inputs(case)
    elif name == "geo_general":
        case = cases.Case("geo_general", 20_003, 700, 3, 4, 1, 0.05, 200, 7, 3, 20); inp = cases.make_inputs(case)
    else:
        case = cases.CASE_BY_NAME[name]; inp = cases.make_inputs(case)
    got = cases.run_gpu(case, inp)
    want = cases.run_oracle(case, inp, canon=True)
    cases.assert_same_history(got, want, exact=True)
    for h in got:
        assert h["ncross"][-1] == 0                                     # (the kernels without continuation ran)
        assert h["nscrt"][-1] > 0, "the screen was never tried"
        assert h["nscrt"][-1] > h["nscr"][-1], "no pass fell back to the exact probabilities"
    if name == "null":
        h = got[0]
        assert h["nscr"][-1] > 10, (h["nscr"], h["nscrt"])               # whole passes decided by the bound (counters are per sweep)
        assert h["nupd"][-1] < 0.2 * case.M and h["nupd"][-1] < h["nupd"][0], h["nupd"]   # the update rate falls sweep by sweep
        print("null case: updates per sweep", h["nupd"], "screen tries", h["nscrt"], "screened passes", h["nscr"])
        ref = cases.run_oracle(case, inp, canon=False)                   # north_star bar against the reference-order arithmetic
        cases.assert_same_history(got, ref, exact=False, rtol=1e-6)


@pytest.mark.parametrize("N,M,env", [(20_000, 6_000, dict(GMRM_SWEEP_R="2")), (20_000, 6_000, dict()),
                                     (250_000, 2_500, dict(GMRM_NB_FACTOR16="256")), (250_000, 2_500, dict(GMRM_SWEEP_R="2", GMRM_NB_FACTOR16="256"))])
def test_long_batch_kernel_crosses_stops_and_is_the_oracle_chain(gpu, monkeypatch, N, M, env):
    """Round 4, GMRM_LONG_CROSS=1: in sparse models without missing genotypes the sweep runs on the long-batch kernel that crosses
    stops (sweep.hip, long_cont_body): batches of up to 240 markers walked by four wavefronts in parallel, the exact sums exchanged, the sums behind a
    crossed marker patched with its genotype products and decided again.  A phenotype with a few causal markers (they stay in
    the model: stops known in advance) on independent genotypes; few (N = 20 000: the reducers' many-rows path) and many
    workgroups (N = 250 000: one or two rows per reducer, on wavefront 0), with and without register-home tiles (R = 2 / R = 1)."""
    monkeypatch.setenv("GMRM_LONG_CROSS", "1")                          # (not the default: measured slower)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    case = cases.Case("lcross", N, M, 1, 4, 1, 0.0, 60, 4711, 8 if N < 100_000 else 5, 12)
    inp = cases.make_inputs(case)
    got = cases.run_gpu(case, inp)
    want = cases.run_oracle(case, inp, canon=True)
    cases.assert_same_history(got, want, exact=True)
    h = got[0]
    print("updates per sweep", h["nupd"], "rounds", h["nbatch"], "crossed stops", h["ncross"])
    assert sum(h["ncross"][2:]) > 0, h["ncross"]                       # (the first sweeps may run on the dense model's kernel)
    assert h["nbatch"][-1] < M // 20                                   # long batches


@pytest.mark.parametrize("name,k", [("small", 1), ("small", 37), ("ragged", 64), ("groups", 200), ("groups", 512)])
def test_sweep_in_parts_is_the_same_chain(gpu, name, k):
    """gmrm_sampler_begin_parts / _launch_part / _finish_part (the building block of `--sync-every k`): one shard that cuts
    every sweep into parts of k markers -- down to one marker per launch, and one part that is the whole sweep -- walks the
    chain of the ordinary sweep bit for bit (effects, components, residual, hyper-parameters, .csv bytes): the kernel takes a
    range of the visit order, the RNG stream continues from part to part, the component counts add up on the device and
    the new effects become current with the part that ends at M."""
    case = cases.CASE_BY_NAME[name]
    inp = cases.make_inputs(case)
    got = cases.run_gpu(case, inp, iters=3, parts=k)
    want = cases.run_oracle(case, inp, iters=3, canon=True)
    cases.assert_same_history(got, want, exact=True)
    whole = cases.run_gpu(case, inp, iters=3)
    for hg, hw in zip(got, whole):
        assert np.array_equal(hg["eps"], hw["eps"]) and hg["nupd"] == hw["nupd"]


@pytest.mark.parametrize("kw", [dict(shuffle=False), dict(mimic_hydra=True), dict(seed=0)])
def test_option_variants(gpu, kw):
    """--shuffle-markers 0, --mimic-hydra, --seed 0 (options.cpp:68-88, bayes.cpp:796-803)."""
    case = cases.CASE_BY_NAME["small"]
    inp = cases.make_inputs(case)
    got = cases.run_gpu(case, inp, iters=3, **kw)
    want = cases.run_oracle(case, inp, iters=3, canon=True, **kw)
    cases.assert_same_history(got, want, exact=True)


def test_empty_group_and_tiny_block(gpu):
    """A group without markers keeps sigmag = 0 (bayes.cpp:329-330,594-595); M smaller than
    one batch; markers in a zero-variance group consume no draw (bayes.cpp:396-400)."""
    case = cases.Case("tiny", 403, 13, 3, 4, 1, 0.1, 9, 5, 6, 3)
    inp = cases.make_inputs(case)
    inp["group_index"] = np.array([0, 0, 2, 0, 2, 0, 0, 2, 0, 0, 2, 0, 0], dtype=np.int32)   # group 1 empty
    got = cases.run_gpu(case, inp)
    want = cases.run_oracle(case, inp, canon=True)
    cases.assert_same_history(got, want, exact=True)
    assert got[0]["sigmag"][-1][1] == 0.0


def test_medium_size_two_sweeps(gpu):
    """N = 50 000 (BASELINE config 2's individuals), 1 500 markers: more workgroups, several
    MT block advances, batches of 8..64."""
    case = cases.Case("medium", 50_000, 1500, 2, 4, 1, 0.01, 500, 171014, 2, 30)
    inp = cases.make_inputs(case)
    got = cases.run_gpu(case, inp)
    want = cases.run_oracle(case, inp, canon=True)
    cases.assert_same_history(got, want, exact=True)
    ref = cases.run_oracle(case, inp, canon=False)
    cases.assert_same_history(got, ref, exact=False, rtol=1e-6)


def test_full_width_properties(gpu):
    """BASELINE individuals (N = 500 000) on synthetic device-generated genotypes: sizes the
    oracle cannot sweep in seconds, checked through size-independent properties."""
    N, M = 500_000, 3000
    rng = np.random.default_rng(1)
    from oracle import orc
    y = rng.normal(size=N)
    isna = (rng.random(N) < 0.05).astype(np.uint8)
    eps, mask4, nonas = orc.phen_prepare(y, isna)
    ctx = gmrm_amd.Context(N, M)
    ctx.synth_bed(171014, 0.4, 0.05)
    ctx.upload_trait(0, eps, mask4, nonas)
    mave, msig = ctx.compute_markers_statistics(0)
    assert np.all(np.isfinite(mave)) and np.all(msig > 0)
    assert abs(mave.mean() - 0.8) < 0.01
    # linearity of the dot product in the residual, exactly: dot(m; eps) - dot(m; eps after
    # update of marker m by dbeta) == dbeta * (nonas_m - 1)-ish is data dependent, so use the
    # oracle on a few columns instead (columns downloaded from the device)
    L = orc.lib()
    n4 = ctx.mbytes
    cols = ctx.download_bed(0, 4)
    eps_grid = orc.grid(eps)                       # what upload_trait stored: the residual on its 2^-44 grid
    for m in range(4):
        want = L.orc_dot_product_canon(cols[m].ctypes.data_as(orc.c_u8_p), eps_grid.ctypes.data_as(orc.c_double_p),
                                       n4, mave[m], msig[m])
        assert ctx.dot_product(m, mave[m], msig[m]) == want
    # one full sweep: every marker visited once, residual stays 0 at NA individuals, and the
    # residual equals y_std - mu - sum_j beta_j z_j (the chain's defining invariant)
    cva = np.array([[0.0, 0.0001, 0.001, 0.01]])
    smp = gmrm_amd.Sampler(ctx, 171014, cva, np.zeros(M, dtype=np.int32))
    smp.iterate(1)
    hy = smp.hyper(0)
    comp, betas, res = ctx.comp(0), ctx.betas(0), ctx.get_epsilon(0)
    assert int(np.bincount(comp, minlength=4).sum()) == M
    na_ind = np.repeat(mask4, 4) >> np.tile(np.arange(4), n4) & 1
    assert np.all(res[na_ind == 0] == 0.0)
    assert (betas != 0).sum() == (comp != 0).sum() == hy.m0_sum
    recon = eps.copy()
    recon[na_ind == 1] -= hy.mu
    nz = np.flatnonzero(betas)
    if nz.size:
        big = ctx.download_bed(int(nz.min()), int(nz.max() - nz.min() + 1))
        for j in nz:
            col = big[j - nz.min()]
            c = np.stack([(col >> (2 * k)) & 3 for k in range(4)], axis=-1).ravel()
            a = np.where(c == 0, 2.0, np.where(c == 2, 1.0, 0.0)); b = (c != 1).astype(np.float64)
            recon -= (a - mave[j] * b) * msig[j] * betas[j] * na_ind
    np.testing.assert_allclose(res, recon, rtol=0, atol=1e-9)
    assert hy.n_batches <= M and hy.n_updates >= nz.size
    smp.close(); ctx.close()


def _sharded_worker(rank, world, port, outdir, name, sync_every=0):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gmrm_amd.dist import HipEngine, ShardedDriver
    case = cases.CASE_BY_NAME[name]
    inp = cases.make_inputs(case)
    eps, mask4, nonas = cases.prepare_traits(inp)[0]
    S, M, _ = gmrm_amd.block_of_markers(case.M, world, rank)
    ctx = gmrm_amd.Context(case.N, M, Mt=case.M, S=S, T=1, device=0)
    ctx.upload_bed(inp["bed"][S:S + M])
    ctx.upload_trait(0, eps, mask4, nonas)
    smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"], rank=rank, nranks=world)
    drv = ShardedDriver(HipEngine(smp, torch.device("cuda", 0), host_staging=True))
    for it in range(1, 4):
        drv.iterate(it, sync_every=sync_every)
    hy = smp.hyper(0)
    np.savez(f"{outdir}/rank{rank}.npz", betas=ctx.betas(0), comp=ctx.comp(0), eps=ctx.get_epsilon(0),
             sigmae=hy.sigmae, sigmag=hy.sigmag, pi=hy.pi_est)
    smp.close(); ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,k", [("k3", 0), ("ragged", 0), ("ragged", 33)])
def test_marker_sharded_schedule_two_ranks(gpu, tmp_path, name, k):
    """The multi-GPU path end to end on one GPU: two processes (gloo, tensors staged through
    the host because both ranks share device 0), each sweeping its marker block with the HIP
    kernel and exchanging the exact residual deltas once per sweep -- or, k > 0, every k markers (`--sync-every k`:
    part launches of the sweep kernel) -- against the oracle's single-process statement of the same schedule
    (orc_ns_iterate / orc_nk_iterate), bit for bit."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path), name, k), nprocs=world, join=True)
    case = cases.CASE_BY_NAME[name]
    inp = cases.make_inputs(case)
    want = cases.run_oracle(case, inp, iters=3, canon=True, nranks=world, sync_every=k)[0]
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    assert np.array_equal(np.concatenate([x["comp"] for x in r]), want["comp"][-1])
    assert np.array_equal(np.concatenate([x["betas"] for x in r]), want["betas"][-1])
    for x in r:
        assert np.array_equal(x["eps"], want["eps"])
        assert float(x["sigmae"]) == want["sigmae"][-1]
        assert np.array_equal(x["sigmag"], want["sigmag"][-1])


def test_four_traits_run_as_concurrent_chains(gpu):
    """BASELINE config 4 in miniature: 4 phenotypes at N = 50 000 individuals sweep as four
    concurrent persistent launches (4 x 49 workgroups fit side by side), sharing the genotype
    block.  Every trait must equal its own single-trait oracle chain bit for bit; traits 0 and 1
    get identical phenotypes and must therefore stay identical (the reference's
    test1.phen / test1_bis.phen property: same seeds for every phenotype, bayes.cpp:796-803)."""
    case = cases.Case("four", 50_000, 600, 2, 4, 4, 0.0, 300, 31, 2, 30)
    inp = cases.make_inputs(case)
    inp["y"][1] = inp["y"][0]
    inp["isna"][1] = inp["isna"][0]
    got = cases.run_gpu(case, inp)
    want = cases.run_oracle(case, inp, canon=True)
    cases.assert_same_history(got, want, exact=True)
    assert np.array_equal(got[0]["betas"][-1], got[1]["betas"][-1]) and got[0]["csv"] == got[1]["csv"]
    assert not np.array_equal(got[0]["betas"][-1], got[2]["betas"][-1])


def test_missing_genotypes_and_24_groups_full_width(gpu):
    """BASELINE config 5's ingredients at full width (N = 500 000): 5 % phenotype NAs, 5 %
    missing genotypes (general exchange layout), 24 groups.  Checked against the oracle on a
    block the oracle can afford: two sweeps over 288 markers (12 per group)."""
    from oracle import orc
    N, M, G = 500_000, 288, 24
    rng = np.random.default_rng(9)
    y = rng.normal(size=N)
    isna = (rng.random(N) < 0.05).astype(np.uint8)
    eps, mask4, nonas = orc.phen_prepare(y, isna)
    ctx = gmrm_amd.Context(N, M)
    ctx.synth_bed(5, 0.4, 0.05)
    bed = ctx.download_bed()
    ctx.upload_trait(0, eps, mask4, nonas)
    cva = np.tile(np.array([0.0, 0.0001, 0.001, 0.01]), (G, 1))
    gi = (np.arange(M) % G).astype(np.int32)
    smp = gmrm_amd.Sampler(ctx, 77, cva, gi)
    ch = orc.Chain(N, bed, eps, mask4, nonas, gi, cva, 77, canon=True)
    ref = orc.Chain(N, bed, eps, mask4, nonas, gi, cva, 77, canon=False)      # the reference's own summation order
    for it in (1, 2):
        smp.iterate(it)
        ch.iterate(it)
        ref.iterate(it)
        assert np.array_equal(ctx.comp(0), ch.comp)
        assert np.array_equal(ctx.betas(0), ch.betas) and np.all(np.isfinite(ch.betas))
        hy = smp.hyper(0)
        assert hy.sigmae == ch.sigmae and np.array_equal(hy.sigmag, ch.sigmag)
        # north_star bar at BASELINE width: identical inclusion indices, effects within 1e-6 relative
        assert np.array_equal(ctx.comp(0), ref.comp), f"reference-order oracle picks other components in sweep {it}"
        np.testing.assert_allclose(ctx.betas(0), ref.betas, rtol=1e-6, atol=1e-300)
    assert np.array_equal(ctx.get_epsilon(0), ch.eps)
    np.testing.assert_allclose(ctx.get_epsilon(0), ref.eps, rtol=0, atol=1e-9)
    assert (ctx.betas(0) != 0).sum() > 0
    smp.close(); ctx.close()


def test_long_chain_stays_bit_exact(gpu):
    """~160 000 Gibbs decisions (N = 20 000, 4 000 markers, 40 sweeps, 3 groups): the chain on
    the GPU and in the oracle's order-independent mode must never part, and the oracle's
    reference-order mode must pick the same components for as long as f64 reassociation noise
    (~1e-16 per dot) does not flip a `prob <= acum` comparison -- reported, not asserted, beyond
    the first 10 sweeps (the reference itself is only that reproducible)."""
    case = cases.Case("long", 20_000, 4000, 3, 4, 1, 0.01, 200, 2024, 40, 60)
    inp = cases.make_inputs(case)
    got = cases.run_gpu(case, inp)
    want = cases.run_oracle(case, inp, canon=True)
    cases.assert_same_history(got, want, exact=True)
    ref = cases.run_oracle(case, inp, canon=False)
    same = [np.array_equal(a, b) for a, b in zip(got[0]["comp"], ref[0]["comp"])]
    assert all(same[:10]), "reference-order oracle diverged within 10 sweeps"
    first = same.index(False) + 1 if False in same else None
    print(f"reference-order oracle: components identical for {'all 40' if first is None else first - 1} sweeps")
    if first is None:
        np.testing.assert_allclose(got[0]["betas"][-1], ref[0]["betas"][-1], rtol=1e-6, atol=1e-300)


@pytest.mark.parametrize("env", [dict(GMRM_SWEEP_R="2"), dict(GMRM_SWEEP_R="4"),
                                 dict(GMRM_NB_FACTOR16="40", GMRM_SWEEP_R="1"), dict(GMRM_NB_FACTOR16="64", GMRM_SWEEP_R="2"),
                                 dict(GMRM_NB_FACTOR16="8"), dict(GMRM_NB_FACTOR16="256", GMRM_SWEEP_R="2", GMRM_NO_CROSS="1"),
                                 dict(GMRM_LONG_CROSS="2", GMRM_LONG_CROSS_FRAC16="1", GMRM_NB_FACTOR16="256", GMRM_SWEEP_R="2"),
                                 dict(GMRM_LONG_CROSS="2", GMRM_LONG_CROSS_FRAC16="4", GMRM_NB_FACTOR16="64", GMRM_SWEEP_R="4"),
                                 dict(GMRM_LONG_CROSS="2", GMRM_LONG_CROSS_FRAC16="1", GMRM_NB_FACTOR16="256"),
                                 dict(GMRM_NO_DIRECT_PUBLISH="1", GMRM_NO_TILE_TRIM="1", GMRM_TOTALS_DELAY="0", GMRM_TOTALS_DELAY2="40"),
                                 dict(GMRM_NB_FACTOR16="64", GMRM_BATCH_CAP="100", GMRM_TOTALS_DELAY="100")])
def test_kernel_geometries_and_schedules_give_the_same_chain(gpu, monkeypatch, env):
    """The results may not depend on how the kernel is laid out or scheduled: bytes per thread
    R = 1 / 2 / 4 (slice width, tile window, register-home tiles at R = 2 / 4, batch cap),
    larger and smaller batches (up to the 240-marker cap of the long-batch kernel, kept on it by GMRM_NO_CROSS); the long-batch
    kernel that crosses stops (GMRM_LONG_CROSS=2: in dense models as well) with crossings wherever a marker lies behind the stop;
    the round-4 schedule knobs at their non-default ends (rows published through LDS, untrimmed batches, waits before the polls,
    a batch cap).
    Cases: no missing genotypes (2-value exchange layout, packed partial sums) and 5 % missing (4-value layout)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for case in (cases.Case("geo_fast", 50_000, 900, 1, 4, 1, 0.0, 300, 171014, 3, 20),
                 cases.Case("geo_general", 20_003, 700, 3, 4, 1, 0.05, 200, 7, 3, 20)):
        inp = cases.make_inputs(case)
        got = cases.run_gpu(case, inp)
        want = cases.run_oracle(case, inp, canon=True)
        cases.assert_same_history(got, want, exact=True)


@pytest.mark.parametrize("K", [2, 5, 8])
def test_mixture_counts_other_than_the_examples(gpu, K):
    """K = 2 .. 8 mixture components per group (options.cpp:222-286 puts no bound on it; this build
    supports up to 8): the out-of-line sampling step and the wavefront-wide component search
    (one lane per (step, component): K (K-1) <= 56 lanes) against the oracle, bit for bit."""
    case = cases.Case(f"k{K}", 6_001, 500, 2, K, 1, 0.02, 40, 11 + K, 4, 40)
    inp = cases.make_inputs(case)
    got = cases.run_gpu(case, inp)
    want = cases.run_oracle(case, inp, canon=True)
    cases.assert_same_history(got, want, exact=True)
    assert max(int(np.max(c)) for c in got[0]["comp"]) >= 1          # some marker left component 0


@pytest.mark.parametrize("G,K", [(40, 4), (64, 8)])
def test_many_groups_tables_beyond_the_lds_budget(gpu, G, K):
    """G (1 + 3K) > 320 doubles: the per-group tables (sigmag, denominators, log pi, -log/2 terms) no
    longer fit their LDS slot and the sampling step reads them from HBM/L2; G = 64, K = 8 are the
    largest supported.  Several groups stay empty or tiny (bayes.cpp:329-330,594-595)."""
    case = cases.Case(f"g{G}", 5_003, 700, G, K, 1, 0.03, 25, 5 + G, 3, 30)
    inp = cases.make_inputs(case)
    got = cases.run_gpu(case, inp)
    want = cases.run_oracle(case, inp, canon=True)
    cases.assert_same_history(got, want, exact=True)


@pytest.mark.parametrize("one_in,calls", [(8, (1, 40)), (40, (300, 3000)), (3, (1, 400))])
def test_exchange_layout_is_chosen_per_batch(gpu, one_in, calls):
    """Only SOME markers have missing genotypes (a genotyping array with a few poorly called SNPs): clean markers
    exchange 2 values, dirty ones 4, inside one batch (round 1 put the whole launch on the 4-value layout as soon as
    one marker had a missing genotype).  A batch with a few dirty markers gathers their missing-genotype terms
    thread by thread (few calls: one LDS add per call; many: one per wavefront), one with many runs the indicator
    MFMAs on their tiles.  Results do not depend on any of that: bit-exact against the oracle; clean batches occur."""
    case = cases.Case("mixed", 30_000, 1500, 2, 4, 1, 0.0, 150, 77, 3, 30)
    inp = cases.make_inputs(case)
    rng = np.random.default_rng(5)
    dirty = rng.choice(case.M, size=case.M // one_in, replace=False)     # these markers get missing genotypes
    for m in dirty:
        who = rng.choice(case.N, size=int(rng.integers(*calls)), replace=False)
        for i in who:
            b, k = divmod(int(i), 4)
            inp["bed"][m, b] = (int(inp["bed"][m, b]) & (0xFF ^ (3 << (2 * k)))) | (1 << (2 * k))     # code 01 = missing
    traits = cases.prepare_traits(inp)
    ctx = gmrm_amd.Context(case.N, case.M, T=1)
    ctx.upload_bed(inp["bed"])
    ctx.upload_trait(0, *traits[0])
    smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"])
    want = cases.run_oracle(case, inp, canon=True)
    for it in range(1, case.iters + 1):
        smp.iterate(it)
        assert np.array_equal(ctx.comp(0), want[0]["comp"][it - 1]), f"component indices differ at iteration {it}"
        assert np.array_equal(ctx.betas(0), want[0]["betas"][it - 1])
        hy = smp.hyper(0)
        assert hy.sigmae == want[0]["sigmae"][it - 1]
        assert hy.n_fast_batches < hy.n_batches and (one_in < 8 or hy.n_fast_batches > 0), (hy.n_fast_batches, hy.n_batches)
    assert np.array_equal(ctx.get_epsilon(0), want[0]["eps"])
    smp.close()
    ctx.close()


@pytest.mark.parametrize("N,M,env", [(30_000, 6_000, dict(GMRM_SWEEP_R="2", GMRM_NB_FACTOR16="256")), (30_000, 6_000, dict()),
                                     (250_000, 2_400, dict(GMRM_SWEEP_R="2", GMRM_NB_FACTOR16="256")),
                                     (30_000, 6_000, dict(GMRM_SWEEP_R="4", GMRM_NB_FACTOR16="64"))])
def test_few_markers_with_missing_genotypes_run_on_the_long_batch_kernel(gpu, monkeypatch, N, M, env):
    """Round 4: a block in which FEW markers have missing genotypes (one in 60 here; real arrays always have some) and a sparse
    model no longer falls back to the short-batch kernel: k_sweep<R, 1, false, true> -- every tile on the one-MFMA-set pass, the Z
    terms of a batch's dirty markers gathered from the digit planes (slices of register-home markers staged by their owners), the
    dirty markers' rows of sum b eps published beside the packed marker rows, at most eight dirty markers per batch.  Bit-exact
    against the oracle, with and without register-home tiles, few and many workgroups."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    case = cases.Case("lmixed", N, M, 1, 4, 1, 0.0, 60, 4242, 7 if N < 100_000 else 5, 10)
    inp = cases.make_inputs(case)
    rng = np.random.default_rng(9)
    dirty = rng.choice(case.M, size=case.M // 60, replace=False)
    for m in dirty:
        who = rng.choice(case.N, size=int(rng.integers(1, 60)), replace=False)
        for i in who:
            b, k = divmod(int(i), 4)
            inp["bed"][m, b] = (int(inp["bed"][m, b]) & (0xFF ^ (3 << (2 * k)))) | (1 << (2 * k))     # code 01 = missing
    got = cases.run_gpu(case, inp)
    want = cases.run_oracle(case, inp, canon=True)
    cases.assert_same_history(got, want, exact=True)
    h = got[0]
    print("updates per sweep", h["nupd"], "rounds", h["nbatch"])
    assert h["nbatch"][-1] < M // 25                                   # long batches: the long-batch kernel ran


def test_per_step_calls_and_the_sweep_kernel_are_one_chain(gpu):
    """gmrm_sampler_begin_steps / _step / _end_steps (the reference's per-marker loop, bayes.cpp:376-492, with the
    residual update applied by the caller, bayes.cpp:681-706) interleaved with kernel sweeps on ONE shard: iteration 1
    and 3 through the persistent kernel, 2 and 4 marker by marker.  Every iteration must be the oracle's plain chain
    bit for bit (acum, the reference's per-step scratch, after the per-step iterations): the host restatement of the Gibbs step
    and the kernel are interchangeable."""
    from oracle import orc
    case = cases.CASE_BY_NAME["ragged"]
    inp = cases.make_inputs(case)
    traits = cases.prepare_traits(inp)
    ctx = gmrm_amd.Context(case.N, case.M, T=len(traits))
    ctx.upload_bed(inp["bed"])
    for t, (eps, mask4, nonas) in enumerate(traits):
        ctx.upload_trait(t, eps, mask4, nonas)
        ctx.compute_markers_statistics(t)
    smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"])
    chains = [orc.Chain(case.N, inp["bed"], eps, mask4, nonas, inp["group_index"], inp["cva"], case.seed, canon=True)
              for eps, mask4, nonas in traits]
    for it in range(1, 5):
        if it % 2:
            smp.iterate(it)
        else:
            smp.begin_steps(smp.draw_mu(it))
            changed = 0
            for mrki in range(case.M):
                mloc, d3 = smp.step(mrki)
                for t in range(len(traits)):
                    if d3[t, 0] != 0.0:
                        ctx.update_epsilon_from(ctx, d3[t], mloc, t)
                        changed += 1
            mloc, d3 = smp.step(case.M)                      # a task with fewer markers than the longest: shares nothing
            assert mloc == 0 and not d3.any()
            cass, bsq = smp.end_steps()
            smp.epilogue(cass, bsq)
            assert changed == sum(smp.hyper(t).n_updates for t in range(len(traits))) > 0
        for t, c in enumerate(chains):
            c.iterate(it)
            hy = smp.hyper(t)
            assert np.array_equal(ctx.betas(t), c.betas) and np.array_equal(ctx.comp(t), c.comp), (it, t)
            if it % 2 == 0:                                  # acum is the reference's per-marker scratch (bayes.cpp:445-474: written and
                assert np.array_equal(ctx.acum(t), c.acum), (it, t)   # read inside one marker step); the per-step entries keep it, the sweep kernel does not
            assert np.array_equal(ctx.get_epsilon(t), c.eps), (it, t)
            assert hy.sigmae == c.sigmae and hy.mu == c.mu and np.array_equal(hy.pi_est, np.asarray(c.pi_est).reshape(-1))
            assert smp.csv_line(t, it) == c.csv_line(it)
    with pytest.raises(gmrm_amd.GmrmError):
        smp.step(0)                                          # outside begin_steps .. end_steps


def _rccl_one_rank_worker(_index, port, out_path):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    from gmrm_amd.dist import HipEngine
    case = cases.CASE_BY_NAME["ragged"]
    inp = cases.make_inputs(case)
    eps, mask4, nonas = cases.prepare_traits(inp)[0]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    ctx = gmrm_amd.Context(case.N, case.M, T=1)
    ctx.upload_bed(inp["bed"])
    ctx.upload_trait(0, eps, mask4, nonas)
    mave, msig = ctx.compute_markers_statistics(0)
    smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"])
    eng = HipEngine(smp, dev)                                   # device tensors straight into the collective
    ctx.eps_snapshot(0)
    start = ctx.get_epsilon(0)
    for m in (3, 17, 40):
        ctx.update_epsilon([0.01 * (m + 1), mave[m], msig[m]], m, 0)
    moved = ctx.get_epsilon(0)
    q = eng.delta_export(0)                                     # written by this library into torch's memory
    assert q.is_cuda
    q_host = q.cpu().numpy().copy()                             # the two exact parts of (residual - snapshot)
    dist.all_reduce(q, op=dist.ReduceOp.SUM)                    # RCCL, one rank: must hand the same bits back
    unchanged = np.array_equal(q.cpu().numpy(), q_host)
    eng.delta_import(0, q)
    got = ctx.get_epsilon(0)
    n4 = len(start)
    want = start + (q_host[:n4] + q_host[n4:])                  # k_delta_import's expression on the host
    parts_ok = np.max(np.abs((q_host[:n4] + q_host[n4:]) - (moved - start))) < 2.0 ** -52
    np.savez(out_path, same=np.array_equal(got, want), moved=not np.array_equal(moved, start), unchanged=unchanged, parts_ok=parts_ok)
    smp.close()
    ctx.close()
    dist.destroy_process_group()


def test_residual_exchange_through_torch_memory_and_rccl(gpu, tmp_path):
    """What `bench.py --gpus N` does between sweeps, with the one rank a one-GPU box allows: the library writes the
    exact residual delta into a TORCH tensor on the device, torch.distributed (backend nccl = RCCL) all-reduces it in
    place, the library reads it back: the parts must survive the collective bit for bit and the residual must be
    snapshot + (part 1 + part 2) -- this is the path where two HIP runtimes in one process, a wrong stream or a
    host copy would show."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "r.npz"
    mp.spawn(_rccl_one_rank_worker, args=(port, str(out)), nprocs=1, join=True)
    z = np.load(out)
    assert bool(z["moved"]) and bool(z["parts_ok"]) and bool(z["unchanged"]) and bool(z["same"])
