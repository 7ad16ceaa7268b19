"""The BASELINE.json configurations that had no test of their own ("configs untested").

c1  example/ shape: 10 000 individuals x 20 000 markers, the reference's own phenotype file
    (example/test1.phen, committed as DATA under tests/golden/ together with test.dim / test.grm), its
    flags (example/part1_gcc_mvapich2.sh:15-24: --shuffle-markers 1 --seed 171014, one group), 100
    iterations, phenotypes test1 / test1_bis (identical file) / test1_nas (line 9 = NA).  example/test.bed
    is missing from the reference checkout (.MISSING_LARGE_BLOBS), so the genotypes are synthesised with
    the seeded recipe of SURVEY 8(d).  Checks: bin/gmrm_hip's .bet/.cpn/.csv are byte-identical to
    records built from the oracle chain; test1 == test1_bis byte for byte (the reference's built-in
    identity property); test1_nas differs.
c4  full width: N = 500 000 with T = 4 phenotypes does not fit four chains side by side; the context
    runs them as two queued pairs at R = 4 (capi.cpp: conc = 2).  Compared with the oracle bit for bit.
"""
import gzip
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

import gmrm_amd
from gmrm_amd import io
from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
BIN = ROOT / "bin" / "gmrm_hip"


def _example_phen_lines():
    return gzip.decompress((GOLD / "example_test1.phen.gz").read_bytes()).decode().splitlines()


def test_c1_example_shape_100_iterations(gpu, tmp_path):
    assert BIN.exists(), "bin/gmrm_hip not built (python __graft_entry__.py)"
    N, M = (int(x) for x in (GOLD / "example_test.dim").read_text().split())
    assert (N, M) == (10000, 20000)
    iters, seed = 100, 171014
    # genotypes: copies ~ Binomial(2, 0.4) (example/data_sim.R:15), the device generator keyed by (seed, marker, byte)
    ctx = gmrm_amd.Context(N, M)
    ctx.synth_bed(seed, 0.4, 0.0)
    bed = ctx.download_bed()
    ctx.close()
    io.write_bed(tmp_path / "test.bed", bed)
    assert (tmp_path / "test.bed").stat().st_size == 50_000_003
    (tmp_path / "test.dim").write_bytes((GOLD / "example_test.dim").read_bytes())
    (tmp_path / "test.grm").write_bytes((GOLD / "example_test.grm").read_bytes())
    (tmp_path / "test.gri").write_text("".join(f"{i} 0\n" for i in range(M)))
    lines = _example_phen_lines()
    assert len(lines) == N and lines[8].split()[:2] == ["9", "9"]
    (tmp_path / "test1.phen").write_text("\n".join(lines) + "\n")
    (tmp_path / "test1_bis.phen").write_text("\n".join(lines) + "\n")
    nas = list(lines)
    nas[8] = "9 9 NA"                                    # example/test1_nas.phen differs from test1.phen in this line only
    (tmp_path / "test1_nas.phen").write_text("\n".join(nas) + "\n")
    out = tmp_path / "test1"
    cmd = [str(BIN), "--bed-file", str(tmp_path / "test.bed"), "--dim-file", str(tmp_path / "test.dim"),
           "--phen-files", ",".join(str(tmp_path / f) for f in ("test1.phen", "test1_bis.phen", "test1_nas.phen")),
           "--group-index-file", str(tmp_path / "test.gri"), "--group-mixture-file", str(tmp_path / "test.grm"),
           "--shuffle-markers", "1", "--seed", str(seed), "--iterations", str(iters), "--out-dir", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"RESULT : It {iters}" in r.stdout
    files = {s: {e: (out / f"{s}.{e}").read_bytes() for e in ("bet", "cpn", "csv")} for s in ("test1", "test1_bis", "test1_nas")}
    for e in ("bet", "cpn", "csv"):
        assert files["test1"][e] == files["test1_bis"][e], f"test1 and test1_bis differ in .{e}"
        assert files["test1"][e] != files["test1_nas"][e], f"test1_nas does not differ in .{e}"
    assert len(files["test1"]["bet"]) == 4 + iters * (4 + 8 * M) and len(files["test1"]["cpn"]) == 4 + iters * (4 + 4 * M)

    # the oracle chain on the same inputs: all 100 iterations of test1, the first 25 of test1_nas
    cva = np.array([[float(v) for v in (GOLD / "example_test.grm").read_text().split()]])
    gi = np.zeros(M, dtype=np.int32)

    def oracle_records(phen_lines, n_it):
        y = np.array([0.0 if ln.split()[2] == "NA" else float(ln.split()[2]) for ln in phen_lines])
        isna = np.array([ln.split()[2] == "NA" for ln in phen_lines], dtype=np.uint8)
        eps, mask4, nonas = orc.phen_prepare(y, isna)
        ch = orc.Chain(N, bed, eps, mask4, nonas, gi, cva, seed, canon=True)
        bet, cpn, csv = [struct.pack("<I", M)], [struct.pack("<I", M)], []
        for it in range(1, n_it + 1):
            ch.iterate(it)
            bet.append(struct.pack("<I", it) + ch.betas.tobytes())
            cpn.append(struct.pack("<I", it) + ch.comp.astype("<i4").tobytes())
            csv.append(ch.csv_line(it))
        return b"".join(bet), b"".join(cpn), b"".join(csv)

    bet, cpn, csv = oracle_records(lines, iters)
    assert files["test1"]["cpn"] == cpn, "component indices differ from the oracle chain"
    assert files["test1"]["bet"] == bet and files["test1"]["csv"] == csv
    n25 = 25
    bet, cpn, csv = oracle_records(nas, n25)
    assert files["test1_nas"]["cpn"][:len(cpn)] == cpn and files["test1_nas"]["bet"][:len(bet)] == bet
    assert files["test1_nas"]["csv"][:len(csv)] == csv


def test_c4_four_traits_at_full_width_run_as_queued_pairs(gpu):
    """N = 500 000, T = 4: two chains side by side at R = 4 (123 workgroups each), the other two queued
    behind them -- BASELINE config 4's geometry on one GPU, against the oracle bit for bit."""
    N, M, T = 500_000, 320, 4
    rng = np.random.default_rng(44)
    ctx = gmrm_amd.Context(N, M, T=T)
    ctx.synth_bed(9, 0.4, 0.0)
    bed = ctx.download_bed()
    traits = []
    for t in range(T):
        eps, mask4, nonas = orc.phen_prepare(rng.normal(size=N), np.zeros(N, dtype=np.uint8))
        ctx.upload_trait(t, eps, mask4, nonas)
        traits.append((eps, mask4, nonas))
    cva = np.array([[0.0, 0.0001, 0.001, 0.01]])
    gi = np.zeros(M, dtype=np.int32)
    smp = gmrm_amd.Sampler(ctx, 171014, cva, gi)
    geo = ctx.geometry()
    assert geo["conc"] == 2 and geo["R"] == 4, geo        # the queued-pairs path, not four at once
    chains = [orc.Chain(N, bed, e, m4, na, gi, cva, 171014, canon=True) for (e, m4, na) in traits]
    refs = [orc.Chain(N, bed, e, m4, na, gi, cva, 171014, canon=False) for (e, m4, na) in traits]   # the reference's summation order
    for it in (1, 2):
        smp.iterate(it)
        for t, ch in enumerate(chains):
            ch.iterate(it)
            assert np.array_equal(ctx.comp(t), ch.comp), f"trait {t} iteration {it}: component indices"
            assert np.array_equal(ctx.betas(t), ch.betas)
            hy = smp.hyper(t)
            assert hy.sigmae == ch.sigmae and np.array_equal(hy.sigmag, ch.sigmag)
            refs[t].iterate(it)                              # north_star bar at N = 500 000: same indices, effects within 1e-6
            assert np.array_equal(ctx.comp(t), refs[t].comp), f"trait {t} iteration {it}: reference-order oracle differs"
            np.testing.assert_allclose(ctx.betas(t), refs[t].betas, rtol=1e-6, atol=1e-300)
    for t, ch in enumerate(chains):
        assert np.array_equal(ctx.get_epsilon(t), ch.eps)
    assert len({c.betas.tobytes() for c in chains}) == T   # four different chains
    smp.close()
    ctx.close()


@pytest.mark.parametrize("name,N,M,T,na_rate,miss_rate,G,dirty",
                         [("c2", 50_000, 100_000, 1, 0.002, 0.0, 1, 0.0),
                          ("c3", 500_000, 1_000_000, 1, 0.002, 0.0, 1, 0.0),
                          ("c4", 500_000, 1_000_000, 4, 0.002, 0.0, 1, 0.0),
                          ("c5", 500_000, 1_000_000, 1, 0.05, 0.05, 24, 0.0),
                          ("mixed", 500_000, 1_000_000, 1, 0.0, 0.001, 1, 0.005)])
def test_full_size_chain_keeps_its_invariant(gpu, name, N, M, T, na_rate, miss_rate, G, dirty):
    """BASELINE configs 3 and 5 (5 % NAs, 5 % missing genotypes, 24 groups: the 4-value exchange layout), and a block
    with missing calls in 0.5 % of the markers only (the per-marker layout with its sparse missing-genotype terms), at FULL size (500 000 individuals x 1 000 000 markers, device-generated genotypes, 125 GB):
    no oracle can sweep this in test time, so the chain is held to the property that defines it.  After k sweeps the
    residual must be  y_std - mu - sum_m beta_m z_m  for the effects the sweeps left behind, with z the standardised
    genotype columns: recomputed here from the kernel's own outputs with gmrm_predict_g (an independent kernel that
    adds the markers in order).  Every residual update of three sweeps (~140 000 columns of 125 KB) has to be right
    for this to hold to 1e-9; the bookkeeping (component counts, markers in the model, batches) is checked beside it.
    c2 (50 000 x 100 000) and c4 (four phenotypes over the 1M-marker block, two queued pairs at R = 4) are BASELINE's other
    two configurations at their own full sizes."""
    rng = np.random.default_rng(2)
    traits = []
    for t in range(T):
        y = rng.normal(size=N)
        isna = (rng.random(N) < na_rate).astype(np.uint8)
        traits.append(orc.phen_prepare(y, isna))
    ctx = gmrm_amd.Context(N, M, T=T)
    try:
        ctx.synth_bed(171014, 0.4, 0.0 if dirty else miss_rate)
        if dirty:                                                    # code 01 (missing) in `miss_rate` of the calls of a few markers
            nd, per = int(M * dirty), int(N * miss_rate)
            cols = ctx.download_bed(0, nd)
            who = rng.integers(0, N, size=(nd, per))
            rows = np.repeat(np.arange(nd), per)
            byte, sh = (who // 4).ravel(), (2 * (who % 4)).ravel().astype(np.uint8)
            cols[rows, byte] = (cols[rows, byte] & ~(np.uint8(3) << sh)) | (np.uint8(1) << sh)
            ctx.upload_bed(cols, 0)
            del cols, who, rows, byte, sh
        for t, (eps0, mask4, nonas) in enumerate(traits):
            ctx.upload_trait(t, eps0, mask4, nonas)
            ctx.compute_markers_statistics(t)
        cva = np.tile(np.array([[0.0, 0.0001, 0.001, 0.01]]), (G, 1)) * np.linspace(1.0, 2.0, G)[:, None]
        smp = gmrm_amd.Sampler(ctx, 171014, cva, rng.integers(0, G, M).astype(np.int32))
        if name == "c4":
            geo = ctx.geometry()
            assert geo["conc"] == 2 and geo["R"] == 4, geo                         # the queued-pairs path
        total_updates = 0
        n_sweeps = 6 if name == "c3" else 3          # c3: on into the sweeps bench.py times (few markers in the model, long runs)
        for it in range(1, n_sweeps + 1):
            smp.iterate(it)
            for t, (eps0, mask4, nonas) in enumerate(traits):
                keep = np.repeat(mask4, 4) >> np.tile(np.arange(4), len(mask4)) & 1        # 1 = phenotype present
                hy = smp.hyper(t)
                total_updates += hy.n_updates
                comp, betas = ctx.comp(t), ctx.betas(t)
                counts = np.bincount(comp, minlength=4)
                assert counts.sum() == M and int((betas != 0.0).sum()) == M - counts[0] == hy.m0_sum, name
                # a round ends at a residual update unless the walk crosses it (markers that were in the model)
                assert hy.n_batches >= hy.n_updates - hy.n_crossed_stops and 0.1 < hy.sigmae < 2.0
                if name == "c3" and it in (2, 3):
                    assert hy.n_crossed_stops > 0                                    # sweeps 2, 3: ~8 % of the markers in the model
                if name == "c3" and it == n_sweeps:
                    # the stationary regime of the headline workload: the launch is on the kernel without continuation and whole
                    # passes of the sampling wavefront are decided by the screen -- the branch that
                    # tests/test_gpu_chain.py::test_screened_sampling_path_is_the_oracle_chain holds to the oracle bit for bit
                    assert hy.n_crossed_stops == 0 and hy.n_screened_passes > 1000, (hy.n_screen_tries, hy.n_screened_passes)
                    assert hy.n_updates < 0.02 * M
                if dirty:
                    assert 0 < hy.n_fast_batches < hy.n_batches                      # clean and mixed batches both occur
                g = ctx.predict_g(t, betas)
                want = (eps0[:N] - hy.mu - g) * keep[:N]
                got = ctx.get_epsilon(t)[:N]
                assert np.max(np.abs(got - want)) < 1e-9, (it, t, float(np.max(np.abs(got - want))))
                assert not got[keep[:N] == 0].any()                                    # NA individuals stay out of the residual
        assert total_updates > (100_000 if M == 1_000_000 else 10_000) * T
        if T > 1:
            assert len({ctx.betas(t).tobytes() for t in range(T)}) == T               # different phenotypes, different chains
        smp.close()
    finally:
        ctx.close()


def test_largest_supported_width_one_workgroup_per_compute_unit(gpu):
    """The widest geometry: N = 256 workgroups x 256 threads x 4 bytes x 4 individuals = 1 048 576 (R = 4, one
    workgroup on every compute unit), and one more individual is refused with the limit in the message."""
    ncu = gmrm_amd.Context(64, 1).geometry()["num_cu"]
    N = min(ncu, 256) * 256 * 4 * 4
    M = 96
    rng = np.random.default_rng(12)
    ctx = gmrm_amd.Context(N, M)
    g = ctx.geometry()
    assert (g["R"], g["W"]) == (4, min(ncu, 256))
    ctx.synth_bed(3, 0.4, 0.0)
    bed = ctx.download_bed()
    eps, mask4, nonas = orc.phen_prepare(rng.normal(size=N), (rng.random(N) < 0.01).astype(np.uint8))
    ctx.upload_trait(0, eps, mask4, nonas)
    cva = np.array([[0.0, 0.0001, 0.001, 0.01]])
    gi = np.zeros(M, dtype=np.int32)
    smp = gmrm_amd.Sampler(ctx, 5, cva, gi)
    ch = orc.Chain(N, bed, eps, mask4, nonas, gi, cva, 5, canon=True)
    for it in (1, 2):
        smp.iterate(it)
        ch.iterate(it)
        assert np.array_equal(ctx.comp(0), ch.comp) and np.array_equal(ctx.betas(0), ch.betas)
    assert np.array_equal(ctx.get_epsilon(0), ch.eps)
    smp.close()
    ctx.close()
    with pytest.raises(gmrm_amd._lib.GmrmError) as ei:
        gmrm_amd.Context(N + 1, M)
    assert "N too large" in str(ei.value) and str(N) in str(ei.value)
