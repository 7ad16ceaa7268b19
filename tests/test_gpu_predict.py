"""Bayes::predict (bayes.cpp:16-284) -- the two device building blocks against the
oracle's restatement of the reference loops, and bin/gmrm_hip --predict end to end (.bet in,
fixed-width .mlma out)."""
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

import gmrm_amd
from oracle import orc
from tests import cases
from tests.test_gpu_cli import _write_inputs, BIN

pytestmark = pytest.mark.gpu


def _ctx_with_trait(case, inp, t=0):
    eps, mask4, nonas = orc.phen_prepare(inp["y"][t], inp["isna"][t])
    ctx = gmrm_amd.Context(case.N, case.M, T=1)
    ctx.upload_bed(inp["bed"])
    ctx.upload_trait(0, eps, mask4, nonas)
    mave, msig = ctx.compute_markers_statistics(0)
    return ctx, eps, mask4, nonas, mave, msig


@pytest.mark.parametrize("name", ["small", "ragged"])
def test_predict_g_and_assoc_match_the_reference_loops(gpu, name):
    case = cases.CASE_BY_NAME[name]
    inp = cases.make_inputs(case)
    ctx, eps, mask4, nonas, mave, msig = _ctx_with_trait(case, inp)
    try:
        rng = np.random.default_rng(3)
        beta = rng.normal(0.0, 0.01, size=case.M)
        beta[rng.random(case.M) < 0.6] = 0.0                      # most posterior means are exactly 0
        g = ctx.predict_g(0, beta)
        want = orc.predict_g(inp["bed"], mask4, mave, msig, beta)[:case.N]
        # the contraction on the matrix cores: the exact sum rounded once (upstream adds the terms in no order)
        assert np.allclose(g, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
        yk = eps[:case.N] - 0.25 * g
        xtx, xty = ctx.assoc(0, yk)
        wxx, wxy = orc.assoc(inp["bed"], mask4, yk)
        assert np.array_equal(xtx, wxx)                           # integer counts
        assert np.allclose(xty, wxy, rtol=1e-12, atol=1e-12)      # f64 sums in a different order: 1e-12
        xtx2, xty2 = ctx.assoc(0, None)                           # default: the residual as uploaded
        wxx2, wxy2 = orc.assoc(inp["bed"], mask4, eps[:case.N])
        assert np.array_equal(xtx2, wxx2) and np.allclose(xty2, wxy2, rtol=1e-12, atol=1e-12)
    finally:
        ctx.close()


@pytest.mark.parametrize("scale", [1.0, 3.7e9, 2.0 ** -40])
def test_assoc_over_many_blocks_and_scales(gpu, scale):
    """gmrm_assoc on the matrix cores (k_assoc_mfma): several 2048-individual blocks with a ragged tail, a marker count
    that is no multiple of the 128 a workgroup takes, NAs and missing genotypes, phenotypes far from unit scale (the
    kernel scales y by a power of two before cutting it into digit planes).  xtx is an integer, xty the exact sum
    rounded once: within 1e-12 of the reference's left-to-right f64 sum.  predict_g on the same block, bit for bit."""
    case = cases.Case("assoc", 21_003, 333, 2, 4, 1, 0.03, 700, 11, 1, 20)
    inp = cases.make_inputs(case)
    ctx, eps, mask4, nonas, mave, msig = _ctx_with_trait(case, inp)
    try:
        rng = np.random.default_rng(8)
        yk = rng.normal(size=case.N) * scale
        yk[rng.integers(0, case.N, 50)] *= 40.0                  # a few outliers set the scale
        xtx, xty = ctx.assoc(0, yk)
        wxx, wxy = orc.assoc(inp["bed"], mask4, yk)
        assert np.array_equal(xtx, wxx)
        assert np.allclose(xty, wxy, rtol=1e-12, atol=1e-12 * scale)
        beta = rng.normal(0.0, 0.01, size=case.M)
        g = ctx.predict_g(0, beta)
        want_g = orc.predict_g(inp["bed"], mask4, mave, msig, beta)[:case.N]
        assert np.abs(g - want_g).max() <= 1e-12 * np.abs(want_g).max()
        zero = ctx.assoc(0, np.zeros(case.N))[1]
        assert not zero.any()
        present = np.flatnonzero(inp["isna"][0] == 0)
        absent = np.flatnonzero(inp["isna"][0] != 0)
        bad = yk.copy()
        bad[present[5]] = np.inf
        assert np.isnan(ctx.assoc(0, bad)[1]).all()              # a non-finite phenotype is reported, not summed
        ok = yk.copy()
        ok[absent[:3]] = [np.nan, np.inf, 1e300]                 # ... but not where the individual has no phenotype:
        xtx2, xty2 = ctx.assoc(0, ok)                            # those never enter the sums and must not set the scale
        assert np.array_equal(xtx2, xtx) and np.array_equal(xty2, xty)
    finally:
        ctx.close()


@pytest.mark.parametrize("scale,miss", [(1.0, 0.0), (2.0 ** 30, 0.0), (2.0 ** -45, 0.0), (1.0, 0.03), (2.0 ** 30, 0.004), (2.0 ** -45, 0.2)])
def test_predict_g_on_the_matrix_cores(gpu, scale, miss):
    """gmrm_predict_g (k_pg_mfma: markers on the K dimension, 2-bit codes transposed across DPP rows, digit planes of
    msig * beta; for a block with missing genotypes the indicator set against the digit planes of mave * msig * beta): several 2048-marker LDS stages with a ragged last one, more than one
    workgroup column of 32 768 markers, a ragged individual tail, phenotype NAs (their g is 0), effects spanning twelve
    orders of magnitude with most of them zero, scaled far from unit size.  Against the reference loop's in-order f64 sum:
    1e-12 of the largest |g| (the kernel's sum is exact and rounded once); and against the in-order device kernel."""
    import os
    case = cases.Case("pgm", 20_011, 40_000, 1, 4, 1, miss, 600, 21, 1, 20)
    inp = cases.make_inputs(case)
    ctx, eps, mask4, nonas, mave, msig = _ctx_with_trait(case, inp)
    try:
        rng = np.random.default_rng(4)
        beta = rng.normal(0.0, 0.01, size=case.M) * 10.0 ** rng.uniform(-9, 3, size=case.M) * scale
        beta[rng.random(case.M) < 0.7] = 0.0
        g = ctx.predict_g(0, beta)
        want = orc.predict_g(inp["bed"], mask4, mave, msig, beta)[:case.N]
        tol = 1e-12 * np.abs(want).max()
        assert np.abs(g - want).max() <= tol, (np.abs(g - want).max(), tol)
        assert not g[inp["isna"][0] != 0].any()
        os.environ["GMRM_PREDICT_LUT"] = "1"
        try:
            g_lut = ctx.predict_g(0, beta)
        finally:
            del os.environ["GMRM_PREDICT_LUT"]
        assert np.array_equal(g_lut, want) and np.abs(g - g_lut).max() <= tol
        assert not ctx.predict_g(0, np.zeros(case.M)).any()
        bad = beta.copy()
        bad[7] = np.nan
        assert np.isnan(ctx.predict_g(0, bad)[inp["isna"][0] == 0]).all()
    finally:
        ctx.close()


def test_cli_predict_writes_the_reference_mlma_records(gpu, tmp_path):
    assert BIN.exists(), "bin/gmrm_hip not built (python __graft_entry__.py)"
    case = cases.CASE_BY_NAME["ragged"]
    inp = cases.make_inputs(case)
    inp["cva"] = np.array([[float(f"{v:.5f}") for v in row] for row in inp["cva"]])
    phens = _write_inputs(tmp_path, case, inp)
    out = tmp_path / "out"
    common = ["--bed-file", str(tmp_path / "t.bed"), "--dim-file", str(tmp_path / "t.dim"),
              "--phen-files", ",".join(str(p) for p in phens), "--out-dir", str(out)]
    r = subprocess.run([str(BIN), *common, "--group-index-file", str(tmp_path / "t.gri"), "--group-mixture-file", str(tmp_path / "t.grm"),
                        "--seed", str(case.seed), "--iterations", "6", "--output-thin-rate", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # bim files: the current one names every marker; the reference one drops two ids and is rotated
    ids = [f"rs{1000 + i}" for i in range(case.M)]
    (tmp_path / "cur.bim").write_text("".join(f"1 {rid} 0 {i + 1} A C\n" for i, rid in enumerate(ids)))
    ref_ids = ids[5:] + ids[:3] + ["rsX1", "rsX2"]               # ids[3], ids[4] missing; same count as Mtot
    assert len(ref_ids) == case.M
    (tmp_path / "ref.bim").write_text("".join(f"1 {rid} 0 {i + 1} A C\n" for i, rid in enumerate(ref_ids)))
    r = subprocess.run([str(BIN), *common, "--predict", "--bim-file", str(tmp_path / "cur.bim"), "--ref-bim-file", str(tmp_path / "ref.bim")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Number of recorded iterations in .bet file 0: 3" in r.stdout
    assert f"WARNING: marker id {ids[3]} excluded -- no match" in r.stdout
    refpos = {rid: i for i, rid in enumerate(ref_ids)}
    for t in range(inp["y"].shape[0]):
        eps, mask4, nonas = orc.phen_prepare(inp["y"][t], inp["isna"][t])
        raw = (out / f"trait{t}.bet").read_bytes()
        M = struct.unpack("<I", raw[:4])[0]
        rec = 4 + 8 * M
        its = [np.frombuffer(raw[4 + k * rec + 4:4 + (k + 1) * rec], dtype="<f8") for k in range(3)]
        beta_sum = np.zeros(M)
        for b in its:
            beta_sum = beta_sum + b
        beta_sum = beta_sum / 3.0
        yk = eps[:case.N].copy()                                  # one rank: nothing is left out of y (bayes.cpp:141)
        sigma = 0.0
        for v in yk:
            sigma += v * v
        sigma /= nonas
        xtx, xty = orc.assoc(inp["bed"], mask4, yk)
        lines = (out / f"trait{t}.mlma").read_bytes()
        kept = [m for m in range(case.M) if ids[m] in refpos]
        assert len(lines) == 123 * len(kept)
        for k, m in enumerate(kept):
            got = lines[123 * k:123 * (k + 1)].decode()
            f = got.split()
            assert f[0] == ids[m] and int(f[1]) == m and int(f[2]) == refpos[ids[m]]
            beta, tdist, se, pval = orc.mlma_stats(xtx[m], xty[m], sigma)
            assert np.allclose([float(x) for x in f[3:7]], [beta, tdist, se, pval], rtol=0, atol=2e-12)
            if k < 3:                                             # the record layout itself, byte for byte, from its own numbers
                assert got.encode() == orc.mlma_line(ids[m], m, refpos[ids[m]], *[float(x) for x in f[3:7]])


def test_cli_predict_over_marker_shards(gpu, tmp_path):
    """--predict with --devices 0,0 (two marker shards; upstream: two MPI tasks, bayes.cpp:16-284): every shard computes g
    for its block, the sum is shared (MPI_Allreduce, :136), every shard tests its own markers against the phenotype minus
    the OTHER shard's genetic values (:141-142) and writes its records behind the first shard's (:246-252)."""
    assert BIN.exists(), "bin/gmrm_hip not built (python __graft_entry__.py)"
    from gmrm_amd.api import block_of_markers
    case = cases.CASE_BY_NAME["small"]
    inp = cases.make_inputs(case)
    inp["cva"] = np.array([[float(f"{v:.5f}") for v in row] for row in inp["cva"]])
    phens = _write_inputs(tmp_path, case, inp)
    out = tmp_path / "out"
    common = ["--bed-file", str(tmp_path / "t.bed"), "--dim-file", str(tmp_path / "t.dim"),
              "--phen-files", ",".join(str(p) for p in phens), "--out-dir", str(out)]
    r = subprocess.run([str(BIN), *common, "--group-index-file", str(tmp_path / "t.gri"), "--group-mixture-file", str(tmp_path / "t.grm"),
                        "--seed", str(case.seed), "--iterations", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ids = [f"rs{1000 + i}" for i in range(case.M)]
    (tmp_path / "cur.bim").write_text("".join(f"1 {rid} 0 {i + 1} A C\n" for i, rid in enumerate(ids)))
    (tmp_path / "ref.bim").write_text("".join(f"1 {rid} 0 {i + 1} A C\n" for i, rid in enumerate(ids)))
    r = subprocess.run([str(BIN), *common, "--predict", "--devices", "0,0", "--bim-file", str(tmp_path / "cur.bim"), "--ref-bim-file", str(tmp_path / "ref.bim")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    eps, mask4, nonas = orc.phen_prepare(inp["y"][0], inp["isna"][0])
    raw = (out / "trait0.bet").read_bytes()
    M = struct.unpack("<I", raw[:4])[0]
    rec = 4 + 8 * M
    beta_sum = np.zeros(M)
    for k in range(4):
        beta_sum = beta_sum + np.frombuffer(raw[4 + k * rec + 4:4 + (k + 1) * rec], dtype="<f8")
    beta_sum = beta_sum / 4.0
    L = orc.lib()
    n4 = cases.im4_of(case.N)
    mave, msig = np.empty(case.M), np.empty(case.M)
    L.orc_marker_stats_canon(inp["bed"].ctypes.data_as(orc.c_u8_p), case.N, case.M, n4, mask4.ctypes.data_as(orc.c_u8_p), nonas,
                             mave.ctypes.data_as(orc.c_double_p), msig.ctypes.data_as(orc.c_double_p))
    lines = (out / "trait0.mlma").read_bytes()
    assert len(lines) == 123 * case.M
    gk = []
    for rnk in range(2):
        S, Ml, _ = block_of_markers(case.M, 2, rnk)
        gk.append(orc.predict_g(inp["bed"][S:S + Ml], mask4, mave[S:S + Ml], msig[S:S + Ml], beta_sum[S:S + Ml])[:case.N])
    g = gk[0] + gk[1]
    for rnk in range(2):
        S, Ml, _ = block_of_markers(case.M, 2, rnk)
        yk = eps[:case.N] - (g - gk[rnk])
        sigma = float(np.sum(yk * yk)) / nonas
        xtx, xty = orc.assoc(inp["bed"][S:S + Ml], mask4, yk)
        for m in (0, 1, Ml // 2, Ml - 1):
            f = lines[123 * (S + m):123 * (S + m + 1)].decode().split()
            assert f[0] == ids[S + m] and int(f[1]) == S + m and int(f[2]) == S + m
            want = orc.mlma_stats(xtx[m], xty[m], sigma)
            assert np.allclose([float(x) for x in f[3:7]], want, rtol=1e-9, atol=1e-11), (rnk, m, f, want)
    # and it differs from the one-shard answer only through y_k: with one shard nothing is removed from y
    r1 = subprocess.run([str(BIN), *common, "--predict", "--bim-file", str(tmp_path / "cur.bim"), "--ref-bim-file", str(tmp_path / "ref.bim")],
                        capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0
    assert (out / "trait0.mlma").read_bytes() != lines
