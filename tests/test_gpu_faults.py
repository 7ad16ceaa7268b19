"""Failure paths of the persistent sweep kernel: its workgroups
wait for each other, so a grid that is not fully resident -- or a residual that leaves the range the exact
integer digits cover -- must end in a prompt, clean error, never in a hang or in silent garbage."""
import os
import time

import numpy as np
import pytest

import gmrm_amd
from gmrm_amd._lib import GmrmError
from tests import cases

pytestmark = pytest.mark.gpu
GMRM_ESTATE, GMRM_EKERNEL = -5, -6


def _setup(case, inp, eps_override=None):
    traits = cases.prepare_traits(inp)
    ctx = gmrm_amd.Context(case.N, case.M, T=1)
    ctx.upload_bed(inp["bed"])
    eps, mask4, nonas = traits[0]
    ctx.upload_trait(0, eps if eps_override is None else eps_override, mask4, nonas)
    return ctx, (eps, mask4, nonas)


def test_geometry_is_checked_against_the_occupancy_query(gpu):
    ctx = gmrm_amd.Context(500_000, 8, T=4)
    g = ctx.geometry()
    assert g["W"] * g["conc"] <= g["max_resident_wg"] and g["max_resident_wg"] >= g["num_cu"] >= 1
    assert (g["R"], g["conc"]) == (4, 2)          # BASELINE config 4 on one GPU: two chains side by side
    ctx.close()


def test_missing_workgroup_times_out_quickly_and_poisons_the_chain(gpu):
    """One workgroup short of the grid the kernel was told about (test hook GMRM_FAULT_DROP_WG): the
    grid-wide wait can never complete.  Expected: GMRM_EKERNEL within the spin bound (set to 200 ms here,
    4 s by default), the phenotype refuses further sweeps until it is uploaded again, and after that the
    chain is exactly the oracle's again."""
    case = cases.Case("fault", 3000, 64, 1, 4, 1, 0.0, 0, 11, 2, 5)      # N = 3000: three workgroups at R = 1
    inp = cases.make_inputs(case)
    os.environ["GMRM_SPIN_TIMEOUT_MS"] = "200"
    try:
        ctx, (eps, mask4, nonas) = _setup(case, inp)
    finally:
        del os.environ["GMRM_SPIN_TIMEOUT_MS"]
    assert ctx.geometry()["W"] >= 2
    smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"])
    os.environ["GMRM_FAULT_DROP_WG"] = "1"
    t0 = time.perf_counter()
    try:
        with pytest.raises(GmrmError) as ei:
            smp.iterate(1)
    finally:
        del os.environ["GMRM_FAULT_DROP_WG"]
    assert time.perf_counter() - t0 < 3.0, "the timeout path took longer than the spin bound allows"
    assert ei.value.code == GMRM_EKERNEL and "timed out" in str(ei.value)
    with pytest.raises(GmrmError) as ei2:                                  # poisoned: partly written outputs are not sampled from
        smp.iterate(1)
    assert ei2.value.code == GMRM_ESTATE and "upload the phenotype again" in str(ei2.value)
    smp.close()
    # recovery: a fresh upload + sampler gives the oracle's chain
    ctx.upload_trait(0, eps, mask4, nonas)
    ctx.compute_markers_statistics(0)
    smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"])
    want = cases.run_oracle(case, inp, iters=2, canon=True)
    for it in (1, 2):
        smp.iterate(it)
        assert np.array_equal(ctx.comp(0), want[0]["comp"][it - 1])
        assert np.array_equal(ctx.betas(0), want[0]["betas"][it - 1])
    smp.close()
    ctx.close()


def test_residual_outside_the_exact_range_is_reported_from_inside_the_sweep(gpu):
    """|eps| >= 2^8 would overflow the integer digit planes (sweep.hip, refresh_planes): the kernel raises
    error 4 itself instead of wrapping the digits silently; every workgroup leaves promptly."""
    case = cases.Case("range", 3000, 64, 1, 4, 1, 0.0, 0, 12, 1, 5)
    inp = cases.make_inputs(case)
    eps0, _, _ = cases.prepare_traits(inp)[0]
    bad = eps0.copy()
    bad[1500] = 300.0
    ctx, _ = _setup(case, inp, eps_override=bad)
    smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"])
    t0 = time.perf_counter()
    with pytest.raises(GmrmError) as ei:
        smp.iterate(1)
    assert time.perf_counter() - t0 < 8.0
    assert ei.value.code == GMRM_EKERNEL and "2^8" in str(ei.value)
    smp.close()
    ctx.close()


def test_open_per_step_sweep_blocks_other_entries_and_can_be_aborted(gpu, tmp_path):
    """while a per-step sweep is open (the device's effects are stale against the host copies, the residual is
    offset by -mu) a kernel sweep, a second per-step sweep, a save and a load are refused; gmrm_sampler_abort_steps puts
    the residual back and the sampler can go on."""
    case = cases.CASE_BY_NAME["small"]
    inp = cases.make_inputs(case)
    ctx, _ = _setup(case, inp)
    ctx.compute_markers_statistics(0)
    smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"])
    smp.iterate(1)
    before = ctx.get_epsilon(0)
    mu_prev = smp.hyper(0).mu
    ck = tmp_path / "ck.bin"
    smp.save(ck, 1)
    mu = smp.draw_mu(2)                                            # residual += previous mu
    smp.begin_steps(mu)                                            # residual -= mu
    smp.step(0)
    for call in (lambda: smp.begin_sweep(mu), lambda: smp.begin_steps(mu), lambda: smp.save(tmp_path / "x.bin", 1), lambda: smp.load(ck)):
        with pytest.raises(GmrmError) as ei:
            call()
        assert ei.value.code == GMRM_ESTATE and "per-step sweep is open" in str(ei.value)
    smp.abort_steps()
    smp.abort_steps()                                              # idempotent
    got = ctx.get_epsilon(0)                                       # draw_mu's offset stays; begin_steps' is undone
    keep = np.repeat(~np.asarray(inp["isna"][0], dtype=bool), 1)
    assert np.max(np.abs(got[:case.N][keep] - (before[:case.N][keep] + mu_prev))) < 1e-12
    # a step had been taken, so the caller may have applied residual updates whose effects were dropped with
    # the host copies -- nothing may sweep on (or save) until the chain state is replaced
    for call in (lambda: smp.iterate(2), lambda: smp.begin_steps(mu), lambda: smp.save(tmp_path / "y.bin", 1)):
        with pytest.raises(GmrmError) as ei:
            call()
        assert ei.value.code == GMRM_ESTATE and "abandoned" in str(ei.value)
    smp.load(ck)                                                   # and the chain state can be replaced again
    smp.iterate(2)
    want = cases.run_oracle(case, inp, iters=2, canon=True)
    assert np.array_equal(ctx.betas(0), want[0]["betas"][1])
    smp.close()
    ctx.close()


def test_abort_before_any_step_leaves_a_chain_that_can_go_on(gpu):
    """gmrm_sampler_abort_steps right after gmrm_sampler_begin_steps (no step taken, no residual update applied): the
    residual has its mu back and still is y - mu - X beta for the device's effects, so the sampler goes on WITHOUT a
    reload (iterate after an abort without loading, residual checked against the effects).  And a sweep in
    parts that has finished only some of its parts cannot be saved or closed."""
    case = cases.CASE_BY_NAME["small"]
    inp = cases.make_inputs(case)
    ctx, (eps0, mask4, nonas) = _setup(case, inp)
    mave, msig = ctx.compute_markers_statistics(0)
    smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"])
    smp.iterate(1)
    mu = smp.draw_mu(2)
    smp.begin_steps(mu)
    smp.abort_steps()                                              # nothing stepped: not "abandoned"
    smp.iterate(2)                                                 # (draws a fresh mu; the streams have moved on)
    smp.iterate(3)

    def invariant():
        hy = smp.hyper(0)
        g = ctx.predict_g(0, ctx.betas(0))
        want = eps0[:case.N] - hy.mu - g
        assert np.max(np.abs(ctx.get_epsilon(0)[:case.N] - want)) < 1e-9
    invariant()
    # a sweep in parts, stopped half way
    mu = smp.draw_mu(4)
    smp.begin_parts(mu)
    smp.launch_part(0, 100)
    smp.finish_part()
    for call in (lambda: smp.save("/tmp/never.bin", 4), lambda: smp.end_sweep(), lambda: smp.begin_steps(mu)):
        with pytest.raises(GmrmError) as ei:
            call()
        assert ei.value.code == GMRM_ESTATE and "sweep in parts is open" in str(ei.value)
    smp.launch_part(100, case.M - 100)                             # the remaining part: the sweep can be closed
    smp.finish_part()
    cass, bsq = smp.end_sweep()
    smp.epilogue(cass, bsq)
    invariant()
    smp.close()
    ctx.close()


def test_checkpoint_with_a_damaged_visit_order_is_refused(gpu, tmp_path):
    """the visit order of a checkpoint indexes the genotype block inside the kernel: a file whose header matches
    but whose order is not a permutation (or whose components exceed K, or that is longer than its header says) is refused."""
    case = cases.CASE_BY_NAME["small"]
    inp = cases.make_inputs(case)
    ctx, _ = _setup(case, inp)
    ctx.compute_markers_statistics(0)
    smp = gmrm_amd.Sampler(ctx, case.seed, inp["cva"], inp["group_index"])
    smp.iterate(1)
    ck = tmp_path / "ck.bin"
    smp.save(ck, 1)
    raw = bytearray(ck.read_bytes())
    # layout: magic 8, header 12 ints, then per phenotype: 3 doubles, 1 int, 624 + 1 ints, 624 + 1 ints, midx[M] ints, ...
    off_midx = 8 + 48 + 24 + 4 + 625 * 4 * 2
    bad = bytearray(raw)
    bad[off_midx:off_midx + 4] = (10 ** 6).to_bytes(4, "little")           # an index far outside the block
    (tmp_path / "bad1.bin").write_bytes(bad)
    bad = bytearray(raw)
    bad[off_midx:off_midx + 4] = bad[off_midx + 4:off_midx + 8]            # a duplicate: not a permutation
    (tmp_path / "bad2.bin").write_bytes(bad)
    (tmp_path / "bad3.bin").write_bytes(bytes(raw) + b"\0" * 8)            # trailing bytes
    for name in ("bad1.bin", "bad2.bin", "bad3.bin"):
        with pytest.raises(GmrmError) as ei:
            smp.load(tmp_path / name)
        assert "checkpoint" in str(ei.value)
    assert smp.load(ck) == 1                                               # the intact file still loads
    smp.close()
    ctx.close()


def _big_grid_worker(rank, outdir, lock_dir):
    """One of two PROCESSES on device 0, each with a sweep that needs (nearly) every compute unit."""
    import os
    os.environ["GMRM_LOCK_DIR"] = lock_dir
    os.environ["GMRM_SPIN_TIMEOUT_MS"] = "1500"
    import numpy as np
    import gmrm_amd
    N, M = 500_000, 1500
    rng = np.random.default_rng(100 + rank)
    ctx = gmrm_amd.Context(N, M)
    ctx.synth_bed(171014 + rank, 0.4, 0.0)
    eps, mask4, nonas = gmrm_amd.prepare_phenotype(rng.normal(size=N), np.zeros(N, dtype=np.uint8))
    ctx.upload_trait(0, eps, mask4, nonas)
    ctx.compute_markers_statistics(0)
    smp = gmrm_amd.Sampler(ctx, 7 + rank, np.array([[0.0, 0.0001, 0.001, 0.01]]), np.zeros(M, dtype=np.int32))
    ms = []
    for it in range(1, 9):
        smp.iterate(it)                                            # GMRM_EKERNEL (spin timeout) would raise here
        ms.append(smp.hyper(0).sweep_device_ms)
    np.save(f"{outdir}/ok{rank}.npy", np.array(ms))
    smp.close()
    ctx.close()


def test_two_processes_with_full_device_sweeps_alternate(gpu, tmp_path):
    """the in-flight accounting is per process.  Two processes whose sweeps each need 245 of the 256
    compute units cannot be co-resident: interleaved, their workgroups would wait for each other until the spin timeout.
    Such sweeps take a per-device advisory lock from launch to finish (capi.cpp, devlock_acquire) and alternate."""
    import torch.multiprocessing as mp
    g = gmrm_amd.Context(500_000, 8).geometry()
    assert 2 * g["W"] * g["conc"] > g["max_resident_wg"]            # the geometry the lock is for
    mp.spawn(_big_grid_worker, args=(str(tmp_path), str(tmp_path)), nprocs=2, join=True)
    for rank in (0, 1):
        ms = np.load(tmp_path / f"ok{rank}.npy")
        assert len(ms) == 8 and (ms > 0).all() and ms.max() < 1000.0
    assert any(p.name.startswith("gmrm_hip_") and p.name.endswith(".lock") for p in tmp_path.iterdir())
