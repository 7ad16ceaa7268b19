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
