"""GPU parity tests, per reference member function, through the C ABI (gmrm_amd.Context ->
libgmrm_hip.so).  The HIP kernels must equal the oracle's order-independent ("canon") mode
bit for bit and its reference-order mode to f64 reassociation noise."""
import ctypes as C

import numpy as np
import pytest

import gmrm_amd
from oracle import orc
from tests import cases

pytestmark = pytest.mark.gpu


def _setup(case, inp, t=0):
    eps, mask4, nonas = cases.prepare_traits(inp)[t]
    ctx = gmrm_amd.Context(case.N, case.M, T=1)
    ctx.upload_bed(inp["bed"])
    ctx.upload_trait(0, eps, mask4, nonas)
    return ctx, eps, mask4, nonas


@pytest.mark.parametrize("name", [c.name for c in cases.CASES])
def test_marker_statistics(gpu, name):
    case = cases.CASE_BY_NAME[name]
    inp = cases.make_inputs(case)
    ctx, eps, mask4, nonas = _setup(case, inp)
    mave, msig = ctx.compute_markers_statistics(0)
    L = orc.lib()
    n4 = cases.im4_of(case.N)
    want_a, want_s = np.empty(case.M), np.empty(case.M)
    ref_a, ref_s = np.empty(case.M), np.empty(case.M)
    args = (inp["bed"].ctypes.data_as(orc.c_u8_p), case.N, case.M, n4, mask4.ctypes.data_as(orc.c_u8_p), nonas)
    L.orc_marker_stats_canon(*args, want_a.ctypes.data_as(orc.c_double_p), want_s.ctypes.data_as(orc.c_double_p))
    L.orc_marker_stats(*args, ref_a.ctypes.data_as(orc.c_double_p), ref_s.ctypes.data_as(orc.c_double_p))
    assert np.array_equal(mave, want_a)
    assert np.array_equal(msig, want_s)
    np.testing.assert_allclose(mave, ref_a, rtol=1e-13)
    np.testing.assert_allclose(msig, ref_s, rtol=1e-12)      # tolerance: f64 reassociation of N terms
    ctx.close()


@pytest.mark.parametrize("name", [c.name for c in cases.CASES])
def test_dot_update_offset_sums(gpu, name):
    case = cases.CASE_BY_NAME[name]
    inp = cases.make_inputs(case)
    ctx, eps, mask4, nonas = _setup(case, inp)
    L = orc.lib()
    n4 = cases.im4_of(case.N)
    mave, msig = ctx.compute_markers_statistics(0)
    host = orc.grid(eps)                          # the library keeps the residual on the grid 2^-44 (gm_common.h)
    hp = host.ctypes.data_as(orc.c_double_p)
    mp = mask4.ctypes.data_as(orc.c_u8_p)
    rng = np.random.default_rng(3)
    for step, m in enumerate(rng.choice(case.M, size=12, replace=False)):
        col = inp["bed"][m].ctypes.data_as(orc.c_u8_p)
        got = ctx.dot_product(int(m), mave[m], msig[m])
        want = L.orc_dot_product_canon(col, hp, n4, mave[m], msig[m])
        ref = L.orc_dot_product(col, hp, n4, mave[m], msig[m])
        assert got == want, f"dot differs from the canon oracle at marker {m}"
        assert abs(got - ref) <= 1e-12 * np.sqrt(case.N)
        d3 = np.array([rng.normal(0, 0.05), mave[m], msig[m]])
        ctx.update_epsilon(d3, int(m))
        L.orc_update_epsilon_canon(hp, d3.ctypes.data_as(orc.c_double_p), col, mp, n4)
        if step % 4 == 0:
            off = float(rng.normal(0, 0.01))
            ctx.offset_epsilon(off)
            L.orc_offset_epsilon_canon(hp, off, mp, n4)
    dev = ctx.get_epsilon(0)
    assert np.array_equal(dev, host), "residual after update/offset differs"
    assert np.all(dev[np.repeat((mask4[:, None] >> np.arange(4)) & 1, 1).ravel() == 0] == 0.0)
    assert ctx.epsilon_sumsqr(0) == L.orc_epsilon_sumsqr_canon(hp, case.N)
    assert abs(ctx.epsilon_sumsqr(0) - L.orc_epsilon_sumsqr(hp, case.N)) <= 1e-12 * case.N
    assert ctx.update_epsilon_sigma(0) == L.orc_epsilon_sigma_canon(hp, mp, n4, nonas)
    ctx.close()


def _synth_numpy(N, M, S, seed, maf, miss):
    """tests-side restatement of the device generator (ops.hip k_synth)."""
    mbytes = cases.im4_of(N)
    mask = (1 << 64) - 1

    def mix(z):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))
    m = np.arange(M, dtype=np.uint64)[:, None]
    b = np.arange(mbytes, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        key = ((np.uint64(S) + m) * np.uint64(mbytes) + b) * np.uint64(3)
        g = np.uint64(0x9E3779B97F4A7C15)
        za = mix(np.uint64(seed) + (key + np.uint64(0)) * g)
        zb = mix(np.uint64(seed) + (key + np.uint64(1)) * g)
        zm = mix(np.uint64(seed) + (key + np.uint64(2)) * g)
    maf16, miss16 = int(maf * 65536.0), int(miss * 65536.0)
    out = np.zeros((M, mbytes), dtype=np.uint8)
    for k in range(4):
        ua = (za >> np.uint64(16 * k)) & np.uint64(0xFFFF)
        ub = (zb >> np.uint64(16 * k)) & np.uint64(0xFFFF)
        um = (zm >> np.uint64(16 * k)) & np.uint64(0xFFFF)
        copies = (ua < maf16).astype(np.int64) + (ub < maf16).astype(np.int64)
        code = np.where(copies == 2, 0, np.where(copies == 1, 2, 3)).astype(np.uint8)
        code[um < miss16] = 1
        valid = (np.arange(mbytes)[None, :] * 4 + k) < N
        out |= np.where(valid, code << (2 * k), 0).astype(np.uint8)
    return out


def test_synthetic_genotypes_and_bed_roundtrip(gpu):
    N, M, S = 1003, 257, 1000
    ctx = gmrm_amd.Context(N, M, Mt=5000, S=S)
    ctx.synth_bed(171014, 0.4, 0.05)
    got = ctx.download_bed()
    want = _synth_numpy(N, M, S, 171014, 0.4, 0.05)
    assert np.array_equal(got, want)
    codes = np.stack([(got >> (2 * k)) & 3 for k in range(4)], axis=-1).reshape(M, -1)[:, :N]
    assert abs((codes == 1).mean() - 0.05) < 0.01
    a = np.where(codes == 0, 2, np.where(codes == 2, 1, 0))[codes != 1]
    assert abs(a.mean() - 0.8) < 0.02                       # E[Binomial(2, 0.4)]
    rng = np.random.default_rng(0)
    blk = rng.integers(0, 256, size=(40, cases.im4_of(N)), dtype=np.uint8)
    ctx.upload_bed(blk, first=100)
    assert np.array_equal(ctx.download_bed(100, 40), blk)
    assert np.array_equal(ctx.download_bed(0, 100), want[:100])
    ctx.close()


def test_synthetic_genotypes_with_linkage_disequilibrium(gpu):
    """gmrm_synth_bed_ld: markers in blocks of 8, neighbours copy a haplotype's allele with probability 0.9.  Allele frequency
    stays at maf, neighbours inside a block correlate with r ~ 0.9 and markers two apart with r ~ 0.81, markers of different
    blocks do not correlate; the block structure follows the GLOBAL marker index (a shard sees the same genotypes), and
    ld_block = 0 is the independent generator bit for bit."""
    N, M, S = 20_000, 64, 40
    ctx = gmrm_amd.Context(N, M, Mt=1000, S=S)
    ctx.synth_bed(7, 0.4, 0.0, ld_block=8, ld_keep=0.9)
    bed = ctx.download_bed()
    codes = np.stack([(bed >> (2 * k)) & 3 for k in range(4)], axis=-1).reshape(M, -1)[:, :N]
    a = np.where(codes == 0, 2.0, np.where(codes == 2, 1.0, 0.0))
    assert np.all(np.abs(a.mean(axis=1) - 0.8) < 0.03)
    r = np.corrcoef(a)
    g = S + np.arange(M)                                     # global marker indices
    same1 = [(i, i + 1) for i in range(M - 1) if g[i] // 8 == g[i + 1] // 8]
    same2 = [(i, i + 2) for i in range(M - 2) if g[i] // 8 == g[i + 2] // 8]
    cross = [(i, i + 1) for i in range(M - 1) if g[i] // 8 != g[i + 1] // 8]
    assert abs(np.mean([r[i, j] for i, j in same1]) - 0.9) < 0.02
    assert abs(np.mean([r[i, j] for i, j in same2]) - 0.81) < 0.03
    assert max(abs(r[i, j]) for i, j in cross) < 0.05
    ctx2 = gmrm_amd.Context(N, 16, Mt=1000, S=S + 8)         # a shard that starts inside the block: the same genotypes
    ctx2.synth_bed(7, 0.4, 0.0, ld_block=8, ld_keep=0.9)
    assert np.array_equal(ctx2.download_bed(), bed[8:24])
    ctx2.synth_bed(7, 0.4, 0.0)
    ctx.synth_bed(7, 0.4, 0.0, ld_block=0, ld_keep=0.0)
    assert np.array_equal(ctx2.download_bed(), ctx.download_bed(8, 16))
    assert np.array_equal(ctx.download_bed(), _synth_numpy(N, M, S, 7, 0.4, 0.0))
    ctx.close(); ctx2.close()


def test_residual_exchange_is_exact(gpu):
    """delta export -> (sum over 'ranks') -> import reproduces the oracle's split2 sums."""
    import torch
    case = cases.CASE_BY_NAME["ragged"]
    inp = cases.make_inputs(case)
    ctx, eps, mask4, nonas = _setup(case, inp)
    eps = orc.grid(eps)                            # as stored by upload_trait
    ctx.eps_snapshot(0)
    mave, msig = ctx.compute_markers_statistics(0)
    ctx.update_epsilon(np.array([0.0123, mave[5], msig[5]]), 5)
    n4 = 4 * ctx.mbytes
    q = torch.zeros(2 * n4, dtype=torch.float64, device="cuda:0")
    ctx.eps_delta_export(0, q.data_ptr())
    after = ctx.get_epsilon(0)
    L = orc.lib()
    d = after - eps
    q1 = np.empty(n4); q2 = np.empty(n4)
    for i in range(n4):
        a, b = C.c_double(), C.c_double()
        L.orc_split2(float(d[i]), C.byref(a), C.byref(b))
        q1[i], q2[i] = a.value, b.value
    host_q = q.cpu().numpy()
    assert np.array_equal(host_q[:n4], q1) and np.array_equal(host_q[n4:], q2)
    q *= 3.0                                                # as if three ranks had made the same change
    ctx.eps_delta_import(0, q.data_ptr())
    assert np.array_equal(ctx.get_epsilon(0), eps + (3.0 * q1 + 3.0 * q2))
    ctx.close()


def test_error_paths(gpu):
    ctx = gmrm_amd.Context(1000, 10)
    with pytest.raises(gmrm_amd.GmrmError) as ei:
        ctx.epsilon_sumsqr(0)
    assert ei.value.code == -5                              # GMRM_ESTATE: phenotype not uploaded
    with pytest.raises(gmrm_amd.GmrmError):
        ctx.dot_product(99, 0.0, 1.0)
    with pytest.raises(gmrm_amd.GmrmError):
        gmrm_amd.Context(5_000_000, 10)                     # beyond the exact-summation range
    ctx.close()


def _selftest(op, x, nout=None):
    lib = gmrm_amd.load_library()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(nout if nout else x.size, dtype=np.float64)
    from gmrm_amd._lib import check
    check(lib.gmrm_selftest_math(0, op, x.ctypes.data_as(orc.c_double_p), y.ctypes.data_as(orc.c_double_p), x.size))
    return y


def test_device_arithmetic_is_bit_identical_to_host(gpu):
    """exp_, sqrt, division, split2 and the ziggurat normal on the GPU against the oracle's C
    (the parity claim is bit-for-bit, so the comparison is on bits, NaNs included)."""
    L = orc.lib()
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.uniform(-750, 712, 200_000), rng.normal(0, 3, 200_000),
                         [0.0, -0.0, 709.782712893384, 709.7827128933841, -745.1332191019412, -745.2,
                          1e-310, np.inf, -np.inf, np.nan, 700.0, -700.0]])
    got = _selftest(0, xs)
    want = np.array([L.orc_exp(float(v)) for v in xs])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    pos = np.abs(np.concatenate([rng.lognormal(0, 20, 100_000), [0.0, 1e-320, 4.0, np.inf]]))
    assert np.array_equal(_selftest(1, pos).view(np.uint64), np.sqrt(pos).view(np.uint64))
    den = np.concatenate([rng.lognormal(0, 30, 100_000) * rng.choice([-1.0, 1.0], 100_000), [3.0, 1e-310, 1e308]])
    assert np.array_equal(_selftest(2, den).view(np.uint64), (1.0 / den).view(np.uint64))
    e = rng.normal(0, 1.5, 50_000)
    q = _selftest(4, e, 2 * e.size).reshape(-1, 2)
    a, b = C.c_double(), C.c_double()
    for i in range(0, e.size, 97):
        L.orc_split2(float(e[i]), C.byref(a), C.byref(b))
        assert q[i, 0] == a.value and q[i, 1] == b.value
    n = 60_000                                  # enough draws to visit the wedge and tail branches
    z = _selftest(3, np.full(n, 4242.0))
    r = orc.OrcRng()
    L.orc_rng_seed(C.byref(r), 4242)
    zz = np.array([L.orc_rng_norm(C.byref(r), 0.0, 1.0) for _ in range(n)])
    assert np.array_equal(z.view(np.uint64), zz.view(np.uint64))
    assert np.abs(z).max() > 3.5                # the ziggurat tail was exercised
