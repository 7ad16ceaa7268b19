"""CPU tests: the oracle against everything the reference itself can pin here
(its lookup-table headers and its own output writers, both compiled in place -- see
oracle/Makefile `ref` and tools/make_golden.py), the RNG spec's known answers, the two
summation modes against each other, and the committed golden chains."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import orc
from tests import cases

GOLD = cases.GOLD


def test_luts_match_reference_headers():
    """tests/golden/ref_luts.bin = dotp_lut_a|dotp_lut_b|dotp_lut_ab|na_lut dumped from the
    reference's src/dotp_lut.hpp and src/na_lut.hpp."""
    raw = np.fromfile(GOLD / "ref_luts.bin", dtype=np.float64)
    assert raw.size == 1024 + 1024 + 2048 + 64
    ref_a, ref_b, ref_ab, ref_na = raw[:1024], raw[1024:2048], raw[2048:4096], raw[4096:]
    L = orc.lib()
    a = np.ctypeslib.as_array(L.orc_dotp_lut_a(), shape=(1024,))
    b = np.ctypeslib.as_array(L.orc_dotp_lut_b(), shape=(1024,))
    na = np.ctypeslib.as_array(L.orc_na_lut(), shape=(64,))
    assert np.array_equal(a, ref_a)
    assert np.array_equal(b, ref_b)
    assert np.array_equal(na, ref_na)
    # dotp_lut_ab = 4 a's then 4 b's per byte (src/lut/mk_lut.cpp:75-118)
    ab = np.concatenate([a.reshape(256, 4), b.reshape(256, 4)], axis=1).ravel()
    assert np.array_equal(ab, ref_ab)


def _parse_spec():
    lines = (GOLD / "ref_xfiles.spec.txt").read_text().strip().split("\n")
    n_it, G, K, Mtot = (int(x) for x in lines[0].split())
    recs = []
    for ln in lines[1:]:
        tok = ln.split()
        p = 0
        it = int(tok[p]); p += 1
        sg = np.array([float(x) for x in tok[p:p + G]]); p += G
        se = float(tok[p]); p += 1
        m0 = int(tok[p]); p += 1
        pi = np.array([float(x) for x in tok[p:p + G * K]]); p += G * K
        betas = np.array([float(x) for x in tok[p:p + Mtot]]); p += Mtot
        comp = np.array([int(x) for x in tok[p:p + Mtot]], dtype=np.int32)
        recs.append((it, sg, se, m0, pi, betas, comp))
    return G, K, Mtot, recs


def test_csv_record_matches_reference_writer():
    """orc_csv_line against records written by the reference's own write_ofile_csv."""
    G, K, Mtot, recs = _parse_spec()
    want = (GOLD / "ref_xfiles.csv").read_bytes()
    L = orc.lib()
    got = b""
    for it, sg, se, m0, pi, _, _ in recs:
        buf = C.create_string_buffer(50000)
        n = L.orc_csv_line(buf, 50000, it, sg.ctypes.data_as(orc.c_double_p), G, se, m0,
                           pi.ctypes.data_as(orc.c_double_p), K)
        got += buf.raw[:n]
    assert got == want


def test_history_writers_match_reference_writer(tmp_path):
    """gmrm_amd.io writers against .bet/.cpn written by the reference's write_ofile_h1."""
    from gmrm_amd import io
    G, K, Mtot, recs = _parse_spec()
    wb = io.HistoryWriter(tmp_path / "x.bet", Mtot, np.float64)
    wc = io.HistoryWriter(tmp_path / "x.cpn", Mtot, np.int32)
    for n, (it, _, _, _, _, betas, comp) in enumerate(recs):
        # two "ranks" writing their own slices, as bayes.cpp:666-667 does
        half = Mtot // 2
        wb.write(it, n, betas[:half], S=0); wb.write(it, n, betas[half:], S=half)
        wc.write(it, n, comp[:half], S=0); wc.write(it, n, comp[half:], S=half)
    assert (tmp_path / "x.bet").read_bytes() == (GOLD / "ref_xfiles.bet").read_bytes()
    assert (tmp_path / "x.cpn").read_bytes() == (GOLD / "ref_xfiles.cpn").read_bytes()
    Mt, its, vals = io.read_history(GOLD / "ref_xfiles.bet", np.float64)
    assert Mt == Mtot and list(its) == [r[0] for r in recs]
    assert np.array_equal(vals[1], recs[1][5])


def test_mt19937_known_answer():
    """C++11 [rand.predef]: the 10000th output of mt19937 seeded with 5489 is 4123659995."""
    L = orc.lib()
    r = orc.OrcRng()
    L.orc_rng_seed(C.byref(r), 5489)
    v = 0
    for _ in range(10000):
        v = L.orc_rng_u32(C.byref(r))
    assert v == 4123659995


def test_uniform_is_one_word_over_2_32():
    L = orc.lib()
    r1, r2 = orc.OrcRng(), orc.OrcRng()
    L.orc_rng_seed(C.byref(r1), 77)
    L.orc_rng_seed(C.byref(r2), 77)
    for _ in range(100):
        assert L.orc_rng_unif(C.byref(r1)) == L.orc_rng_u32(C.byref(r2)) / 4294967296.0


def test_exp_spec_accuracy():
    L = orc.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-745, 709, 20000), rng.uniform(-3, 3, 20000), [0.0, -0.0, 1.0, 700.0, -700.0]])
    got = np.array([L.orc_exp(float(x)) for x in xs])
    want = np.exp(xs)
    ok = want > 1e-300
    assert np.max(np.abs(got[ok] - want[ok]) / want[ok]) < 4.5e-16
    assert L.orc_exp(0.0) == 1.0
    assert L.orc_exp(800.0) == math.inf and L.orc_exp(-800.0) == 0.0


def test_split2_is_exact_and_order_free():
    L = orc.lib()
    rng = np.random.default_rng(5)
    x = rng.normal(0, 1.3, 4096)
    q1 = np.empty_like(x); q2 = np.empty_like(x)
    for i, v in enumerate(x):
        a, b = C.c_double(), C.c_double()
        L.orc_split2(float(v), C.byref(a), C.byref(b))
        q1[i], q2[i] = a.value, b.value
    assert np.all(np.abs(x - (q1 + q2)) <= 2.0 ** -54)
    assert np.all(q1 == np.round(q1 * 2.0 ** 22) / 2.0 ** 22)
    for _ in range(5):
        p = rng.permutation(x.size)
        assert float(np.sum(2.0 * q1[p])) == float(np.sum(2.0 * q1))      # exact => order-free
        assert float(np.sum(q2[p])) == float(np.sum(q2))


@pytest.mark.parametrize("name", [c.name for c in cases.CASES])
def test_canon_and_reference_order_agree(name):
    """The order-independent reductions equal the reference-order loops up to f64
    reassociation noise, per call."""
    case = cases.CASE_BY_NAME[name]
    inp = cases.make_inputs(case)
    eps, mask4, nonas = cases.prepare_traits(inp)[0]
    L = orc.lib()
    n4 = cases.im4_of(case.N)
    bed = inp["bed"]
    M = case.M
    mave_r, msig_r = np.empty(M), np.empty(M)
    mave_c, msig_c = np.empty(M), np.empty(M)
    bp, mp = bed.ctypes.data_as(orc.c_u8_p), mask4.ctypes.data_as(orc.c_u8_p)
    L.orc_marker_stats(bp, case.N, M, n4, mp, nonas, mave_r.ctypes.data_as(orc.c_double_p), msig_r.ctypes.data_as(orc.c_double_p))
    L.orc_marker_stats_canon(bp, case.N, M, n4, mp, nonas, mave_c.ctypes.data_as(orc.c_double_p), msig_c.ctypes.data_as(orc.c_double_p))
    np.testing.assert_allclose(mave_c, mave_r, rtol=1e-13)
    np.testing.assert_allclose(msig_c, msig_r, rtol=1e-12)
    ep = eps.ctypes.data_as(orc.c_double_p)
    scale = np.sqrt(case.N)
    for m in range(0, M, 37):
        col = bed[m].ctypes.data_as(orc.c_u8_p)
        a = L.orc_dot_product(col, ep, n4, mave_r[m], msig_r[m])
        b = L.orc_dot_product_canon(col, ep, n4, mave_r[m], msig_r[m])
        assert abs(a - b) <= 1e-12 * scale
    s_r, s_c = L.orc_epsilon_sumsqr(ep, case.N), L.orc_epsilon_sumsqr_canon(ep, case.N)
    assert abs(s_r - s_c) <= 1e-12 * s_r
    g_r, g_c = L.orc_epsilon_sigma(ep, mp, n4, nonas), L.orc_epsilon_sigma_canon(ep, mp, n4, nonas)
    assert abs(g_r - g_c) <= 1e-12 * g_r


@pytest.mark.parametrize("name", [c.name for c in cases.CASES])
def test_oracle_reproduces_golden_chain(name):
    """The oracle as built here reproduces the committed golden outputs bit for bit, and
    its two summation modes pick the same components with betas within 1e-9."""
    case = cases.CASE_BY_NAME[name]
    inp, z = cases.load_golden(name)
    fresh = cases.make_inputs(case)
    assert np.array_equal(fresh["bed"], inp["bed"]) and np.array_equal(fresh["y"], inp["y"])
    hist = cases.run_oracle(case, inp, canon=True)
    ref = cases.run_oracle(case, inp, canon=False)
    for t, (h, hr) in enumerate(zip(hist, ref)):
        assert np.array_equal(np.array(h["comp"], dtype=np.int8), z[f"t{t}_comp"])
        assert np.array_equal(np.array(h["betas"]), z[f"t{t}_betas"])
        assert np.array_equal(np.array(h["sigmae"]), z[f"t{t}_sigmae"])
        assert np.array_equal(np.array(h["sigmag"]), z[f"t{t}_sigmag"])
        assert b"".join(h["csv"]) == z[f"t{t}_csv"].tobytes()
        assert np.array_equal(np.array(hr["comp"], dtype=np.int8), z[f"t{t}_ref_comp"])
        assert np.array_equal(z[f"t{t}_ref_comp"], z[f"t{t}_comp"])
        np.testing.assert_allclose(np.array(hr["betas"]), z[f"t{t}_betas"], rtol=1e-9, atol=1e-300)


def test_reference_fixture_identity_property():
    """example/test1.phen == example/test1_bis.phen by design: identical inputs
    with the shared seeds give identical chains; a different trait gives a different one."""
    case = cases.CASE_BY_NAME["ragged"]
    inp = cases.make_inputs(case)
    inp2 = dict(inp)
    inp2["y"] = np.array([inp["y"][0], inp["y"][0]])
    inp2["isna"] = np.array([inp["isna"][0], inp["isna"][0]])
    h = cases.run_oracle(case, inp2, iters=3)
    assert np.array_equal(h[0]["betas"][-1], h[1]["betas"][-1])
    assert h[0]["csv"] == h[1]["csv"]
    h3 = cases.run_oracle(case, inp, iters=3)
    assert not np.array_equal(h3[0]["betas"][-1], h3[1]["betas"][-1])


def test_sharded_schedule_single_rank_is_plain_iterate():
    case = cases.CASE_BY_NAME["k3"]
    inp = cases.make_inputs(case)
    a = cases.run_oracle(case, inp, iters=3, nranks=1)
    b = cases.run_oracle(case, inp, iters=3, nranks=2)
    # two shards run a different (sweep-synchronous) chain; it must still be a sane sampler
    assert not np.array_equal(a[0]["betas"][-1], b[0]["betas"][-1])
    assert 0.1 < b[0]["sigmae"][-1] < 2.0
    assert np.all(np.isfinite(b[0]["eps"]))


def test_per_step_schedule_of_the_reference():
    """orc_ps_iterate (bayes.cpp:374-553 with several MPI tasks).  One task: it IS the plain chain.  Several: every
    replica receives the same updates in the same order, so the replicas differ by their own mu only
    (bayes.cpp:348-358 draws it per task) -- and the chain is neither the 1-task nor the per-sweep-exchange one."""
    case = cases.CASE_BY_NAME["k3"]
    inp = cases.make_inputs(case)
    a = cases.run_oracle(case, inp, iters=3, nranks=1)
    b = cases.run_oracle(case, inp, iters=3, nranks=1, schedule="steps")
    for k in ("betas", "comp", "sigmag", "pi"):
        assert all(np.array_equal(x, y) for x, y in zip(a[0][k], b[0][k])), k
    assert a[0]["csv"] == b[0]["csv"] and np.array_equal(a[0]["eps"], b[0]["eps"])

    from gmrm_amd.api import block_of_markers
    eps, mask4, nonas = cases.prepare_traits(inp)[0]
    chains = []
    for r in range(3):
        S, Ml, _ = block_of_markers(case.M, 3, r)
        chains.append(orc.Chain(case.N, inp["bed"][S:S + Ml], eps, mask4, nonas, inp["group_index"], inp["cva"],
                                case.seed, Mt=case.M, S=S, rank=r, canon=True))
    for it in range(1, 4):
        orc.ps_iterate(chains, it)
    keep = ~np.asarray(inp["isna"][0], dtype=bool)
    base = chains[0].eps[:case.N][keep] + chains[0].mu
    assert len({c.mu for c in chains}) == 3                       # per-task seeds (bayes.cpp:796-803)
    for c in chains[1:]:
        assert np.max(np.abs(c.eps[:case.N][keep] + c.mu - base)) < 1e-10
        assert c.sigmae == chains[0].sigmae and np.array_equal(c.pi_est, chains[0].pi_est)
    ns = cases.run_oracle(case, inp, iters=3, nranks=3)
    ps = cases.run_oracle(case, inp, iters=3, nranks=3, schedule="steps")
    assert np.array_equal(ps[0]["betas"][-1], np.concatenate([c.betas for c in chains]))
    assert not np.array_equal(ps[0]["betas"][-1], ns[0]["betas"][-1])
    assert not np.array_equal(ps[0]["betas"][-1], a[0]["betas"][-1])
    assert 0.1 < ps[0]["sigmae"][-1] < 2.0


def test_multi_rank_schedules_reproduce_their_fixture():
    """tests/golden/shards_k3_3ranks.npz: 4 iterations of case k3 on 3 ranks under the two multi-rank schedules
    (orc_ns_iterate: one exchange per sweep; orc_ps_iterate: the reference's exchange per marker step).  Written by
    this oracle (the reference's MPI build cannot run here): a regression pin for both restatements, which the GPU
    tests then hold the product to."""
    z = np.load(GOLD / "shards_k3_3ranks.npz")
    case = cases.CASE_BY_NAME["k3"]
    inp, _ = cases.load_golden("k3")
    for sched in ("sweep", "steps"):
        h = cases.run_oracle(case, inp, iters=4, canon=True, nranks=3, schedule=sched)[0]
        assert np.array_equal(np.array(h["comp"], dtype=np.int8), z[f"{sched}_comp"]), sched
        assert np.array_equal(np.array(h["betas"]), z[f"{sched}_betas"]), sched
        assert b"".join(h["csv"]) == z[f"{sched}_csv"].tobytes(), sched


def _stl_shuffle_fixture():
    import gzip
    out = []
    for ln in gzip.open(GOLD / "stl_shuffle.txt.gz", "rt"):
        head, perm = ln.split(":")
        n, seed = (int(x) for x in head.split())
        out.append((n, seed, np.array(perm.split(), dtype=np.int32)))
    return out


def test_shuffle_matches_libstdcxx_random_shuffle():
    """Phenotype::shuffle_midx (src/phenotype.cpp:314-323) = boost::range::random_shuffle = std::random_shuffle(first, last,
    rand) driven by an mt19937.  tests/golden/stl_shuffle.txt.gz holds the permutations libstdc++'s own std::random_shuffle
    and std::mt19937 produce (oracle/ref_harness/stl_shuffle.cpp, generator = the restated uniform_int rule; Boost itself is
    absent): the oracle's restatement of the loop and of the engine must give the same permutations, n = 2, 3, 777, 20000."""
    L = orc.lib()
    fx = _stl_shuffle_fixture()
    assert len(fx) == 12 and {n for n, _, _ in fx} == {2, 3, 777, 20000}
    for n, seed, want in fx:
        assert sorted(want.tolist()) == list(range(n))
        r = orc.OrcRng()
        L.orc_rng_seed(C.byref(r), seed)
        v = np.arange(n, dtype=np.int32)
        L.orc_rng_shuffle(C.byref(r), v.ctypes.data_as(orc.c_int_p), n)
        assert np.array_equal(v, want), (n, seed)


def test_product_shuffle_matches_libstdcxx_random_shuffle():
    """The same fixture against the product's host shuffle (gm_rng.h gm::shuffle through gmrm_selftest_shuffle; host code, no GPU)."""
    import gmrm_amd
    lib = gmrm_amd.load_library()
    for n, seed, want in _stl_shuffle_fixture():
        v = np.zeros(n, dtype=np.int32)
        assert lib.gmrm_selftest_shuffle(seed, n, v.ctypes.data_as(orc.c_int_p)) == 0
        assert np.array_equal(v, want), (n, seed)


def test_predict_restatement_against_numpy():
    """Bayes::predict's loops (bayes.cpp:93-122, 172-205, 233-234) as restated in the oracle, checked
    against a direct numpy evaluation of the same formulas on decoded genotypes (no GPU)."""
    import math
    from tests import cases
    case = cases.Case("pred", 203, 37, 1, 4, 1, 0.06, 9, 3, 1, 5)
    inp = cases.make_inputs(case)
    eps, mask4, nonas = orc.phen_prepare(inp["y"][0], inp["isna"][0])
    N, M = case.N, case.M
    n4 = inp["bed"].shape[1]
    code = np.stack([(inp["bed"] >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(M, 4 * n4)
    a = np.select([code == 0, code == 2], [2.0, 1.0], 0.0)
    b = (code != 1).astype(np.float64)
    na = np.zeros(4 * n4)
    na[:N] = 1.0 - inp["isna"][0]
    mave = (a * na).sum(1) / (b * na).sum(1)
    msig = 1.0 / np.sqrt((((a - mave[:, None]) * b * na) ** 2).sum(1) / (nonas - 1))
    rng = np.random.default_rng(4)
    beta = rng.normal(0, 0.02, size=M)
    beta[::3] = 0.0
    g = orc.predict_g(inp["bed"], mask4, mave, msig, beta)
    want_g = (((a - mave[:, None]) * b * na * msig[:, None]) * beta[:, None]).sum(0)
    assert np.allclose(g, want_g, rtol=1e-12, atol=1e-14)
    yk = eps - 0.5 * g
    xtx, xty = orc.assoc(inp["bed"], mask4, yk)
    assert np.array_equal(xtx, ((a * b * na) ** 2).sum(1))
    assert np.allclose(xty, (a * b * na * yk).sum(1), rtol=1e-12, atol=1e-13)
    sigma = float((yk[:N] ** 2).sum() / nonas)
    bt, td, se, pv = orc.mlma_stats(xtx[5], xty[5], sigma)
    assert bt == xty[5] / xtx[5] and td == xty[5] / math.sqrt(sigma * xtx[5]) and se == bt / td
    assert abs(pv - math.erfc(abs(td) / math.sqrt(2.0))) < 1e-15      # 1 - gamma_p(1/2, t^2/2) = erfc(|t|/sqrt 2)
    line = orc.mlma_line("rs12562034", 7, 11, bt, td, se, pv)
    assert len(line) == 123 and line.endswith(b"\n")
    f = line.split()
    assert f[0] == b"rs12562034" and int(f[1]) == 7 and int(f[2]) == 11
    assert [float(x) for x in f[3:]] == pytest.approx([bt, td, se, pv], abs=5e-16)
