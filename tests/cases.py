"""Seeded small cases shared by the CPU (oracle) and GPU (HIP) tests, plus the golden
fixture writer/reader.  Genotypes follow the reference's simulation recipe
(example/data_sim.R:5-41: copies ~ Binomial(2, 0.4), a few causal markers, h2 = 0.5)."""
from dataclasses import dataclass
from pathlib import Path

import numpy as np

GOLD = Path(__file__).resolve().parent / "golden"


@dataclass(frozen=True)
class Case:
    name: str
    N: int
    M: int
    G: int
    K: int
    T: int
    miss: float      # fraction of genotype code 01
    n_na: int        # phenotype NAs per trait
    seed: int
    iters: int
    n_causal: int = 20


CASES = [
    Case("small", 1000, 400, 1, 4, 1, 0.0, 0, 171014, 6),
    Case("ragged", 1003, 300, 3, 4, 2, 0.05, 27, 4242, 5),        # N % 4 != 0, NAs, missing, groups, 2 traits
    Case("groups", 4096, 512, 24, 4, 1, 0.02, 40, 99, 4, 40),
    Case("k3", 777, 200, 2, 3, 1, 0.0, 5, 7, 5),
]
CASE_BY_NAME = {c.name: c for c in CASES}


def im4_of(N):
    return N // 4 if N % 4 == 0 else N // 4 + 1


def make_inputs(case: Case):
    rng = np.random.default_rng(case.seed)
    N, M = case.N, case.M
    copies = rng.binomial(2, 0.4, size=(M, N)).astype(np.int64)
    code = np.where(copies == 2, 0, np.where(copies == 1, 2, 3)).astype(np.uint8)
    if case.miss > 0:
        code[rng.random((M, N)) < case.miss] = 1
    n4 = im4_of(N)
    padded = np.zeros((M, n4 * 4), dtype=np.uint8)
    padded[:, :N] = code
    p = padded.reshape(M, n4, 4)
    bed = (p[:, :, 0] | (p[:, :, 1] << 2) | (p[:, :, 2] << 4) | (p[:, :, 3] << 6)).astype(np.uint8)
    ys, nas = [], []
    for t in range(case.T):
        z = copies.astype(np.float64)
        z = (z - z.mean(axis=1, keepdims=True)) / (z.std(axis=1, keepdims=True) + 1e-12)
        causal = rng.choice(M, size=min(case.n_causal, M), replace=False)
        beta = rng.normal(0.0, np.sqrt(0.5 / len(causal)), size=len(causal))
        g = beta @ z[causal]
        y = g + rng.normal(0.0, np.sqrt(max(1e-6, 1.0 - g.var())), size=N)
        isna = np.zeros(N, dtype=np.uint8)
        if case.n_na:
            isna[rng.choice(N, size=case.n_na, replace=False)] = 1
        ys.append(y)
        nas.append(isna)
    group_index = (np.arange(M) % case.G).astype(np.int32) if case.G > 1 else np.zeros(M, dtype=np.int32)
    base = np.array([0.0, 0.0001, 0.001, 0.01, 0.1, 1.0, 10.0, 100.0])[:case.K]
    cva = np.tile(base, (case.G, 1)) * (1.0 + 0.25 * np.arange(case.G))[:, None]
    return dict(bed=bed, y=np.array(ys), isna=np.array(nas), group_index=group_index, cva=cva)


def prepare_traits(inp):
    from oracle import orc
    out = []
    for t in range(inp["y"].shape[0]):
        out.append(orc.phen_prepare(inp["y"][t], inp["isna"][t]))
    return out


def run_oracle(case: Case, inp, iters=None, canon=True, nranks=1, seed=None, shuffle=True, mimic_hydra=False,
               schedule="sweep", sync_every=0):
    """History of the oracle chain(s): per trait, per iteration comp / betas / hyper-parameters.
    schedule: "sweep" = the build's once-per-sweep exchange (orc_ns_iterate), "steps" = the reference's exchange
    after every marker step (orc_ps_iterate, bayes.cpp:495-553); they coincide for nranks == 1."""
    from oracle import orc
    from gmrm_amd.api import block_of_markers
    iters = case.iters if iters is None else iters
    seed = case.seed if seed is None else seed
    traits = prepare_traits(inp)
    hist = []
    for t, (eps, mask4, nonas) in enumerate(traits):
        chains = []
        for r in range(nranks):
            S, Ml, _ = block_of_markers(case.M, nranks, r)
            chains.append(orc.Chain(case.N, inp["bed"][S:S + Ml], eps, mask4, nonas, inp["group_index"], inp["cva"],
                                    seed, Mt=case.M, S=S, rank=r, shuffle=shuffle, mimic_hydra=mimic_hydra,
                                    canon=canon))
        h = dict(comp=[], betas=[], sigmae=[], sigmag=[], pi=[], mu=[], m0=[], csv=[], eps=None, nupd=[])
        for it in range(1, iters + 1):
            if schedule == "steps":
                orc.ps_iterate(chains, it)
            elif sync_every:                                     # the residual exchange every sync_every markers (orc_nk_iterate)
                orc.nk_iterate(chains, it, sync_every)
            elif nranks == 1:
                chains[0].iterate(it)
            else:
                orc.ns_iterate(chains, it)
            h["comp"].append(np.concatenate([c.comp for c in chains]))
            h["betas"].append(np.concatenate([c.betas for c in chains]))
            h["sigmae"].append(chains[0].sigmae)
            h["sigmag"].append(chains[0].sigmag)
            h["pi"].append(chains[0].pi_est)
            h["mu"].append(chains[0].mu)
            h["m0"].append(chains[0].m0_sum)
            h["csv"].append(chains[0].csv_line(it))
            h["nupd"].append(sum(c.n_updates for c in chains))
        h["eps"] = chains[0].eps
        h["mave"] = np.concatenate([c.mave for c in chains])
        h["msig"] = np.concatenate([c.msig for c in chains])
        hist.append(h)
    return hist


def run_gpu(case: Case, inp, iters=None, seed=None, shuffle=True, mimic_hydra=False, device=0, parts=0):
    """The same history from the HIP path (single shard), through the C ABI.  parts = k: every sweep cut into parts of k
    markers (gmrm_sampler_launch_part: the same chain)."""
    import gmrm_amd
    iters = case.iters if iters is None else iters
    seed = case.seed if seed is None else seed
    traits = prepare_traits(inp)
    ctx = gmrm_amd.Context(case.N, case.M, T=len(traits), device=device)
    ctx.upload_bed(inp["bed"])
    for t, (eps, mask4, nonas) in enumerate(traits):
        ctx.upload_trait(t, eps, mask4, nonas)
    stats = [ctx.compute_markers_statistics(t) for t in range(len(traits))]
    smp = gmrm_amd.Sampler(ctx, seed, inp["cva"], inp["group_index"], shuffle=shuffle, mimic_hydra=mimic_hydra)
    hist = [dict(comp=[], betas=[], sigmae=[], sigmag=[], pi=[], mu=[], m0=[], csv=[], eps=None, nupd=[], nbatch=[], ncross=[], nscrt=[], nscr=[])
            for _ in traits]
    for it in range(1, iters + 1):
        if parts:
            smp.iterate_parts(it, parts)
        else:
            smp.iterate(it)
        for t in range(len(traits)):
            hy = smp.hyper(t)
            h = hist[t]
            h["comp"].append(ctx.comp(t)); h["betas"].append(ctx.betas(t))
            h["sigmae"].append(hy.sigmae); h["sigmag"].append(hy.sigmag); h["pi"].append(hy.pi_est)
            h["mu"].append(hy.mu); h["m0"].append(hy.m0_sum); h["csv"].append(smp.csv_line(t, it))
            h["nupd"].append(hy.n_updates); h["nbatch"].append(hy.n_batches); h["ncross"].append(hy.n_crossed_stops)
            h["nscrt"].append(hy.n_screen_tries); h["nscr"].append(hy.n_screened_passes)
    for t in range(len(traits)):
        hist[t]["eps"] = ctx.get_epsilon(t)
        hist[t]["mave"], hist[t]["msig"] = stats[t]
    smp.close()
    ctx.close()
    return hist


def assert_same_history(a, b, exact=True, rtol=1e-9):
    """Inclusion indices bit-exact; floats bit-exact (exact=True) or within rtol."""
    assert len(a) == len(b)
    for ha, hb in zip(a, b):
        for it, (ca, cb) in enumerate(zip(ha["comp"], hb["comp"])):
            assert np.array_equal(ca, cb), f"component indices differ at iteration {it + 1}"
        for key in ("betas", "sigmag", "pi"):
            for it, (xa, xb) in enumerate(zip(ha[key], hb[key])):
                if exact:
                    assert np.array_equal(xa, xb), f"{key} differ at iteration {it + 1}"
                else:
                    np.testing.assert_allclose(xa, xb, rtol=rtol, atol=1e-300, err_msg=f"{key} it {it + 1}")
        for key in ("sigmae", "mu"):
            if exact:
                assert ha[key] == hb[key], f"{key}: {ha[key]} vs {hb[key]}"
            else:
                np.testing.assert_allclose(ha[key], hb[key], rtol=rtol)
        assert ha["m0"] == hb["m0"]
        if exact:
            assert ha["csv"] == hb["csv"]
            assert np.array_equal(ha["eps"], hb["eps"]), "residuals differ"
        else:
            np.testing.assert_allclose(ha["eps"], hb["eps"], rtol=0, atol=1e-9)


# ---- golden fixtures ------------------------------------------------------------------
def golden_path(name):
    return GOLD / f"chain_{name}.npz"


def write_golden(gold_dir):
    for case in CASES:
        inp = make_inputs(case)
        canon = run_oracle(case, inp, canon=True)
        ref = run_oracle(case, inp, canon=False)
        out = dict(bed=inp["bed"], y=inp["y"], isna=inp["isna"], group_index=inp["group_index"], cva=inp["cva"])
        for t, (hc, hr) in enumerate(zip(canon, ref)):
            out[f"t{t}_comp"] = np.array(hc["comp"], dtype=np.int8)
            out[f"t{t}_betas"] = np.array(hc["betas"])
            out[f"t{t}_sigmae"] = np.array(hc["sigmae"])
            out[f"t{t}_sigmag"] = np.array(hc["sigmag"])
            out[f"t{t}_pi"] = np.array(hc["pi"])
            out[f"t{t}_mu"] = np.array(hc["mu"])
            out[f"t{t}_csv"] = np.frombuffer(b"".join(hc["csv"]), dtype=np.uint8)
            out[f"t{t}_ref_comp"] = np.array(hr["comp"], dtype=np.int8)
            out[f"t{t}_ref_betas"] = np.array(hr["betas"])
            out[f"t{t}_mave"] = hc["mave"]
            out[f"t{t}_msig"] = hc["msig"]
            out[f"t{t}_nupd"] = np.array(hc["nupd"])
        np.savez_compressed(Path(gold_dir) / f"chain_{case.name}.npz", **out)
        print(case.name, "written; updates per sweep:", [int(x) for x in canon[0]["nupd"]])


def write_golden_shards(gold_dir):
    """shards_k3_3ranks.npz: case k3 on 3 ranks, 4 iterations, under both multi-rank schedules of the oracle."""
    case = CASE_BY_NAME["k3"]
    inp = make_inputs(case)
    out = {}
    for sched in ("sweep", "steps"):
        h = run_oracle(case, inp, iters=4, canon=True, nranks=3, schedule=sched)[0]
        out[f"{sched}_comp"] = np.array(h["comp"], dtype=np.int8)
        out[f"{sched}_betas"] = np.array(h["betas"])
        out[f"{sched}_csv"] = np.frombuffer(b"".join(h["csv"]), dtype=np.uint8)
    np.savez_compressed(Path(gold_dir) / "shards_k3_3ranks.npz", **out)


def load_golden(name):
    z = np.load(golden_path(name))
    inp = dict(bed=z["bed"], y=z["y"], isna=z["isna"], group_index=z["group_index"], cva=z["cva"])
    return inp, z
