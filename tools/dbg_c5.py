import sys, hashlib
sys.path.insert(0, '.')
import numpy as np, gmrm_amd
from oracle import orc
N, M, G = 500_000, 288, 24
for rep in range(2):
    rng = np.random.default_rng(9)
    y = rng.normal(size=N); isna = (rng.random(N) < 0.05).astype(np.uint8)
    eps, mask4, nonas = orc.phen_prepare(y, isna)
    ctx = gmrm_amd.Context(N, M)
    ctx.synth_bed(5, 0.4, 0.05)
    bed = ctx.download_bed()
    print("rep", rep, "bed sha", hashlib.sha1(bed.tobytes()).hexdigest()[:12], "eps sha", hashlib.sha1(eps.tobytes()).hexdigest()[:12])
    ctx.upload_trait(0, eps, mask4, nonas)
    cva = np.tile(np.array([0.0, 0.0001, 0.001, 0.01]), (G, 1)); gi = (np.arange(M) % G).astype(np.int32)
    smp = gmrm_amd.Sampler(ctx, 77, cva, gi)
    ch = orc.Chain(N, bed, eps, mask4, nonas, gi, cva, 77, canon=True)
    for it in (1, 2, 3):
        smp.iterate(it); ch.iterate(it)
        hy = smp.hyper(0)
        print(" it", it, "gpu nan betas", int(np.isnan(ctx.betas(0)).sum()), "orc nan", int(np.isnan(ch.betas).sum()), "sigmag gpu", hy.sigmag[:6], "orc", ch.sigmag[:6], "sigmae", hy.sigmae, ch.sigmae)
    smp.close(); ctx.close()
