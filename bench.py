#!/usr/bin/env python3
"""bench.py -- SNP-updates/sec of the Gibbs sweep (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (default "c3"): 500 000 individuals x 1 000 000 SNPs, 1 phenotype, 1 group,
synthetic genotypes generated on the device (copies ~ Binomial(2, 0.4), example/data_sim.R),
phenotype y ~ N(0,1), seed 171014, mixtures 0 / 1e-4 / 1e-3 / 1e-2 (example/test.grm).
Markers are block-partitioned over the N GPUs (the reference's own rule, bayes.cpp:903-925):
total work is fixed, so scaling is "strong".  A step = one full Gibbs sweep (one iteration of
Bayes::process: prologue draws, marker loop on every shard, residual exchange, hyper-parameter
draws).  Genotypes, residual and all chain state are resident in HBM before the timed region.
Defaults follow SURVEY.md 8(d): 2 warm-up sweeps (the chain starts from all-zero effects, so its
first two sweeps update ~10x more markers than any later one), then 5 timed sweeps.

The JSON line carries, besides the driver's contract fields:
  roofline     dominant kernel = the persistent sweep kernel; achieved = algorithmic bytes
               (ceil(N/4) per SNP-update x markers per launch) / average launch duration
               measured with HIP events on the launch stream inside this run.
  cpu_baseline the oracle's reference-order OpenMP path (a port of the reference's own
               vectorised loops, built -Ofast -march=native -fopenmp on this host) timed on
               a bounded sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WORKLOADS = {
    # name: (N individuals, Mt markers, T traits, G groups, phenotype-NA rate, genotype-missing rate)
    "c2": (50_000, 100_000, 1, 1, 0.0, 0.0),
    "c3": (500_000, 1_000_000, 1, 1, 0.0, 0.0),
    "c4": (500_000, 1_000_000, 4, 1, 0.0, 0.0),
    "c5": (500_000, 1_000_000, 1, 24, 0.05, 0.05),
    # not a BASELINE configuration: c3 with missing genotypes (0.1 % of the calls) in 1 % of the markers only -- what the
    # per-batch choice of the exchange layout is for; compare its rate with c3's
    "c6": (500_000, 1_000_000, 1, 1, 0.0, 0.001),
    # not a BASELINE configuration: c3 with linkage disequilibrium -- markers in blocks of 20 whose neighbours correlate with
    # r ~ 0.9 (gmrm_synth_bed_ld) -- and a phenotype WITH signal (0.1 % causal markers, h2 = 0.5): correlated markers share
    # the signal, so more of them sit in the model and more visits change an effect than with the independent draws of
    # example/data_sim.R; the headline's sensitivity to the update rate, on record
    "ld": (500_000, 1_000_000, 1, 1, 0.0, 0.0),
}
LD_BLOCKS = {"ld": (20, 0.9)}          # workload -> (block length, probability that a haplotype copies its neighbour's allele)
DIRTY_MARKER_FRACTION = {"c6": 0.01}   # workloads whose missing genotypes sit in this fraction of the markers only
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
CPU_THREADS = 1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--markers", type=int, default=0, help="override the total marker count (debug)")
    ap.add_argument("--individuals", type=int, default=0, help="override N (debug)")
    ap.add_argument("--traits", type=int, default=0, help="override the number of phenotypes (debug)")
    ap.add_argument("--seed", type=int, default=171014)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time")
    ap.add_argument("--no-signal", action="store_true", help="skip the signal-bearing extra block")
    ap.add_argument("--signal-sweeps", type=int, default=8, help="sweeps of the signal-bearing chain (first 2 reported apart)")
    ap.add_argument("--sync-every", type=int, default=0, help="residual exchange every k markers of a shard's block instead of once per "
                                                              "sweep (the sweep runs as ceil(M/k) kernel launches); 0 = once per sweep")
    return ap.parse_args()


def cpu_baseline(ctx, eps, mask4, nonas, cva, seed, target_s):
    """The reference-order OpenMP port on a bounded sample: the first `ms` markers of this
    GPU's block (downloaded from the device), the same phenotype, 1 warm-up + timed sweeps."""
    import numpy as np
    from oracle import orc
    threads = CPU_THREADS
    orc.lib_fast_native()
    N = ctx.N

    def run(ms, sweeps):
        bed = ctx.download_bed(0, ms)
        ch = orc.Chain(N, bed, eps, mask4, nonas, np.zeros(ms, dtype=np.int32), cva, seed, canon=False, fast=True)
        ch.iterate(1)                       # warm-up sweep (first touch, sigmae initialisation)
        t0 = time.perf_counter()
        for it in range(2, 2 + sweeps):
            ch.iterate(it)
        return (time.perf_counter() - t0) / sweeps

    probe_ms = min(ctx.M, 256)
    t_probe = run(probe_ms, 1)
    per_marker = t_probe / probe_ms
    ms = int(min(ctx.M, max(probe_ms, target_s / 3.0 / max(per_marker, 1e-9))))
    ms = min(ms, 40_000)
    t_sweep = run(ms, 2)
    out = {"value": ms / t_sweep, "unit": "SNP-updates/s", "cores": threads, "kind": "port",
           "sample": f"first {ms} markers of the workload x 2 timed sweeps after 1 warm-up, N={N}, "
                     f"reference-order loops (bayes.cpp:749-766, phenotype.cpp:375-390) with OpenMP, "
                     f"-Ofast -march=native, OMP_NUM_THREADS={threads}"}
    # SURVEY 8(d) also asks for the one-thread figure: same loops, OMP_NUM_THREADS = 1, a ~4 s sample
    try:
        import ctypes
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(1)
        ms1 = max(64, min(ms, int(ms / max(1, threads) / 2)))
        t1 = run(ms1, 1)
        omp.omp_set_num_threads(threads)
        out["value_1_thread"] = ms1 / t1
        out["sample_1_thread"] = f"first {ms1} markers x 1 timed sweep after 1 warm-up"
    except Exception as e:                                   # never blocks the headline numbers
        out["value_1_thread"] = None
        out["sample_1_thread"] = f"failed: {e!r}"
    return out


def signal_block(ctx, gmrm_amd, np, N, Mt, cva, group_index, seed, sweeps):
    """The same genotypes with a phenotype that carries signal (example/data_sim.R:18-41): 0.5 % of the
    markers causal, effects ~ N(0, h2 / n_causal) on standardised genotypes, h2 = 0.5, y = g + e.  The
    genetic values g come from the device (gmrm_predict_g, the product's own kernel).  Reported beside
    the headline (null) workload: with signal more visits change an effect, and every change costs one
    exchange round, so throughput moves with the update rate."""
    rng = np.random.default_rng(seed + 1)
    n_causal = max(1, Mt // 200)
    beta = np.zeros(ctx.M)
    idx = rng.choice(ctx.M, size=min(n_causal, ctx.M), replace=False)
    beta[idx] = rng.normal(0.0, np.sqrt(0.5 / n_causal), size=len(idx))
    g = ctx.predict_g(0, beta)
    vg = float(np.var(g))
    y = g + rng.normal(0.0, np.sqrt(max(1e-6, 1.0 - vg)), size=N)
    eps, mask4, nonas = gmrm_amd.prepare_phenotype(y, np.zeros(N, dtype=np.uint8))
    ctx.upload_trait(0, eps, mask4, nonas)
    ctx.compute_markers_statistics(0)
    smp = gmrm_amd.Sampler(ctx, seed, cva, group_index)
    ms, upd, rounds = [], [], []
    t0 = time.perf_counter()
    for it in range(1, sweeps + 1):
        smp.iterate(it)
        hy = smp.hyper(0)
        ms.append(hy.sweep_device_ms); upd.append(hy.n_updates); rounds.append(hy.n_batches)
    wall = time.perf_counter() - t0
    hy = smp.hyper(0)
    smp.close()
    tail = ms[2:] if len(ms) > 2 else ms
    return {"workload": f"same genotypes, y = X beta + e: {len(idx)} causal markers (0.5 %), h2 = 0.5 (var(g) = {vg:.3f}), "
                        f"{sweeps} sweeps from an all-zero start",
            "value_after_2_sweeps": ctx.M / (sum(tail) / len(tail) / 1e3), "unit": "SNP-updates/s (kernel time)",
            "kernel_ms_per_sweep": ms, "updates_per_sweep": upd, "sync_rounds_per_sweep": rounds,
            "update_fraction": [u / float(ctx.M) for u in upd], "first_two_sweeps_ms": ms[:2], "wall_s": wall,
            "sigmaG_final": float(np.sum(hy.sigmag)), "sigmaE_final": float(hy.sigmae)}


def launcher_command(ngpus, argv, port):
    """The command `python bench.py --gpus N ...` re-launches itself with (one rank per GPU, RCCL)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]


def self_launch(ngpus):
    import socket
    import subprocess
    with socket.socket() as sk:                      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = launcher_command(ngpus, sys.argv[1:], port)
    print("[bench] launching:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver calls it: start one rank per GPU under
        # torch.distributed.run as a CHILD process and relay its output and exit code.  Nothing in
        # this process has imported torch or touched the GPU yet (never exec from a process that has).
        raise SystemExit(self_launch(a.gpus))
    # CPU-baseline threads: the GPU box's CPU share for one GPU is 16 hardware threads; fixed
    # here, before any OpenMP runtime starts (no OMP_PROC_BIND: it would pin this thread too).
    global CPU_THREADS
    CPU_THREADS = min(len(os.sched_getaffinity(0)), int(os.environ.get("GMRM_CPU_THREADS", "16")))
    os.environ["OMP_NUM_THREADS"] = str(CPU_THREADS)

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before HIP initialises: four phenotype chains on four hardware queues
    import numpy as np
    import torch
    import torch.distributed as dist
    import gmrm_amd
    from gmrm_amd.dist import HipEngine, ShardedDriver

    lib = gmrm_amd.load_library()
    if not torch.cuda.is_available() or lib.gmrm_device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_pg = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ     # also under torchrun with 1 rank
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    N, Mt, T, G, na_rate, miss = WORKLOADS[a.workload]
    if a.markers:
        Mt = a.markers
    if a.individuals:
        N = a.individuals
    if a.traits:
        T = a.traits
    S, M, _ = gmrm_amd.block_of_markers(Mt, world, rank)

    t_setup = time.perf_counter()
    ctx = gmrm_amd.Context(N, M, Mt=Mt, S=S, T=T, device=local)
    dirty_frac = DIRTY_MARKER_FRACTION.get(a.workload, 0.0)
    ld_block, ld_keep = LD_BLOCKS.get(a.workload, (0, 0.0))
    ctx.synth_bed(a.seed, 0.4, 0.0 if dirty_frac else miss, ld_block, ld_keep)
    rng = np.random.default_rng(a.seed)
    if dirty_frac:                                               # code 01 (missing) in `miss` of the calls of the first markers
        nd = max(1, int(M * dirty_frac))
        cols = ctx.download_bed(0, nd)
        per = max(1, int(N * miss))
        who = rng.integers(0, N, size=(nd, per))
        rows = np.repeat(np.arange(nd), per)
        byte, sh = (who // 4).ravel(), (2 * (who % 4)).ravel().astype(np.uint8)
        cols[rows, byte] = (cols[rows, byte] & ~(np.uint8(3) << sh)) | (np.uint8(1) << sh)
        ctx.upload_bed(cols, 0)
        del cols, who, rows, byte, sh
    traits = []
    pheno_note = "y~N(0,1)"
    for t in range(T):
        y = rng.normal(size=N)
        isna = (rng.random(N) < na_rate).astype(np.uint8) if na_rate > 0 else np.zeros(N, dtype=np.uint8)
        if a.workload in LD_BLOCKS and world == 1:
            # a phenotype WITH signal on the correlated markers: 0.1 % causal, h2 = 0.5, g from the device (gmrm_predict_g needs
            # the marker statistics of SOME phenotype on these individuals: the null one first)
            eps0, mask0, nonas0 = gmrm_amd.prepare_phenotype(y, isna)
            ctx.upload_trait(t, eps0, mask0, nonas0)
            ctx.compute_markers_statistics(t)
            n_causal = max(1, Mt // 1000)
            beta = np.zeros(ctx.M)
            idx = rng.choice(ctx.M, size=min(n_causal, ctx.M), replace=False)
            beta[idx] = rng.normal(0.0, np.sqrt(0.5 / n_causal), size=len(idx))
            g = ctx.predict_g(t, beta)
            y = g + rng.normal(0.0, np.sqrt(max(1e-6, 1.0 - float(np.var(g)))), size=N)
            pheno_note = f"y = X beta + e, {len(idx)} causal markers (0.1 %), h2 = 0.5 (var(g) = {float(np.var(g)):.3f})"
        eps, mask4, nonas = gmrm_amd.prepare_phenotype(y, isna)
        ctx.upload_trait(t, eps, mask4, nonas)
        traits.append((eps, mask4, nonas))
    t_stats = time.perf_counter()
    for t in range(T):
        ctx.compute_markers_statistics(t)
    t_stats = time.perf_counter() - t_stats
    base = np.array([0.0, 0.0001, 0.001, 0.01])                 # example/test.grm
    cva = np.tile(base, (G, 1))
    group_index = (np.arange(Mt) % G).astype(np.int32)
    smp = gmrm_amd.Sampler(ctx, a.seed, cva, group_index, rank=rank, nranks=world)
    # GMRM_BENCH_FORCE_EXCHANGE=1: the residual exchange also with one rank (an identity) -- its host + device cost on a one-GPU box
    driver = ShardedDriver(HipEngine(smp, dev), force_exchange=bool(os.environ.get("GMRM_BENCH_FORCE_EXCHANGE"))) if use_pg else None
    t_setup = time.perf_counter() - t_setup

    def step(it):
        if driver is not None:
            driver.iterate(it, sync_every=a.sync_every)
        else:
            smp.iterate_parts(it, a.sync_every) if a.sync_every > 0 else smp.iterate(it)

    def fence():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()
        lib.gmrm_ctx_sync(ctx.h)

    it = 0
    warm_ms = []
    for _ in range(a.warmup):
        it += 1
        step(it)
        warm_ms.append(max(smp.hyper(t).sweep_device_ms for t in range(T)))
    fence()
    t0 = time.perf_counter()
    kern_ms, upd, batches, planned, stale, crossed = [], [], [], [], [], []
    for _ in range(a.steps):
        it += 1
        step(it)
        hy = smp.hyper(0)
        kern_ms.append(max(smp.hyper(t).sweep_device_ms for t in range(T)))
        upd.append(hy.n_updates)
        batches.append(hy.n_batches)
        planned.append(hy.n_planned_stops)
        stale.append(hy.n_stale_dots)
        crossed.append(hy.n_crossed_stops)
    fence()
    dt = time.perf_counter() - t0
    per_gpu_kernel_ms = [sum(kern_ms) / max(1, len(kern_ms))]
    if use_pg:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        km = torch.tensor(per_gpu_kernel_ms, dtype=torch.float64, device=dev)
        allk = [torch.zeros_like(km) for _ in range(world)]
        dist.all_gather(allk, km)
        per_gpu_kernel_ms = [float(x.item()) for x in allk]

    if rank == 0:
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (it cannot
        # be read inside this process); the committed summary of the same workload is quoted.
        traffic = None
        mfma_busy = None
        pmc = ROOT / "profiles" / "r04_pmc_summary.json"
        sq = ROOT / "profiles" / "r04_pmc_sq_summary.json"
        if a.workload == "c3" and world == 1 and not a.markers and not a.individuals:
            try:
                traffic = json.loads(pmc.read_text())["traffic_bytes_per_launch_k_sweep"]
            except Exception:
                traffic = None
            try:
                mfma_busy = json.loads(sq.read_text())["c3"]["ratios"]["mfma_busy_cycles_over_busy_cu_cycles"]
            except Exception:
                mfma_busy = None
        mbytes = ctx.mbytes
        value = Mt * T * a.steps / dt
        avg_kernel_s = (sum(kern_ms) / len(kern_ms)) / 1e3
        alg_bytes = float(M) * mbytes                          # ceil(N/4) bytes per SNP-update x markers per launch
        achieved = alg_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        out = {
            "metric": "SNP-updates/sec (Gibbs sweep)", "value": value, "unit": "SNP-updates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {N} individuals x {Mt} SNPs, {T} phenotype(s), {G} group(s), "
                                   f"K=4, genotypes Binomial(2,0.4) generated on device"
                                   + (f" in LD blocks of {ld_block} markers (neighbours r~{ld_keep})" if ld_block > 1 else "")
                                   + f", {pheno_note}, seed {a.seed}",
                       "markers_per_gpu": M, "parallelism": f"marker-shard x{world}, 1 residual all-reduce/sweep" if a.sync_every <= 0 else
                                                            f"marker-shard x{world}, 1 residual all-reduce every {a.sync_every} markers",
                       "phenotype_na_rate": na_rate, "genotype_missing_rate": miss,
                       "markers_with_missing_genotypes": dirty_frac if dirty_frac else (1.0 if miss > 0 else 0.0)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": "bytes per launch: 2 x FETCH_SIZE (the gfx950 correction for 16-B/lane loads) + WRITE_SIZE from "
                                         "profiles/r04_pmc_summary.json (separate rocprofv3 --pmc passes of this workload, stationary sweeps)" if traffic else None,
                         "mfma_busy_frac": mfma_busy,
                         "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES of the sweep kernel from profiles/r04_pmc_sq_summary.json "
                                           "(rocprofv3 --pmc pass of this workload): the matrix cores take the dots, but the kernel waits ~half of "
                                           "its wave-cycles (SQ_WAIT_ANY) on the grid-wide exchange" if mfma_busy is not None else None,
                         "bytes_per_unit": mbytes,
                         "bytes_per_unit_note": "algorithmic bytes per SNP-update = ceil(N/4): the marker's 2-bit genotype column, read once; the residual "
                                                "and every table stay on chip for the whole sweep (DESIGN.md 5)",
                         "kernel": "gm::k_sweep (persistent marker loop)",
                         "kernel_ms_avg": avg_kernel_s * 1e3,
                         "kernel_ms_per_launch": kern_ms, "kernel_ms_warmup_launches": warm_ms,
                         "kernel_ms_avg_all_launches": (sum(kern_ms) + sum(warm_ms)) / max(1, len(kern_ms) + len(warm_ms)),
                         "kernel_ms_median": sorted(kern_ms)[len(kern_ms) // 2],
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "measured_stream_read_GBps": (M * ctx.mbytes * T) / t_stats / 1e9 if t_stats > 0 else None,
                         "measured_stream_note": "gm::k_marker_stats over the same genotype block (one pass, 16-B loads): "
                                                 "what a pure read stream of this data reaches on this GPU"},
            "sweep": {"updates_per_sweep": upd, "sync_rounds_per_sweep": batches,
                      "update_fraction": [u / float(M) for u in upd],
                      "planned_stops_per_sweep": planned, "stale_dots_per_sweep": stale, "crossed_stops_per_sweep": crossed,
                      "fast_layout_batches_last_sweep": smp.hyper(0).n_fast_batches,
                      "screen_tries_last_sweep": smp.hyper(0).n_screen_tries, "screened_passes_last_sweep": smp.hyper(0).n_screened_passes,
                      "note": "a round = one batch of dots + one grid-wide exchange.  It ends at the first marker whose effect changes, "
                              "unless that marker was in the model before the visit and the walk can cross it (crossed stops: the "
                              "sums behind it are patched exactly inside the round); planned stops: rounds that ended at a marker "
                              "that was in the model (known in advance, the batch ends there); stale dots: computed behind a stop "
                              "and thrown away"},
            "rccl_ranks": world if use_pg else 0,
            "collectives_per_sweep": (driver.collectives / float(a.steps + a.warmup)) if driver is not None else 0,
            "exchange_forced_with_one_rank": bool(os.environ.get("GMRM_BENCH_FORCE_EXCHANGE")) if driver is not None else False,
            "per_gpu": {"kernel_ms_avg": per_gpu_kernel_ms,
                        "roofline_frac": [(float(gmrm_amd.block_of_markers(Mt, world, r)[1]) * mbytes / (k / 1e3) / 1e9 / HBM_PEAK_GBS) if k > 0 else 0.0
                                          for r, k in enumerate(per_gpu_kernel_ms)],
                        "note": "every shard sweeps its block against its own residual replica; one exact residual "
                                "all-reduce per sweep (RCCL) -- a sweep-synchronous approximation of the sequential scan, "
                                "not the reference's per-step exchange (DESIGN.md section 6)" if world > 1 else None},
            "setup_s": t_setup, "marker_stats_s": t_stats,
            "marker_stats_GBps": (M * ctx.mbytes * T) / t_stats / 1e9 if t_stats > 0 else None,
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                eps, mask4, nonas = traits[0]
                out["cpu_baseline"] = cpu_baseline(ctx, eps, mask4, nonas, cva[:1], a.seed, a.cpu_seconds)
                out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            except Exception as e:                               # the baseline never blocks the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "SNP-updates/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        if world == 1 and T == 1 and not a.no_signal and a.signal_sweeps > 0:
            try:
                smp.close()
                out["extra"] = {"signal_bearing": signal_block(ctx, gmrm_amd, np, N, Mt, cva, group_index, a.seed, a.signal_sweeps)}
            except Exception as e:                               # never blocks the headline numbers
                out["extra"] = {"signal_bearing": {"failed": repr(e)}}
        print(json.dumps(out), flush=True)
    smp.close()
    ctx.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
