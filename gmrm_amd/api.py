"""Host-side mirror of the reference's call surface for the hot path.

`Context` plays the part of class Bayes after setup_processing() (reference
src/bayes.hpp:84-105, src/bayes.cpp:774-812) for one GPU and one block of markers;
its methods are the reference's member functions on the path, same names and argument
meaning (dot_product, update_epsilon, offset_epsilon, epsilon_sumsqr,
update_epsilon_sigma, compute_markers_statistics).  `Sampler` is Bayes::process()
(src/bayes.cpp:318-677).  Everything computes on the GPU through libgmrm_hip.so.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import check, c_double_p, c_int_p, c_u8_p


def im4_of(N: int) -> int:
    """ceil(N/4): bytes per marker column (src/bayes.cpp:776, src/phenotype.cpp:22)."""
    return N // 4 if N % 4 == 0 else N // 4 + 1


def block_of_markers(Mt: int, nranks: int, rank: int):
    """Bayes::set_block_of_markers (src/bayes.cpp:903-925): (S, M, Mm) of `rank`."""
    modu, size = Mt % nranks, Mt // nranks
    Mm = size + 1 if modu != 0 else size
    lens = [size + 1 if i < modu else size for i in range(nranks)]
    S = sum(lens[:rank])
    return S, lens[rank], Mm


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _bp(a):
    return a.ctypes.data_as(c_u8_p)


def prepare_phenotype(y, isna):
    """Phenotype::read_file (src/phenotype.cpp:587-673) -> (eps[4*ceil(N/4)], mask4, nonas)."""
    lib = _lib.load_library()
    y = np.ascontiguousarray(y, dtype=np.float64)
    isna = np.ascontiguousarray(isna, dtype=np.uint8)
    N = y.shape[0]
    n4 = im4_of(N)
    eps = np.zeros(4 * n4, dtype=np.float64)
    mask4 = np.zeros(n4, dtype=np.uint8)
    nonas = C.c_int(0)
    check(lib.gmrm_phen_prepare(_dp(y), _bp(isna), N, _dp(eps), _bp(mask4), C.byref(nonas)))
    return eps, mask4, nonas.value


@dataclass
class Hyper:
    sigmae: float
    mu: float
    m0_sum: int
    sigmag: np.ndarray
    pi_est: np.ndarray
    n_updates: int
    n_batches: int
    sweep_device_ms: float
    n_planned_stops: int = 0
    n_stale_dots: int = 0
    n_fast_batches: int = 0
    n_crossed_stops: int = 0
    n_screen_tries: int = 0
    n_screened_passes: int = 0


class Context:
    def __init__(self, N, M, Mt=None, S=0, T=1, device=0):
        self.lib = _lib.load_library()
        self.N, self.M = int(N), int(M)
        self.Mt = int(Mt if Mt is not None else M)
        self.S, self.T, self.device = int(S), int(T), int(device)
        self.mbytes = im4_of(self.N)
        self.h = C.c_void_p()
        check(self.lib.gmrm_ctx_create(C.byref(self.h), self.device, self.N, self.M, self.Mt, self.S, self.T))

    def close(self):
        if getattr(self, "h", None):
            self.lib.gmrm_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- genotypes: Bayes::load_genotype (src/bayes.cpp:867-900) ----
    def geometry(self):
        """Launch geometry of the persistent sweep kernel (R bytes per thread, W workgroups, conc chains side by side)."""
        g = _lib.GeometryC()
        check(self.lib.gmrm_ctx_geometry(self.h, C.byref(g)))
        return dict(R=g.R, W=g.W, conc=g.conc, num_cu=g.num_cu, max_resident_wg=g.max_resident_wg, hw_queues=g.hw_queues)

    def upload_bed(self, cols, first=0):
        cols = np.ascontiguousarray(cols, dtype=np.uint8)
        if cols.ndim != 2 or cols.shape[1] != self.mbytes:
            raise ValueError(f"bed block must be [n_markers, {self.mbytes}] bytes")
        check(self.lib.gmrm_upload_bed(self.h, _bp(cols), int(first), int(cols.shape[0])))

    def load_bed_file(self, path, file_first_marker=None, threads=8):
        """Stream this context's marker block from a PLINK .bed file (validated magic + size) through
        pinned buffers; returns a dict with bytes, seconds, GB/s."""
        from ._lib import IngestStatsC
        st = IngestStatsC()
        first = self.S if file_first_marker is None else int(file_first_marker)
        check(self.lib.gmrm_load_bed_file(self.h, str(path).encode(), first, int(threads), C.byref(st)))
        return {"bytes": st.bytes, "seconds": st.seconds, "read_seconds": st.read_seconds, "threads": st.threads,
                "chunk_bytes": st.chunk_bytes, "GBps": st.bytes / st.seconds / 1e9 if st.seconds > 0 else None}

    # ---- Bayes::predict building blocks (src/bayes.cpp:16-284) ----
    def predict_g(self, t, beta_local):
        beta_local = np.ascontiguousarray(beta_local, dtype=np.float64)
        if beta_local.shape != (self.M,):
            raise ValueError("beta_local must hold one value per local marker")
        g = np.empty(self.N, dtype=np.float64)
        check(self.lib.gmrm_predict_g(self.h, int(t), _dp(beta_local), _dp(g)))
        return g

    def assoc(self, t, yk=None):
        xtx = np.empty(self.M, dtype=np.float64)
        xty = np.empty(self.M, dtype=np.float64)
        if yk is not None:
            yk = np.ascontiguousarray(yk, dtype=np.float64)
            if yk.shape != (self.N,):
                raise ValueError("yk must hold one value per individual")
        check(self.lib.gmrm_assoc(self.h, int(t), _dp(yk) if yk is not None else None, _dp(xtx), _dp(xty)))
        return xtx, xty

    def download_bed(self, first=0, n=None):
        n = self.M - first if n is None else n
        out = np.empty((n, self.mbytes), dtype=np.uint8)
        check(self.lib.gmrm_download_bed(self.h, _bp(out), int(first), int(n)))
        return out

    def synth_bed(self, seed, maf=0.4, miss_rate=0.0, ld_block=0, ld_keep=0.0):
        """Synthetic genotypes on the device; ld_block > 1: markers correlated in blocks (gmrm_synth_bed_ld)."""
        check(self.lib.gmrm_synth_bed_ld(self.h, int(seed), float(maf), float(miss_rate), int(ld_block), float(ld_keep)))

    # ---- phenotypes: Phenotype ctor + read_file (src/phenotype.cpp:18-55,587-673) ----
    def upload_trait(self, t, eps, mask4, nonas):
        eps = np.ascontiguousarray(eps, dtype=np.float64)
        mask4 = np.ascontiguousarray(mask4, dtype=np.uint8)
        if eps.shape[0] != 4 * self.mbytes or mask4.shape[0] != self.mbytes:
            raise ValueError("eps must hold 4*ceil(N/4) doubles and mask4 ceil(N/4) bytes")
        check(self.lib.gmrm_upload_trait(self.h, int(t), _dp(eps), _bp(mask4), int(nonas)))

    def get_epsilon(self, t=0):
        out = np.empty(4 * self.mbytes, dtype=np.float64)
        check(self.lib.gmrm_download_eps(self.h, int(t), _dp(out)))
        return out

    def set_epsilon(self, t, eps):
        eps = np.ascontiguousarray(eps, dtype=np.float64)
        check(self.lib.gmrm_upload_eps(self.h, int(t), _dp(eps)))

    # ---- the reference's per-call kernels ----
    def compute_markers_statistics(self, t=0):
        """PhenMgr::compute_markers_statistics (src/phenotype.cpp:466-556) -> (mave, msig)."""
        check(self.lib.gmrm_marker_stats(self.h, int(t)))
        mave = np.empty(self.M, dtype=np.float64)
        msig = np.empty(self.M, dtype=np.float64)
        check(self.lib.gmrm_get_marker_stats(self.h, int(t), _dp(mave), _dp(msig)))
        return mave, msig

    def set_markers_statistics(self, t, mave, msig):
        mave = np.ascontiguousarray(mave, dtype=np.float64)
        msig = np.ascontiguousarray(msig, dtype=np.float64)
        check(self.lib.gmrm_set_marker_stats(self.h, int(t), _dp(mave), _dp(msig)))

    def dot_product(self, mloc, mu, sigma_inv, t=0):
        """Bayes::dot_product(mloc, phen, mu, sigma_inv) (src/bayes.cpp:709-770)."""
        out = C.c_double(0.0)
        check(self.lib.gmrm_dot(self.h, int(t), int(mloc), float(mu), float(sigma_inv), C.byref(out)))
        return out.value

    def update_epsilon(self, dbeta3, mloc, t=0):
        """Phenotype::update_epsilon(dbeta[3], bed column) (src/phenotype.cpp:326-393)."""
        d = np.ascontiguousarray(dbeta3, dtype=np.float64)
        check(self.lib.gmrm_update_eps(self.h, int(t), int(mloc), _dp(d)))

    def update_epsilon_from(self, src, dbeta3, mloc, t=0):
        """Bayes::update_epsilon for one sender (src/bayes.cpp:681-706): marker `mloc` of context `src`."""
        d = np.ascontiguousarray(dbeta3, dtype=np.float64)
        check(self.lib.gmrm_update_eps_from(self.h, int(t), src.h, int(mloc), _dp(d)))

    def offset_epsilon(self, offset, t=0):
        """Phenotype::offset_epsilon (src/phenotype.cpp:395-411)."""
        check(self.lib.gmrm_offset_eps(self.h, int(t), float(offset)))

    def epsilon_sumsqr(self, t=0):
        """Phenotype::epsilon_sumsqr (src/phenotype.cpp:251-261)."""
        out = C.c_double(0.0)
        check(self.lib.gmrm_sumsqr(self.h, int(t), C.byref(out)))
        return out.value

    def update_epsilon_sigma(self, t=0):
        """Phenotype::update_epsilon_sigma (src/phenotype.cpp:432-459) -> sigmae."""
        out = C.c_double(0.0)
        check(self.lib.gmrm_eps_sigma(self.h, int(t), C.byref(out)))
        return out.value

    # ---- per-marker chain state ----
    def betas(self, t=0):
        out = np.empty(self.M, dtype=np.float64)
        check(self.lib.gmrm_get_betas(self.h, int(t), _dp(out)))
        return out

    def comp(self, t=0):
        out = np.empty(self.M, dtype=np.int32)
        check(self.lib.gmrm_get_comp(self.h, int(t), _ip(out)))
        return out

    def acum(self, t=0):
        out = np.empty(self.M, dtype=np.float64)
        check(self.lib.gmrm_get_acum(self.h, int(t), _dp(out)))
        return out

    def set_groups(self, group_local):
        g = np.ascontiguousarray(group_local, dtype=np.int32)
        check(self.lib.gmrm_set_groups(self.h, _ip(g)))

    # ---- fused marker loop (one persistent launch) ----
    def sweep(self, t, order, sigmag, pi_est, cva, sigmae, rng_state, rng_index):
        """Marker loop of Bayes::process for phenotype t (src/bayes.cpp:375-553).
        Returns (cass[G*K], rng_state, rng_index, n_updates, n_batches, device_ms)."""
        self.sweep_launch(t, order, sigmag, pi_est, cva, sigmae, rng_state, rng_index)
        return self.sweep_finish(t)

    def sweep_launch(self, t, order, sigmag, pi_est, cva, sigmae, rng_state, rng_index):
        cva = np.ascontiguousarray(cva, dtype=np.float64)
        G, K = cva.shape
        self._sw = dict(order=np.ascontiguousarray(order, dtype=np.int32),
                        sigmag=np.ascontiguousarray(sigmag, dtype=np.float64),
                        pi=np.ascontiguousarray(pi_est, dtype=np.float64).reshape(-1), cva=cva.reshape(-1))
        si = _lib.SweepIn()
        si.G, si.K = G, K
        si.order, si.sigmag = _ip(self._sw["order"]), _dp(self._sw["sigmag"])
        si.pi_est, si.cva = _dp(self._sw["pi"]), _dp(self._sw["cva"])
        si.sigmae = float(sigmae)
        st = np.ascontiguousarray(rng_state, dtype=np.uint32)
        C.memmove(si.rng_state, st.ctypes.data, 624 * 4)
        si.rng_index = int(rng_index)
        self._gk = (G, K)
        check(self.lib.gmrm_sweep_launch(self.h, int(t), C.byref(si)))

    def sweep_finish(self, t):
        G, K = self._gk
        cass = np.zeros(G * K, dtype=np.int32)
        so = _lib.SweepOut()
        so.cass = _ip(cass)
        check(self.lib.gmrm_sweep_finish(self.h, int(t), C.byref(so)))
        state = np.frombuffer(bytes(so.rng_state), dtype=np.uint32).copy()
        return cass, state, so.rng_index, so.n_updates, so.n_batches, so.device_ms

    # ---- multi-GPU residual exchange ----
    def eps_snapshot(self, t=0):
        check(self.lib.gmrm_eps_snapshot(self.h, int(t)))

    def eps_delta_export(self, t, dev_ptr):
        check(self.lib.gmrm_eps_delta_export(self.h, int(t), C.c_void_p(int(dev_ptr))))

    def eps_delta_import(self, t, dev_ptr):
        check(self.lib.gmrm_eps_delta_import(self.h, int(t), C.c_void_p(int(dev_ptr))))


class Sampler:
    """Bayes::process() (src/bayes.cpp:318-677) for one marker shard, all T phenotypes."""

    def __init__(self, ctx: Context, seed, cva, group_index, rank=0, nranks=1, shuffle=True, mimic_hydra=False):
        self.ctx = ctx
        self.lib = ctx.lib
        self.cva = np.ascontiguousarray(cva, dtype=np.float64)
        self.G, self.K = self.cva.shape
        self.group_index = np.ascontiguousarray(group_index, dtype=np.int32)
        if self.group_index.shape[0] != ctx.Mt:
            raise ValueError("group_index must cover all Mt markers")
        self.rank, self.nranks = int(rank), int(nranks)
        o = _lib.SamplerOpts()
        o.seed, o.rank, o.nranks = int(seed), self.rank, self.nranks
        o.shuffle, o.mimic_hydra = int(bool(shuffle)), int(bool(mimic_hydra))
        o.G, o.K = self.G, self.K
        o.cva, o.group_index = _dp(self.cva), _ip(self.group_index)
        self.h = C.c_void_p()
        check(self.lib.gmrm_sampler_create(C.byref(self.h), ctx.h, C.byref(o)))
        check(self.lib.gmrm_sampler_init(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.gmrm_sampler_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def iterate(self, it):
        check(self.lib.gmrm_sampler_iterate(self.h, int(it)))

    # the same iteration cut at its exchange points (multi-GPU; gmrm_amd/dist.py)
    def draw_mu(self, it):
        mu = np.zeros(self.ctx.T, dtype=np.float64)
        check(self.lib.gmrm_sampler_draw_mu(self.h, int(it), _dp(mu)))
        return mu

    def begin_sweep(self, mu):
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        check(self.lib.gmrm_sampler_begin_sweep(self.h, _dp(mu)))

    # a sweep in parts (--sync-every k: marker shards exchange their residuals every k markers; include/gmrm_hip.h)
    def begin_parts(self, mu):
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        check(self.lib.gmrm_sampler_begin_parts(self.h, _dp(mu)))

    def launch_part(self, first, count):
        check(self.lib.gmrm_sampler_launch_part(self.h, int(first), int(count)))

    def finish_part(self):
        check(self.lib.gmrm_sampler_finish_part(self.h))

    def preshuffle(self):
        """The NEXT iteration's marker shuffle, on the idle host while a part is in flight (gmrm_sampler_preshuffle)."""
        check(self.lib.gmrm_sampler_preshuffle(self.h))

    def iterate_parts(self, it, k):
        """One iteration of a single shard with the sweep cut into parts of k markers: the same chain as iterate()."""
        mu = self.draw_mu(it)
        self.begin_parts(mu)
        M = self.ctx.M
        first = 0
        while True:
            n = min(int(k), M - first)
            self.launch_part(first, n)
            if first == 0:
                self.preshuffle()
            self.finish_part()
            first += n
            if first >= M:
                break
        cass, bsq = self.end_sweep()
        self.epilogue(cass, bsq)

    def end_sweep(self):
        T = self.ctx.T
        cass = np.zeros((T, self.G, self.K), dtype=np.int32)
        bsq = np.zeros((T, self.G), dtype=np.float64)
        check(self.lib.gmrm_sampler_end_sweep(self.h, _ip(cass), _dp(bsq)))
        return cass, bsq

    # the reference's per-step schedule (src/bayes.cpp:374-553 with several MPI tasks; see include/gmrm_hip.h)
    def begin_steps(self, mu):
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        check(self.lib.gmrm_sampler_begin_steps(self.h, _dp(mu)))

    def step(self, mrki):
        """-> (mloc, dbeta3[T, 3]); rows are zero where the effect did not change."""
        mloc = C.c_int(0)
        d3 = np.zeros((self.ctx.T, 3), dtype=np.float64)
        check(self.lib.gmrm_sampler_step(self.h, int(mrki), C.byref(mloc), _dp(d3)))
        return mloc.value, d3

    def abort_steps(self):
        """Leave a per-step sweep that cannot be completed (gmrm_sampler_abort_steps)."""
        check(self.lib.gmrm_sampler_abort_steps(self.h))

    def end_steps(self):
        T = self.ctx.T
        cass = np.zeros((T, self.G, self.K), dtype=np.int32)
        bsq = np.zeros((T, self.G), dtype=np.float64)
        check(self.lib.gmrm_sampler_end_steps(self.h, _ip(cass), _dp(bsq)))
        return cass, bsq

    def epilogue(self, cass, beta_sqn):
        cass = np.ascontiguousarray(cass, dtype=np.int32)
        bsq = np.ascontiguousarray(beta_sqn, dtype=np.float64)
        check(self.lib.gmrm_sampler_epilogue(self.h, _ip(cass), _dp(bsq)))

    def adopt(self, t, sigmag, pi_est, sigmae):
        sg = np.ascontiguousarray(sigmag, dtype=np.float64)
        pi = np.ascontiguousarray(pi_est, dtype=np.float64).reshape(-1)
        check(self.lib.gmrm_sampler_adopt(self.h, int(t), _dp(sg), _dp(pi), float(sigmae)))

    def save(self, path, it):
        """Checkpoint after iteration `it` (gmrm_sampler_save)."""
        check(self.lib.gmrm_sampler_save(self.h, str(path).encode(), int(it)))

    def load(self, path) -> int:
        """Restore a checkpoint; returns the iteration it was written after."""
        it = C.c_int(0)
        check(self.lib.gmrm_sampler_load(self.h, str(path).encode(), C.byref(it)))
        return it.value

    def hyper(self, t=0) -> Hyper:
        h = _lib.HyperC()
        check(self.lib.gmrm_sampler_get(self.h, int(t), C.byref(h)))
        return Hyper(h.sigmae, h.mu, h.m0_sum, np.array(h.sigmag[:self.G]),
                     np.array(h.pi_est[:self.G * self.K]), h.n_updates, h.n_batches, h.sweep_device_ms,
                     h.n_planned_stops, h.n_stale_dots, h.n_fast_batches, h.n_crossed_stops,
                     h.n_screen_tries, h.n_screened_passes)

    def csv_line(self, t, it) -> bytes:
        buf = C.create_string_buffer(50000)      # LENBUF, src/const.hpp:3
        n = check(self.lib.gmrm_sampler_csv_line(self.h, int(t), int(it), buf, 50000))
        return buf.raw[:n]
