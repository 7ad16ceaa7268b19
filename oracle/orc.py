"""ctypes binding of the CPU oracle (oracle/gmrm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under gmrm_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
BUILD = HERE / "_build"

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_u8_p = C.POINTER(C.c_uint8)


class OrcRng(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


def _cpu_key() -> str:
    """Identify this host's ISA so a -march=native build made elsewhere is never loaded."""
    import hashlib
    try:
        txt = Path("/proc/cpuinfo").read_text()
        flags = next(ln for ln in txt.splitlines() if ln.startswith("flags"))
    except Exception:
        flags = "unknown"
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _bp(a):
    return a.ctypes.data_as(c_u8_p)


_SIGS = {
    "orc_dotp_lut_a": (c_double_p, []),
    "orc_dotp_lut_b": (c_double_p, []),
    "orc_na_lut": (c_double_p, []),
    "orc_dot_product": (C.c_double, [c_u8_p, c_double_p, C.c_int, C.c_double, C.c_double]),
    "orc_dot_product_canon": (C.c_double, [c_u8_p, c_double_p, C.c_int, C.c_double, C.c_double]),
    "orc_update_epsilon": (None, [c_double_p, c_double_p, c_u8_p, c_u8_p, C.c_int]),
    "orc_offset_epsilon": (None, [c_double_p, C.c_double, c_u8_p, C.c_int]),
    "orc_update_epsilon_canon": (None, [c_double_p, c_double_p, c_u8_p, c_u8_p, C.c_int]),
    "orc_offset_epsilon_canon": (None, [c_double_p, C.c_double, c_u8_p, C.c_int]),
    "orc_grid": (C.c_double, [C.c_double]),
    "orc_grid_array": (None, [c_double_p, C.c_int]),
    "orc_epsilon_sumsqr": (C.c_double, [c_double_p, C.c_int]),
    "orc_epsilon_sumsqr_canon": (C.c_double, [c_double_p, C.c_int]),
    "orc_epsilon_sigma": (C.c_double, [c_double_p, c_u8_p, C.c_int, C.c_int]),
    "orc_epsilon_sigma_canon": (C.c_double, [c_double_p, c_u8_p, C.c_int, C.c_int]),
    "orc_marker_stats": (None, [c_u8_p, C.c_int, C.c_int, C.c_int, c_u8_p, C.c_int, c_double_p, c_double_p]),
    "orc_marker_stats_canon": (None, [c_u8_p, C.c_int, C.c_int, C.c_int, c_u8_p, C.c_int, c_double_p, c_double_p]),
    "orc_marker_counts": (None, [c_u8_p, C.c_int, c_u8_p, C.POINTER(C.c_int64)]),
    "orc_split2": (None, [C.c_double, c_double_p, c_double_p]),
    "orc_exp": (C.c_double, [C.c_double]),
    "orc_phen_prepare": (None, [c_double_p, c_u8_p, C.c_int, c_double_p, c_u8_p, c_int_p]),
    "orc_rng_seed": (None, [C.POINTER(OrcRng), C.c_uint32]),
    "orc_rng_u32": (C.c_uint32, [C.POINTER(OrcRng)]),
    "orc_rng_unif": (C.c_double, [C.POINTER(OrcRng)]),
    "orc_rng_norm": (C.c_double, [C.POINTER(OrcRng), C.c_double, C.c_double]),
    "orc_rng_exponential": (C.c_double, [C.POINTER(OrcRng)]),
    "orc_rng_gamma": (C.c_double, [C.POINTER(OrcRng), C.c_double, C.c_double]),
    "orc_rng_beta": (C.c_double, [C.POINTER(OrcRng), C.c_double, C.c_double]),
    "orc_rng_inv_scaled_chisq": (C.c_double, [C.POINTER(OrcRng), C.c_double, C.c_double]),
    "orc_rng_shuffle": (None, [C.POINTER(OrcRng), c_int_p, C.c_int]),
    "orc_chain_create": (C.c_void_p, [C.c_int] * 6 + [c_u8_p, c_double_p, c_u8_p, C.c_int, c_int_p, c_double_p,
                                                    C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int]),
    "orc_chain_destroy": (None, [C.c_void_p]),
    "orc_chain_init": (None, [C.c_void_p]),
    "orc_chain_iterate": (None, [C.c_void_p, C.c_int]),
    "orc_chain_prologue": (None, [C.c_void_p, C.c_int]),
    "orc_chain_prologue_draw": (C.c_double, [C.c_void_p, C.c_int]),
    "orc_chain_prologue_apply": (None, [C.c_void_p, C.c_double]),
    "orc_chain_markers": (None, [C.c_void_p]),
    "orc_chain_markers_range": (None, [C.c_void_p, C.c_int, C.c_int]),
    "orc_chain_local_sums": (None, [C.c_void_p]),
    "orc_chain_epilogue": (None, [C.c_void_p]),
    "orc_ns_iterate": (None, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "orc_ps_iterate": (None, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "orc_nk_iterate": (None, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "orc_chain_eps": (c_double_p, [C.c_void_p]),
    "orc_chain_betas": (c_double_p, [C.c_void_p]),
    "orc_chain_acum": (c_double_p, [C.c_void_p]),
    "orc_chain_comp": (c_int_p, [C.c_void_p]),
    "orc_chain_midx": (c_int_p, [C.c_void_p]),
    "orc_chain_cass": (c_int_p, [C.c_void_p]),
    "orc_chain_m0": (c_int_p, [C.c_void_p]),
    "orc_chain_sigmag": (c_double_p, [C.c_void_p]),
    "orc_chain_pi_est": (c_double_p, [C.c_void_p]),
    "orc_chain_beta_sqn": (c_double_p, [C.c_void_p]),
    "orc_chain_mave": (c_double_p, [C.c_void_p]),
    "orc_chain_msig": (c_double_p, [C.c_void_p]),
    "orc_chain_sigmae": (C.c_double, [C.c_void_p]),
    "orc_chain_mu": (C.c_double, [C.c_void_p]),
    "orc_chain_set_sigmae": (None, [C.c_void_p, C.c_double]),
    "orc_chain_m0_sum": (C.c_int, [C.c_void_p]),
    "orc_chain_nupdates": (C.c_long, [C.c_void_p]),
    "orc_chain_rng_d": (C.POINTER(OrcRng), [C.c_void_p]),
    "orc_chain_rng_m": (C.POINTER(OrcRng), [C.c_void_p]),
    "orc_csv_line": (C.c_int, [C.c_char_p, C.c_size_t, C.c_uint, c_double_p, C.c_int, C.c_double, C.c_int,
                               c_double_p, C.c_int]),
    "orc_predict_g": (None, [c_u8_p, C.c_int, C.c_int, c_u8_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
    "orc_assoc": (None, [c_u8_p, C.c_int, C.c_int, c_u8_p, C.c_int, c_double_p, c_double_p, c_double_p]),
    "orc_mlma_stats": (None, [C.c_double, C.c_double, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p]),
    "orc_mlma_line": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                C.c_double, C.c_double]),
}

_libs = {}


def lib(fast: bool = False):
    """Load (building if needed) the strict oracle -- the parity checker -- or, with
    fast=True, the timing build: the reference's flags (-Ofast -march=native -fopenmp,
    setup/Make.gcc_mvapich2:15-16) compiled for THIS host."""
    key = "fast" if fast else "strict"
    if key in _libs:
        return _libs[key]
    src = HERE / "gmrm_oracle.c"
    BUILD.mkdir(exist_ok=True)
    if fast:
        path = BUILD / f"liborc_fast_{_cpu_key()}.so"
        flags = ["-Ofast", "-fPIC", "-std=c11", "-fopenmp", "-march=native"]
    else:
        path = BUILD / "liborc.so"
        flags = ["-O2", "-fPIC", "-std=c11", "-mfma", "-ffp-contract=off", "-fno-fast-math"]
    if not path.exists() or path.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["gcc", *flags, "-shared", "-o", str(path), str(src), "-lm"], check=True)
    L = C.CDLL(str(path))
    for name, (res, args) in _SIGS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _libs[key] = L
    return L


def lib_fast_native():
    return lib(fast=True)


def im4_of(N: int) -> int:
    return N // 4 if N % 4 == 0 else N // 4 + 1


def phen_prepare(y, isna):
    """phenotype.cpp:587-673 -> (eps[4*im4], mask4[im4], nonas)."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    isna = np.ascontiguousarray(isna, dtype=np.uint8)
    N = y.shape[0]
    n4 = im4_of(N)
    eps = np.zeros(4 * n4, dtype=np.float64)
    mask4 = np.zeros(n4, dtype=np.uint8)
    nonas = C.c_int(0)
    lib().orc_phen_prepare(_dp(y), _bp(isna), N, _dp(eps), _bp(mask4), C.byref(nonas))
    return eps, mask4, nonas.value


def grid(x):
    """The canon residual's grid: x rounded to the nearest multiple of 2^-44 (ties to even)."""
    a = np.array(x, dtype=np.float64, copy=True).reshape(-1)
    lib().orc_grid_array(_dp(a), a.shape[0])
    return a if np.ndim(x) else float(a[0])


class Chain:
    """One phenotype's Gibbs chain on one rank, run by the oracle."""

    def __init__(self, N, bed_local, eps0, mask4, nonas, group_index, cva, seed, *, Mt=None, S=0,
                 rank=0, shuffle=True, mimic_hydra=False, canon=True, fast=False):
        self.L = lib(fast)
        self.bed = np.ascontiguousarray(bed_local, dtype=np.uint8)
        self.N = int(N)
        self.M = int(self.bed.shape[0])
        self.Mt = int(Mt if Mt is not None else self.M)
        self.S = int(S)
        cva = np.ascontiguousarray(cva, dtype=np.float64)
        self.G, self.K = cva.shape
        self.n4 = im4_of(self.N)
        assert self.bed.shape[1] == self.n4
        gi = np.ascontiguousarray(group_index, dtype=np.int32)
        assert gi.shape[0] == self.Mt
        eps0 = np.ascontiguousarray(eps0, dtype=np.float64)
        mask4 = np.ascontiguousarray(mask4, dtype=np.uint8)
        self.h = self.L.orc_chain_create(self.N, self.M, self.Mt, self.S, self.G, self.K, _bp(self.bed),
                                         _dp(eps0), _bp(mask4), int(nonas), _ip(gi), _dp(cva),
                                         int(seed), int(rank), int(shuffle), int(mimic_hydra), int(canon))
        if not self.h:
            raise RuntimeError("orc_chain_create failed")
        self.L.orc_chain_init(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_chain_destroy(self.h)
            self.h = None

    def _arr(self, fn, n, dtype):
        p = getattr(self.L, fn)(self.h)
        return np.ctypeslib.as_array(p, shape=(n,)).astype(dtype, copy=True)

    def iterate(self, it):
        self.L.orc_chain_iterate(self.h, int(it))

    eps = property(lambda s: s._arr("orc_chain_eps", 4 * s.n4, np.float64))
    betas = property(lambda s: s._arr("orc_chain_betas", s.M, np.float64))
    acum = property(lambda s: s._arr("orc_chain_acum", s.M, np.float64))
    comp = property(lambda s: s._arr("orc_chain_comp", s.M, np.int32))
    midx = property(lambda s: s._arr("orc_chain_midx", s.M, np.int32))
    cass = property(lambda s: s._arr("orc_chain_cass", s.G * s.K, np.int32))
    m0 = property(lambda s: s._arr("orc_chain_m0", s.G, np.int32))
    sigmag = property(lambda s: s._arr("orc_chain_sigmag", s.G, np.float64))
    pi_est = property(lambda s: s._arr("orc_chain_pi_est", s.G * s.K, np.float64))
    mave = property(lambda s: s._arr("orc_chain_mave", s.M, np.float64))
    msig = property(lambda s: s._arr("orc_chain_msig", s.M, np.float64))
    sigmae = property(lambda s: s.L.orc_chain_sigmae(s.h))
    mu = property(lambda s: s.L.orc_chain_mu(s.h))
    m0_sum = property(lambda s: s.L.orc_chain_m0_sum(s.h))
    n_updates = property(lambda s: s.L.orc_chain_nupdates(s.h))

    def csv_line(self, it):
        buf = C.create_string_buffer(50000)
        sg = self.sigmag
        pi = self.pi_est
        n = self.L.orc_csv_line(buf, 50000, int(it), _dp(sg), self.G, self.sigmae, self.m0_sum, _dp(pi), self.K)
        return buf.raw[:n]


def ns_iterate(chains, it):
    """The build's sweep-synchronous multi-rank schedule (orc_ns_iterate)."""
    arr = (C.c_void_p * len(chains))(*[c.h for c in chains])
    chains[0].L.orc_ns_iterate(arr, len(chains), int(it))


def nk_iterate(chains, it, k):
    """orc_ns_iterate with the residual exchange every k marker positions (orc_nk_iterate)."""
    arr = (C.c_void_p * len(chains))(*[c.h for c in chains])
    chains[0].L.orc_nk_iterate(arr, len(chains), int(it), int(k))


def ps_iterate(chains, it):
    """The reference's per-step multi-rank schedule, bayes.cpp:374-553 (orc_ps_iterate)."""
    arr = (C.c_void_p * len(chains))(*[c.h for c in chains])
    chains[0].L.orc_ps_iterate(arr, len(chains), int(it))


def predict_g(bed, mask4, mave, msig, beta):
    """bayes.cpp:93-122 for one block: bed [M, mbytes] uint8 -> g_k [4*im4]."""
    bed = np.ascontiguousarray(bed, dtype=np.uint8)
    M, mbytes = bed.shape
    mask4 = np.ascontiguousarray(mask4, dtype=np.uint8)
    g = np.zeros(4 * mbytes, dtype=np.float64)
    lib().orc_predict_g(_bp(bed), M, mbytes, _bp(mask4), mbytes, _dp(np.ascontiguousarray(mave, dtype=np.float64)),
                        _dp(np.ascontiguousarray(msig, dtype=np.float64)), _dp(np.ascontiguousarray(beta, dtype=np.float64)), _dp(g))
    return g


def assoc(bed, mask4, y_k):
    """bayes.cpp:172-196 for one block: xtx, xty per marker (y_k has 4*im4 entries)."""
    bed = np.ascontiguousarray(bed, dtype=np.uint8)
    M, mbytes = bed.shape
    y = np.zeros(4 * mbytes, dtype=np.float64)
    y[:len(y_k)] = y_k
    xtx = np.zeros(M, dtype=np.float64)
    xty = np.zeros(M, dtype=np.float64)
    lib().orc_assoc(_bp(bed), M, mbytes, _bp(np.ascontiguousarray(mask4, dtype=np.uint8)), mbytes, _dp(y), _dp(xtx), _dp(xty))
    return xtx, xty


def mlma_stats(xtx, xty, sigma):
    out = [np.zeros(1) for _ in range(4)]
    lib().orc_mlma_stats(float(xtx), float(xty), float(sigma), *[_dp(o) for o in out])
    return tuple(float(o[0]) for o in out)


def mlma_line(rsid, mglo, rmglo, beta, tdist, se, pval):
    buf = C.create_string_buffer(256)
    n = lib().orc_mlma_line(buf, 256, rsid.encode(), int(mglo), int(rmglo), float(beta), float(tdist), float(se), float(pval))
    return buf.raw[:n]
