"""CPU test of the multi-GPU exchange logic (gmrm_amd/dist.py) under torch.distributed's
gloo backend, world_size 2.  The driver code is the product's; the engine behind it here is
a stand-in built on the oracle (tests may use the oracle; the product never does), so what is
checked is the schedule: which values are broadcast / all-reduced / all-gathered, in which
order, and that two ranks reproduce the single-process statement of the same schedule
(orc_ns_iterate) bit for bit."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


class OracleEngine:
    """Duck-typed like gmrm_amd.dist.HipEngine, computing with the oracle chain (CPU)."""

    def __init__(self, case, inp, rank, world):
        import ctypes as C
        from oracle import orc
        from gmrm_amd.api import block_of_markers
        from tests import cases
        self.orc, self.C = orc, C
        eps, mask4, nonas = cases.prepare_traits(inp)[0]
        S, Ml, _ = block_of_markers(case.M, world, rank)
        self.ch = orc.Chain(case.N, inp["bed"][S:S + Ml], eps, mask4, nonas, inp["group_index"], inp["cva"],
                            case.seed, Mt=case.M, S=S, rank=rank, canon=True)
        self.L = self.ch.L
        self.T, self.G, self.K = 1, self.ch.G, self.ch.K
        self.n4 = 4 * self.ch.n4
        self.M = Ml
        self.start = None

    def _eps_view(self):
        return np.ctypeslib.as_array(self.L.orc_chain_eps(self.ch.h), shape=(self.n4,))

    def draw_mu(self, it):
        return np.array([self.L.orc_chain_prologue_draw(self.ch.h, it)])

    def begin_sweep(self, mu):
        self.L.orc_chain_prologue_apply(self.ch.h, float(mu[0]))
        self.start = self._eps_view().copy()
        self.L.orc_chain_markers(self.ch.h)

    def begin_parts(self, mu):
        self.L.orc_chain_prologue_apply(self.ch.h, float(mu[0]))

    def launch_part(self, first, count):
        self.start = self._eps_view().copy()
        self.L.orc_chain_markers_range(self.ch.h, int(first), int(count))

    def finish_part(self):
        pass

    def preshuffle(self):
        pass

    def end_sweep(self):
        self.L.orc_chain_local_sums(self.ch.h)
        cass = self.ch.cass.reshape(1, self.G, self.K)
        bsq = np.ctypeslib.as_array(self.L.orc_chain_beta_sqn(self.ch.h), shape=(self.G,)).copy().reshape(1, self.G)
        return cass, bsq

    def delta_export(self, t):
        d = self._eps_view() - self.start
        q = np.empty(2 * self.n4)
        a, b = self.C.c_double(), self.C.c_double()
        for i in range(self.n4):
            self.L.orc_split2(float(d[i]), self.C.byref(a), self.C.byref(b))
            q[i], q[self.n4 + i] = a.value, b.value
        return torch.from_numpy(q)

    def delta_import(self, t, q):
        qq = q.numpy()
        self._eps_view()[:] = self.start + (qq[:self.n4] + qq[self.n4:])

    def epilogue(self, cass, bsq):
        c = np.ascontiguousarray(cass, dtype=np.int32).ravel()
        np.ctypeslib.as_array(self.L.orc_chain_cass(self.ch.h), shape=(self.G * self.K,))[:] = c
        np.ctypeslib.as_array(self.L.orc_chain_beta_sqn(self.ch.h), shape=(self.G,))[:] = np.asarray(bsq).ravel()
        self.L.orc_chain_epilogue(self.ch.h)

    def get_hyper(self, t):
        return self.ch.sigmag, self.ch.pi_est, self.ch.sigmae

    def adopt(self, t, sigmag, pi, sigmae):
        np.ctypeslib.as_array(self.L.orc_chain_sigmag(self.ch.h), shape=(self.G,))[:] = sigmag
        np.ctypeslib.as_array(self.L.orc_chain_pi_est(self.ch.h), shape=(self.G * self.K,))[:] = pi
        self.L.orc_chain_set_sigmae(self.ch.h, float(sigmae))

    def small(self, arr):
        return torch.as_tensor(np.asarray(arr))


def _worker(rank, world, port, outdir, sync_every=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gmrm_amd.dist import ShardedDriver
    from tests import cases
    case = cases.CASE_BY_NAME["k3"]
    inp = cases.make_inputs(case)
    eng = OracleEngine(case, inp, rank, world)
    drv = ShardedDriver(eng)
    for it in range(1, 4):
        drv.iterate(it, sync_every=sync_every)
    np.savez(Path(outdir) / f"rank{rank}.npz", betas=eng.ch.betas, comp=eng.ch.comp, eps=eng.ch.eps,
             sigmae=eng.ch.sigmae, sigmag=eng.ch.sigmag, pi=eng.ch.pi_est)
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_ranks_reproduce_the_schedule(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from tests import cases
    case = cases.CASE_BY_NAME["k3"]
    inp = cases.make_inputs(case)
    want = cases.run_oracle(case, inp, iters=3, canon=True, nranks=world)[0]
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    assert np.array_equal(np.concatenate([x["betas"] for x in r]), want["betas"][-1])
    assert np.array_equal(np.concatenate([x["comp"] for x in r]), want["comp"][-1])
    for x in r:                                           # replicas agree after the exchange
        assert np.array_equal(x["eps"], want["eps"])
        assert float(x["sigmae"]) == want["sigmae"][-1]
        assert np.array_equal(x["sigmag"], want["sigmag"][-1])
        assert np.array_equal(x["pi"], want["pi"][-1])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,k", [(2, 23), (3, 7)])
def test_ranks_reproduce_the_exchange_every_k_markers(tmp_path, world, k):
    """ShardedDriver.iterate(it, sync_every=k): blocks of different lengths (k3: 200 markers over 3 ranks), a k that does not
    divide them; bit for bit the single-process statement of the schedule (orc_nk_iterate), and not the once-per-sweep chain."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), k), nprocs=world, join=True)
    from tests import cases
    case = cases.CASE_BY_NAME["k3"]
    inp = cases.make_inputs(case)
    want = cases.run_oracle(case, inp, iters=3, canon=True, nranks=world, sync_every=k)[0]
    once = cases.run_oracle(case, inp, iters=3, canon=True, nranks=world)[0]
    r = [np.load(tmp_path / f"rank{j}.npz") for j in range(world)]
    assert np.array_equal(np.concatenate([x["betas"] for x in r]), want["betas"][-1])
    assert np.array_equal(np.concatenate([x["comp"] for x in r]), want["comp"][-1])
    assert not np.array_equal(want["betas"][-1], once["betas"][-1])
    for x in r:
        assert np.array_equal(x["eps"], want["eps"])
        assert float(x["sigmae"]) == want["sigmae"][-1]
        assert np.array_equal(x["pi"], want["pi"][-1])
