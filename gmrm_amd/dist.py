"""Marker-sharded sampling over several GPUs of one node: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference partitions markers over MPI ranks and exchanges every updated column at
every marker step (src/bayes.cpp:495-553: Barrier, Allgather, 2x Allgatherv per step).
Here each rank sweeps its own block against its own residual replica and the replicas are
reconciled ONCE per sweep (DESIGN.md "Multi-GPU"):

  1. every rank draws mu on its own stream; rank 0's draw is adopted          (broadcast, T doubles)
  2. every rank launches its marker loop (persistent HIP kernel)               (no communication)
  3. ONE all-reduce (sum, f64) of one device buffer per sweep (round 4; five collectives before):
       [ delta of every phenotype | cass | beta_sqn slots ]
     delta = eps - eps_start pre-rounded into two exact bins (2*4*ceil(N/4) doubles per phenotype; exact, so the result
     does not depend on the reduction order RCCL picks), written into the buffer by the library and read back from it
     -- it never leaves the device; cass as exact doubles (src/bayes.cpp:575-588); every rank's beta_sqn in a slot of
     its own (zeros elsewhere), so the sum IS the all-gather and the ranks add the slots in rank order, as a sequential
     MPI_SUM would
  4. every rank runs the hyper-parameter draws on its own stream, then adopts  (src/bayes.cpp:626,638,649)
     rank 0's sigmag / pi_est / sigmae: one broadcast for all phenotypes

iterate(it, sync_every=k) with k > 0 cuts steps 2-3 into parts of k marker positions: every rank sweeps positions
[p k, (p + 1) k) of its own visit order, then the replicas are reconciled (step 3), and so on to the end of the longest
block -- between the once-per-sweep exchange and the reference's exchange after every marker (`--sync-every k`).

`engine` is anything with the Sampler/Context split-call surface (gmrm_amd.api); the CPU
tests drive this same code with a stand-in engine to check the exchange logic under gloo.
"""
import numpy as np
import torch
import torch.distributed as dist


class HipEngine:
    """The product engine: gmrm_amd.Sampler + Context on this rank's GPU."""

    def __init__(self, sampler, device, host_staging=False):
        """host_staging=True routes the exchanged tensors through host memory (for the gloo
        backend, e.g. two test ranks sharing one GPU); the default hands RCCL device memory."""
        self.s = sampler
        self.ctx = sampler.ctx
        self.T, self.G, self.K = self.ctx.T, sampler.G, sampler.K
        self.n4 = 4 * self.ctx.mbytes
        self.device = device
        self.host_staging = host_staging
        self.M = self.ctx.M
        self._q = torch.empty(2 * self.n4, dtype=torch.float64, device=device)

    def draw_mu(self, it):
        return self.s.draw_mu(it)

    def begin_sweep(self, mu):
        self.s.begin_sweep(mu)

    def end_sweep(self):
        return self.s.end_sweep()

    def begin_parts(self, mu):
        self.s.begin_parts(mu)

    def launch_part(self, first, count):
        self.s.launch_part(first, count)

    def finish_part(self):
        self.s.finish_part()

    def preshuffle(self):
        self.s.preshuffle()

    def alloc(self, n):
        """The exchange buffer: device memory RCCL reduces in place (host memory under host_staging)."""
        return torch.zeros(n, dtype=torch.float64, device="cpu" if self.host_staging else self.device)

    def delta_export_into(self, t, view):
        """The two exact parts of (residual - snapshot) of phenotype t, written by the library straight into `view`
        (a contiguous slice of the exchange buffer)."""
        if self.host_staging:
            self.ctx.eps_delta_export(t, self._q.data_ptr())
            view.copy_(self._q)
        else:
            self.ctx.eps_delta_export(t, view.data_ptr())        # (the library synchronises its stream before it returns)

    def delta_import_from(self, t, view):
        if self.host_staging:
            self._q.copy_(view)
            torch.cuda.current_stream(self.device).synchronize()
            self.ctx.eps_delta_import(t, self._q.data_ptr())
        else:
            self.ctx.eps_delta_import(t, view.data_ptr())

    # (kept for callers that exchange phenotype by phenotype: tests/test_gpu_chain.py)
    def delta_export(self, t):
        torch.cuda.current_stream(self.device).synchronize()
        self.ctx.eps_delta_export(t, self._q.data_ptr())
        return self._q.cpu() if self.host_staging else self._q

    def delta_import(self, t, q):
        if self.host_staging:
            self._q.copy_(q)
            q = self._q
        torch.cuda.current_stream(self.device).synchronize()
        self.ctx.eps_delta_import(t, q.data_ptr())

    def epilogue(self, cass, bsq):
        self.s.epilogue(cass, bsq)

    def get_hyper(self, t):
        h = self.s.hyper(t)
        return h.sigmag, h.pi_est, h.sigmae

    def adopt(self, t, sigmag, pi, sigmae):
        self.s.adopt(t, sigmag, pi, sigmae)

    def small(self, arr):
        return torch.as_tensor(arr, device="cpu" if self.host_staging else self.device)


class ShardedDriver:
    def __init__(self, engine, group=None, force_exchange=False):
        """force_exchange: run the residual exchange even with ONE rank (an identity: what a one-GPU box can measure of its cost)."""
        self.e = engine
        self.group = group
        self.world = dist.get_world_size(group)
        self.exchange = self.world > 1 or bool(force_exchange)
        self.rank = dist.get_rank(group)
        self._mmax = None
        e = engine
        self.nq = 2 * e.n4                               # doubles of one phenotype's delta
        self.n_cass = e.T * e.G * e.K
        self.n_bsq = e.T * e.G
        self.n_delta = e.T * self.nq
        alloc = getattr(e, "alloc", None)
        n = self.n_delta + self.n_cass + self.world * self.n_bsq
        self.buf = alloc(n) if alloc else torch.zeros(n, dtype=torch.float64)
        self.collectives = 0                             # counted, for the tests and the bench line

    # ---- the exchange buffer: [ deltas of every phenotype | cass | beta_sqn slot of rank 0 | ... | of rank world-1 ]
    def _export_deltas(self):
        e = self.e
        for t in range(e.T):
            view = self.buf[t * self.nq:(t + 1) * self.nq]
            if hasattr(e, "delta_export_into"):
                e.delta_export_into(t, view)
            else:                                        # (engines of the CPU tests: tensor in, tensor out)
                view.copy_(e.delta_export(t))

    def _import_deltas(self):
        e = self.e
        if self.buf.is_cuda:
            torch.cuda.current_stream(self.buf.device).synchronize()     # the collective is done before the library reads
        for t in range(e.T):
            view = self.buf[t * self.nq:(t + 1) * self.nq]
            if hasattr(e, "delta_import_from"):
                e.delta_import_from(t, view)
            else:
                e.delta_import(t, view)

    def _reduce(self, with_deltas, cass=None, bsq=None):
        """One all-reduce over the part of the buffer that is needed; returns (cass, beta_sqn summed in rank order) when asked."""
        lo = 0 if with_deltas else self.n_delta
        hi = self.n_delta if cass is None else self.buf.numel()
        if cass is not None:
            small = np.zeros(self.n_cass + self.world * self.n_bsq)
            small[:self.n_cass] = np.asarray(cass, dtype=np.float64).ravel()            # exact: small integers
            o = self.n_cass + self.rank * self.n_bsq
            small[o:o + self.n_bsq] = np.asarray(bsq, dtype=np.float64).ravel()         # this rank's slot; zeros elsewhere
            self.buf[self.n_delta:].copy_(torch.from_numpy(small))
        if with_deltas:
            self._export_deltas()
        dist.all_reduce(self.buf[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        self.collectives += 1
        out = None
        if cass is not None:
            small = self.buf[self.n_delta:].cpu().numpy()                               # (waits for the collective)
            cass_sum = np.rint(small[:self.n_cass]).astype(np.int32).reshape(np.asarray(cass).shape)
            slots = small[self.n_cass:].reshape(self.world, -1)
            total = slots[0].copy()
            for r in range(1, self.world):
                total += slots[r]                        # rank order, as a sequential MPI_SUM would
            out = (cass_sum, total.reshape(np.asarray(bsq).shape))
        if with_deltas:
            self._import_deltas()
        return out

    def iterate(self, it, sync_every=0):
        e = self.e
        T, G, K = e.T, e.G, e.K
        mu = e.small(np.asarray(e.draw_mu(it), dtype=np.float64))
        dist.broadcast(mu, src=0, group=self.group)
        self.collectives += 1
        if sync_every and sync_every > 0:
            if self._mmax is None:                      # the longest block decides how many parts a sweep has
                m = e.small(np.asarray([e.M], dtype=np.int64))
                dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
                self._mmax = int(m.cpu().numpy()[0])
            e.begin_parts(mu.cpu().numpy())
            first = 0
            while first < self._mmax:
                f = min(first, e.M)
                e.launch_part(f, min(int(sync_every), e.M - f))
                if first == 0:
                    e.preshuffle()                      # the next iteration's shuffle, beside the first part
                e.finish_part()
                first += int(sync_every)
                if self.exchange and first < self._mmax:
                    self._reduce(True)                  # the deltas of this part
            cass, bsq = e.end_sweep()
            cass, total = self._reduce(self.exchange, cass, bsq)         # the last part's deltas ride with the counts
        else:
            e.begin_sweep(mu.cpu().numpy())
            cass, bsq = e.end_sweep()
            # one shard: nothing to reconcile in the residual (exactly the reference's single-rank chain)
            cass, total = self._reduce(self.exchange, cass, bsq)
        e.epilogue(cass, total)
        # rank 0's hyper-parameters for every phenotype, one broadcast
        per = G + G * K + 1
        pack = np.empty(T * per)
        for t in range(T):
            sg, pi, se = e.get_hyper(t)
            pack[t * per:(t + 1) * per] = np.concatenate([np.asarray(sg, dtype=np.float64), np.asarray(pi, dtype=np.float64).ravel(), [float(se)]])
        pk = e.small(pack)
        dist.broadcast(pk, src=0, group=self.group)
        self.collectives += 1
        p = pk.cpu().numpy()
        for t in range(T):
            q = p[t * per:(t + 1) * per]
            e.adopt(t, q[:G], q[G:G + G * K], float(q[-1]))
