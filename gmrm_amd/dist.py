"""Marker-sharded sampling over several GPUs of one node: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference partitions markers over MPI ranks and exchanges every updated column at
every marker step (src/bayes.cpp:495-553: Barrier, Allgather, 2x Allgatherv per step).
Here each rank sweeps its own block against its own residual replica and the replicas are
reconciled ONCE per sweep (DESIGN.md "Multi-GPU"):

  1. every rank draws mu on its own stream; rank 0's draw is adopted          (broadcast, T doubles)
  2. every rank launches its marker loop (persistent HIP kernel)               (no communication)
  3. delta = eps - eps_start is pre-rounded into two exact bins and summed     (ONE all-reduce of
     over ranks; exact, so the result does not depend on the reduction order    2*4*ceil(N/4) f64 per
     RCCL picks                                                                 phenotype)
  4. cass (int) is all-reduced; beta_sqn is all-gathered and summed in rank    (src/bayes.cpp:575-588)
     order
  5. every rank runs the hyper-parameter draws on its own stream, then adopts  (src/bayes.cpp:626,638,649)
     rank 0's sigmag / pi_est / sigmae

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
    def __init__(self, engine, group=None):
        self.e = engine
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._mmax = None

    def _exchange(self):
        e = self.e
        for t in range(e.T):
            q = e.delta_export(t)
            dist.all_reduce(q, op=dist.ReduceOp.SUM, group=self.group)
            e.delta_import(t, q)

    def iterate(self, it, sync_every=0):
        e = self.e
        T, G, K = e.T, e.G, e.K
        mu = e.small(np.asarray(e.draw_mu(it), dtype=np.float64))
        dist.broadcast(mu, src=0, group=self.group)
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
                if self.world > 1:
                    self._exchange()
                first += int(sync_every)
            cass, bsq = e.end_sweep()
        else:
            e.begin_sweep(mu.cpu().numpy())
            cass, bsq = e.end_sweep()
            if self.world > 1:                          # one shard: nothing to reconcile (exactly the
                self._exchange()                        # reference's single-rank chain)
        cass_t = e.small(np.ascontiguousarray(cass, dtype=np.int32))
        dist.all_reduce(cass_t, op=dist.ReduceOp.SUM, group=self.group)
        bsq_t = e.small(np.ascontiguousarray(bsq, dtype=np.float64))
        parts = [torch.empty_like(bsq_t) for _ in range(self.world)]
        dist.all_gather(parts, bsq_t, group=self.group)
        total = parts[0].clone()
        for p in parts[1:]:
            total += p                                  # rank order, as a sequential MPI_SUM would
        e.epilogue(cass_t.cpu().numpy(), total.cpu().numpy())
        for t in range(T):
            sg, pi, se = e.get_hyper(t)
            pack = e.small(np.concatenate([np.asarray(sg, dtype=np.float64), np.asarray(pi, dtype=np.float64).ravel(),
                                           [float(se)]]))
            dist.broadcast(pack, src=0, group=self.group)
            p = pack.cpu().numpy()
            e.adopt(t, p[:G], p[G:G + G * K], float(p[-1]))
