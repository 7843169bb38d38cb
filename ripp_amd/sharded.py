"""Multi-GPU SIPP prover: one process per GPU, vectors sharded by index residue, partial GT products combined over
RCCL (torch.distributed backend "nccl" on ROCm) -- the host logic of SURVEY.md section 8(e).

Sharding: global element i lives on rank i mod G at local index i div G.  Every halving round pairs (i, i + L/2);
while L/2 is a multiple of G both partners have the same residue, so the fold is 100% local and each rank simply
halves its own shard with the common challenge.  The only exchange per round is the rank's two partial Miller values
(2 x 576 B): all-gather + local multiply (RCCL has no Fq12-product reduction op); each rank folds its own 68 per-step
products first, which is valid because the f <- f^2 * L recurrence is multiplicative.  When every rank is
down to ONE element (L == G) the G remaining elements are all-gathered and the last log2(G) rounds run replicated.

The engine (device job) and the communicator are injected, so the N > 1 control flow is testable on CPU with gloo
and an oracle-backed stand-in (tests/test_sharded_gloo.py); production uses ripp_amd.api.SippJob + TorchComm.
"""
import numpy as np


class TorchComm:
    """torch.distributed plumbing: gloo on CPU tensors, nccl (= RCCL over xGMI) on the rank's GPU."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")

    def all_gather(self, arr):
        """arr: np.uint64 array, same shape on every rank -> list of `world` arrays in rank order."""
        t = self.torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy()).to(self.device)
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [o.cpu().numpy().view(np.uint64).reshape(arr.shape) for o in outs]

    def broadcast_bytes(self, data, src=0):
        buf = np.frombuffer(data if data is not None else bytes(32), dtype=np.uint8).copy()
        t = self.torch.from_numpy(buf).to(self.device)
        self.dist.broadcast(t, src=src)
        return bytes(t.cpu().numpy())

    def barrier(self):
        self.dist.barrier()


class SingleComm:
    rank, world = 0, 1

    def all_gather(self, arr): return [arr]
    def broadcast_bytes(self, data, src=0): return data
    def barrier(self): pass


def shard(arr, rank, world):
    """rank's shard of a full vector: elements rank, rank + world, ..."""
    return np.ascontiguousarray(arr[rank::world])


class ShardedSippProver:
    """Drives one SIPP proof (sipp/src/lib.rs:42-106) across `comm.world` ranks.

    job: object with begin(), local_len(), round_partials(), combine(list), round_finish(combined, digest),
         export(), import_(a, b)   (ripp_amd.api.SippJob, or the oracle-backed stand-in of the tests)
    """

    def __init__(self, job, comm):
        self.job, self.comm = job, comm

    def prove(self, seed_digest_fn):
        """seed_digest_fn(): Blake2s digest of the full statement -- evaluated on rank 0 only, concurrently with
        the first kernels where the implementation allows, then broadcast.  Returns (proof (2*rounds,72), challenges)."""
        job, comm = self.job, self.comm
        job.begin()
        digest = None
        proof, challenges = [], []
        first = True
        while True:
            if job.local_len() == 1:
                if comm.world == 1 or getattr(self, "_tail", False):
                    break
                # tail: gather the world's remaining elements (rank order == global order), continue replicated
                a1, b1 = job.export()
                ga = np.concatenate(comm.all_gather(a1), axis=0); gb = np.concatenate(comm.all_gather(b1), axis=0)
                job.import_(ga, gb)
                self._tail = True
                continue
            partials = job.round_partials()
            if getattr(self, "_tail", False) or comm.world == 1:
                combined = partials
            else:
                combined = job.combine(comm.all_gather(partials))
            if first:
                digest = comm.broadcast_bytes(seed_digest_fn() if comm.rank == 0 else None, src=0)
                first = False
            zl, zr, x = job.round_finish(combined, digest)
            proof += [zl, zr]; challenges.append(x)
        self._tail = False
        if not proof:
            return np.zeros((0, 72), dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64)
        return np.stack(proof), np.stack(challenges)
