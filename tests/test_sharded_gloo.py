"""N > 1 host logic of the sharded SIPP prover (ripp_amd/sharded.py) on CPU: world_size 2, gloo backend.
The device job is replaced by an oracle-backed stand-in (test infrastructure), so what is exercised is exactly the
multi-GPU control flow: residue sharding, local halving, all-gather + multiply of partial GT products, replicated
Fiat-Shamir, tail gather.  The resulting proof must equal the single-process oracle proof byte for byte.
A second test (-m gpu) runs the REAL SippJob on two ranks sharing cuda:0 with gloo as the transport."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class OracleJob:
    """Stand-in with the SippJob staged interface, computing with the CPU oracle (Miller values as partials)."""

    def __init__(self, a, b, r):
        import orclib as o
        self.o, self.a0, self.b0, self.r0 = o, a, b, r

    def begin(self):
        self.a = self.o.scale_g1_a(self.a0, self.r0); self.b = self.b0.copy(); self.seed = None

    def local_len(self): return len(self.a)

    def round_partials(self):
        h = len(self.a) // 2
        return np.stack([self.o.miller_product_a(self.a[h:], self.b[:h]), self.o.miller_product_a(self.a[:h], self.b[h:])])

    def combine(self, gathered):
        out = gathered[0].copy()
        for g in gathered[1:]:
            for k in range(len(out)):
                out[k] = self.o.gt_mul(out[k], g[k])
        return out

    def round_finish(self, combined, digest):
        import bls381_model as m
        o = self.o
        zl, zr = o.final_exp(combined[0]), o.final_exp(combined[1])
        if self.seed is None:
            self.seed = digest
        self.seed = hashlib.blake2s(o.ser_gt(zl) + o.ser_gt(zr) + self.seed).digest()          # sipp/src/rng.rs:67-72
        x_int = int.from_bytes(m.chacha20_block(self.seed, 0)[:16], "little")                   # u128::rand
        x = o.fr_array([x_int])[0]; xinv = o.fr_array([pow(x_int, -1, o.R)])[0]
        h = len(self.a) // 2
        self.a, self.b = o.fold_g1_a(self.a[h:], self.a[:h], x), o.fold_g2_a(self.b[h:], self.b[:h], xinv)
        return zl, zr, x

    def export(self): return self.a, self.b
    def import_(self, a, b): self.a, self.b = a.copy(), b.copy()


def _worker(rank, world, port, n, use_gpu, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "model")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import orclib as o
    from ripp_amd.sharded import ShardedSippProver, TorchComm, shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b, r = o.gen_g1(123, n), o.gen_g2(456, n), o.gen_scalars(7, n)
        value = o.product_of_pairings_with_coeffs(a, b, r)
        if use_gpu:
            import ripp_amd as R
            R.init(0)
            job = R.SippJob(shard(a, rank, world), shard(b, rank, world), shard(r, rank, world), rank=rank, world=world)
            digest_fn = lambda: R.sipp_seed_digest(a, b, r, value)
        else:
            job = OracleJob(shard(a, rank, world), shard(b, rank, world), shard(r, rank, world))
            digest_fn = lambda: o.sipp_seed_digest(a, b, r, value)
        proof, ch = ShardedSippProver(job, TorchComm("cpu")).prove(digest_fn)
        rc, eproof, ech = o.sipp_prove(a, b, r, value)
        ok = rc == 0 and np.array_equal(proof, eproof) and np.array_equal(ch, ech)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _run(world, n, use_gpu):
    import torch.multiprocessing as mp
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, use_gpu, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


@pytest.mark.parametrize("n", [2, 8, 32])
def test_sharded_prover_world2_gloo_matches_single_process(n):
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    _run(2, n, use_gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4, 64])
def test_sharded_prover_world2_real_engine(engine, n):
    """Two ranks, both driving cuda:0 through the C ABI's staged interface, gloo as the transport."""
    _run(2, n, use_gpu=True)
