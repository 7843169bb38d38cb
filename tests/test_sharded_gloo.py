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
@pytest.mark.parametrize("n", [64])
def test_sharded_prover_world2_real_engine(engine, n):
    """Two ranks, both driving cuda:0 through the C ABI's staged interface, gloo as the transport."""
    _run(2, n, use_gpu=True)


# ---------------------------------------------------------------- sharded inner products
class OraclePrimitives:
    """Stand-in for ripp_amd.sharded.HipPrimitives computing with the CPU oracle."""

    def __init__(self):
        import orclib as o
        self.o = o

    def pairing_miller(self, left, right): return self.o.miller_product_a(self.o.normalize_g1(left), self.o.normalize_g2(right))
    def final_exp(self, f): return self.o.final_exp(f)
    def gt_mul(self, a, b): return self.o.gt_mul(a, b)
    def msm_g1(self, bases, scalars): return self.o.msm_g1_j(bases, scalars)[1]
    def msm_g2(self, bases, scalars): return self.o.msm_g2_j(bases, scalars)[1]

    def sum_points(self, pts, cols):
        o = self.o; acc = pts[0].copy()
        import ctypes
        for p in pts[1:]:
            out = np.zeros(cols, dtype=np.uint64)
            (o.lib().orc_g1_add_j if cols == 18 else o.lib().orc_g2_add_j)(o._p(acc), o._p(np.ascontiguousarray(p)), o._p(out)); acc = out
        return acc


def _ip_worker(rank, world, port, n, use_gpu, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "model")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import orclib as o
    from ripp_amd.sharded import TorchComm, shard, sharded_pairing_inner_product, sharded_msm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b, s = o.blind_g1(o.gen_g1(31, n), 1), o.blind_g2(o.gen_g2(41, n), 2), o.gen_scalars(9, n)
        if use_gpu:
            import ripp_amd as R
            R.init(0); prim = None
        else:
            prim = OraclePrimitives()
        comm = TorchComm("cpu")
        ip = sharded_pairing_inner_product(comm, shard(a, rank, world), shard(b, rank, world), prim)
        m1 = sharded_msm(comm, shard(a, rank, world), shard(s, rank, world), "g1", prim)
        m2 = sharded_msm(comm, shard(b, rank, world), shard(s, rank, world), "g2", prim)
        ok = np.array_equal(ip, o.pairing_product_j(a, b)[1])
        ok = ok and np.array_equal(o.g1_to_affine(m1), o.g1_to_affine(o.msm_g1_j(a, s)[1])) and np.array_equal(o.g2_to_affine(m2), o.g2_to_affine(o.msm_g2_j(b, s)[1]))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _run_ip(world, n, use_gpu):
    import torch.multiprocessing as mp
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_ip_worker, args=(world, _free_port(), n, use_gpu, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


@pytest.mark.parametrize("n", [2, 10])
def test_sharded_inner_products_world2_gloo(n):
    """PairingInnerProduct and both MSMs over residue-sharded vectors: all-gather of one Miller value / one point per rank, one final
    exponentiation -- equal to the unsharded result (host logic on CPU, oracle-backed primitives)."""
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    _run_ip(2, n, use_gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1000])
def test_sharded_inner_products_world2_real_engine(engine, n):
    _run_ip(2, n, use_gpu=True)


# ---------------------------------------------------------------- native driver: round loop + collective inside libripp_hip.so
def _native_worker(rank, world, port, n, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "model")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import orclib as o
    import ripp_amd as R
    from ripp_amd.sharded import NativeComm, shard, native_sipp_job_prove, native_pairing_inner_product, native_msm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        R.init(0)
        # both ranks share cuda:0: RCCL needs one device per rank, gloo carries the all-gather.  With RIPP_COMM_NO_RCCL the RCCL transport is
        # REQUESTED and its bring-up made to fail, so the ranks must agree on the fallback to the host's process group (what bench.py relies on)
        comm = NativeComm("rccl" if os.environ.get("RIPP_COMM_NO_RCCL") else "callback")
        assert comm.transport == "callback"
        a, b, r = o.gen_g1(123, n), o.gen_g2(456, n), o.gen_scalars(7, n)
        value = o.product_of_pairings_with_coeffs(a, b, r)
        job = R.SippJob(shard(a, rank, world), shard(b, rank, world), shard(r, rank, world), rank=rank, world=world)
        proof, ch, _ = native_sipp_job_prove(job, value, full=(a, b, r) if rank == 0 else None)
        rc, eproof, ech = o.sipp_prove(a, b, r, value)
        ok = rc == 0 and np.array_equal(proof, eproof) and np.array_equal(ch, ech)
        # second proof on the same resident shard, digest precomputed by the host
        proof2, _, _ = native_sipp_job_prove(job, value, seed_digest=R.sipp_seed_digest(a, b, r, value) if rank == 0 else None)
        ok = ok and np.array_equal(proof2, eproof)
        job.close()
        aj, bj, s = o.blind_g1(a, 1), o.blind_g2(b, 2), o.gen_scalars(9, n)
        ok = ok and np.array_equal(native_pairing_inner_product(shard(aj, rank, world), shard(bj, rank, world)), o.pairing_product_j(aj, bj)[1])
        ok = ok and np.array_equal(o.g1_to_affine(native_msm(shard(aj, rank, world), shard(s, rank, world), "g1")), o.g1_to_affine(o.msm_g1_j(aj, s)[1]))
        ok = ok and np.array_equal(o.g2_to_affine(native_msm(shard(bj, rank, world), shard(s, rank, world), "g2")), o.g2_to_affine(o.msm_g2_j(bj, s)[1]))
        comm.close()
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4, 64, 1 << 12])
def test_native_sharded_driver_world2_callback_transport(engine, n):
    """ripp_sipp_job_prove_sharded / ripp_*_sharded_j: the library's own round loop and collectives, two ranks on cuda:0 with the
    all-gather supplied by the host (gloo): proofs, pairing product and MSMs equal the oracle's on the unsharded vectors."""
    import torch.multiprocessing as mp
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_native_worker, args=(2, _free_port(), n, ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.gpu
@pytest.mark.parametrize("n,items", [(16, 6), (64, 4), (1 << 12, 6), (1 << 14, 3)])
def test_native_sharded_driver_world2_with_lookahead(engine, n, items):
    """The sharded prover with the multi-GPU look-ahead plan forced onto small statements (RIPP_LOOK_ITEMS): every rank pre-evaluates the
    values of rounds 1..3 from ITS shard's round-0 blocks, reduces them with the challenges and contributes the partial GT values; folds of
    those rounds are not waited for.  Proofs equal the oracle's on the unsharded vectors."""
    import torch.multiprocessing as mp
    os.environ["RIPP_LOOK_ITEMS"] = str(items)
    try:
        mgr = mp.Manager(); ret = mgr.dict()
        mp.spawn(_native_worker, args=(2, _free_port(), n, ret), nprocs=2, join=True)
    finally:
        del os.environ["RIPP_LOOK_ITEMS"]
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.gpu
def test_native_comm_falls_back_to_the_process_group(engine):
    """bench.py asks for the library's own RCCL communicator; when that cannot be brought up on every rank the ranks agree (one all-reduce)
    to run the library's collectives through torch.distributed instead.  Forced here with RIPP_COMM_NO_RCCL on two ranks sharing cuda:0."""
    import torch.multiprocessing as mp
    os.environ["RIPP_COMM_NO_RCCL"] = "1"
    try:
        mgr = mp.Manager(); ret = mgr.dict()
        mp.spawn(_native_worker, args=(2, _free_port(), 64, ret), nprocs=2, join=True)
    finally:
        del os.environ["RIPP_COMM_NO_RCCL"]
    assert dict(ret) == {0: True, 1: True}


_RCCL_SCRIPT = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist            # torch FIRST: the order bench.py uses (the library then shares torch's HIP runtime)
for p in sys.argv[1:4]:
    sys.path.insert(0, p)
torch.cuda.set_device(0)
import ripp_amd as R
R.init(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import ctypes
from ripp_amd._lib import lib
from ripp_amd import api
from ripp_amd.sharded import NativeComm, native_sipp_job_prove
comm = NativeComm("rccl")                          # ripp_comm_unique_id -> (broadcast over torch.distributed) -> ripp_comm_init
assert lib().ripp_comm_world() == 1 and lib().ripp_comm_rank() == 0
send = np.arange(1152, dtype=np.uint8); recv = np.zeros(1152, dtype=np.uint8)
api._check(lib().ripp_comm_allgather(api._p(send), api._p(recv), ctypes.c_size_t(1152)))       # ncclAllGather on the engine's stream
assert np.array_equal(send, recv)
import orclib as o
n = 64
a, b, r = o.gen_g1(123, n), o.gen_g2(456, n), o.gen_scalars(7, n)
value = o.product_of_pairings_with_coeffs(a, b, r)
job = R.SippJob(a, b, r)
proof, ch, _ = native_sipp_job_prove(job, value, full=(a, b, r))
job.close()
rc, eproof, ech = o.sipp_prove(a, b, r, value)
assert rc == 0 and np.array_equal(proof, eproof) and np.array_equal(ch, ech)
comm.close(); dist.destroy_process_group()
print("RCCL-TRANSPORT-OK")
"""


@pytest.mark.gpu
def test_native_rccl_transport_single_rank(engine):
    """The RCCL transport itself (librccl.so loaded by the library beside the HIP runtime in use, ncclCommInitRank / ncclAllGather on
    the engine's stream) with one rank -- all a 1-GPU box can run; the multi-rank protocol is the callback test above.  Runs in a fresh
    process in bench.py's import order (a process that has mixed two ROCm installations in the other order cannot initialise RCCL)."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT, os.path.dirname(HERE), HERE, os.path.join(HERE, "model")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL-TRANSPORT-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


# ---------------------------------------------------------------- sharded GIPA / aggregate_proofs (config 5 across ranks)
def _agg_worker(rank, world, port, n, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "model")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import orclib as o
    import helpers as h
    import ripp_amd as R
    from ripp_amd.sharded import NativeComm, shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        R.init(0)
        comm = NativeComm("callback")
        ok = True
        # GIPA / TIPP on sharded vectors vs the oracle's prover on the whole vectors
        m_a, m_b = o.blind_g1(o.gen_g1(11, n), 1), o.blind_g2(o.gen_g2(22, n), 2)
        ck_a, ck_b = o.blind_g2(o.gen_g2(33, n), 3), o.blind_g1(o.gen_g1(44, n), 4)
        steps, tr, (ba, bb), (ka, kb) = R.gipa_tipp_prove_sharded(shard(m_a, rank, world), shard(m_b, rank, world), shard(ck_a, rank, world), shard(ck_b, rank, world))
        rc, esteps, etr, eba, ebb, eka, ekb = o.gipa_tipp_prove(m_a, m_b, ck_a, ck_b)
        ok = ok and rc == 0 and np.array_equal(steps, esteps) and np.array_equal(tr, etr)
        ok = ok and np.array_equal(o.g1_to_affine(ba), o.g1_to_affine(eba)) and np.array_equal(o.g2_to_affine(bb), o.g2_to_affine(ebb))
        ok = ok and np.array_equal(o.g2_to_affine(ka), o.g2_to_affine(eka)) and np.array_equal(o.g1_to_affine(kb), o.g1_to_affine(ekb))
        # aggregate_proofs: every member equal to the oracle's, and the oracle's verifier accepts
        osrs = h.make_srs(n, 0xa1fa + n, 0xbe7a + n); srs = R.SRS(osrs[0], osrs[1], osrs[2], osrs[3])
        vk, pub, a, b, c = h.fake_groth16(n, 2, seed=n)
        got, _ = R.aggregate_proofs_sharded(srs, shard(a, rank, world), shard(b, rank, world), shard(c, rank, world))
        rc, exp = o.aggregate_proofs(osrs[0], osrs[1], a, b, c)
        ok = ok and rc == 0
        rounds = n.bit_length() - 1
        for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c"):
            ok = ok and np.array_equal(got.field(k), exp.field(k))
        for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript"):
            ok = ok and np.array_equal(getattr(got, k), getattr(exp, k))
        for k in ("agg_c", "ab_base_a", "ab_final_ck_b", "ab_opening_b", "c_base_a"):
            ok = ok and np.array_equal(o.g1_to_affine(got.field(k)), o.g1_to_affine(exp.field(k)))
        for k in ("ab_base_b", "ab_final_ck_a", "ab_opening_a", "c_final_ck_a", "c_opening_a"):
            ok = ok and np.array_equal(o.g2_to_affine(got.field(k)), o.g2_to_affine(exp.field(k)))
        ok = ok and np.array_equal(o.normalize_g1(got.c_com_g1[: 2 * rounds]), o.normalize_g1(exp.c_com_g1[: 2 * rounds]))
        ok = ok and o.verify_aggregate_proof(h.verifier_srs(osrs), vk, pub, got) == 1
        srs.close(); comm.close()
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 8, 256, 1 << 14])
def test_sharded_gipa_and_aggregate_world2(engine, n):
    """ripp_gipa_tipp_prove_sharded / ripp_aggregate_proofs_sharded with two ranks (callback transport on cuda:0): commitments of every
    round, transcripts, base cases, KZG openings and the aggregate's members equal the oracle's on the unsharded vectors."""
    import torch.multiprocessing as mp
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_agg_worker, args=(2, _free_port(), n, ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device_reports_the_hash_excluded_figures(engine):
    """`bench.py --gpus 2` end to end (self-launch through torch.distributed.run, two rank processes on cuda:0, the library's collectives over
    gloo): the JSON line keeps `value` end to end and adds what makes an N > 1 curve readable -- the time rank 0 was blocked on the statement
    hash, the rest of the step, the hash-excluded rate, the look-ahead the ranks ran in the window, the time in the per-round exchanges."""
    import json
    import subprocess
    env = dict(os.environ, RIPP_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--log-n", "17"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["n"] == 1 << 17
    for k in ("hash_wait_ms", "statement_hash_ms", "exchange_ms", "look_ms"):
        assert k in out["phase_ms"], k
    assert out["gpu_phase_ms"] > 0 and abs(out["gpu_phase_ms"] - (out["ms_per_step"] - out["phase_ms"]["hash_wait_ms"])) < 1e-3
    assert abs(out["value_excl_hash"] - out["config"]["n"] / (out["gpu_phase_ms"] * 1e-3)) < 1e-6 * out["value_excl_hash"]
    assert out["value"] <= out["value_excl_hash"] * (1 + 1e-9)
    assert abs(out["post_hash_ms"] - (out["ms_per_step"] - out["phase_ms"]["statement_hash_ms"])) < 2e-3
    assert set(out["look_ahead"]) >= {"items", "pairs"}
    assert "cpu_baseline" not in out                                  # rank 0 at N = 1 only
