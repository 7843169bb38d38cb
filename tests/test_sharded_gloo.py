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


@pytest.mark.parametrize("world,n", [(2, 2), (2, 8), (2, 32), (4, 4), (4, 8), (4, 32), (8, 8), (8, 16), (8, 64)])
def test_sharded_prover_gloo_matches_single_process(world, n):
    """World sizes 2, 4 and 8 (config 4 names 8 GPUs), incl. n == world -- every rank starts with ONE element, the proof is the tail gather
    plus log2(world) replicated rounds -- and n == 2 * world (one sharded round)."""
    os.environ["OMP_NUM_THREADS"] = str(max(1, 8 // world))
    _run(world, n, use_gpu=False)


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


@pytest.mark.parametrize("world,n", [(2, 2), (2, 10), (4, 4), (4, 12), (8, 8), (8, 24)])
def test_sharded_inner_products_gloo(world, n):
    """PairingInnerProduct and both MSMs over residue-sharded vectors: all-gather of one Miller value / one point per rank, one final
    exponentiation -- equal to the unsharded result (host logic on CPU, oracle-backed primitives); 2, 4 and 8 ranks."""
    os.environ["OMP_NUM_THREADS"] = str(max(1, 8 // world))
    _run_ip(world, n, use_gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1000])
def test_sharded_inner_products_world2_real_engine(engine, n):
    _run_ip(2, n, use_gpu=True)


# ---------------------------------------------------------------- native driver: round loop + collective inside libripp_hip.so
def _native_worker(rank, world, port, cases, ret):
    """One rank of `world`, all on cuda:0: the library's round loop with the all-gather supplied by the host (gloo), for EVERY case of `cases`
    ((n, env) pairs: one process start-up, rendezvous and communicator serve the whole list -- the suite's ~40 native cases used to pay ~3 s of
    start-up each).  Hands what it computed back to the parent (ret[(rank, case index)]), which compares every rank's outputs with the oracle's on
    the unsharded vectors (ONE oracle run per case, not one per rank)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), RIPP_RANKS_PER_DEVICE=str(world))
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "model")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import ripp_amd as R
    from ripp_amd.sharded import NativeComm, shard, native_sipp_job_prove, native_pairing_inner_product, native_msm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        R.init(0)
        # the ranks share cuda:0: RCCL needs one device per rank, gloo carries the all-gather.  With RIPP_COMM_NO_RCCL (first case's env) the RCCL transport
        # is REQUESTED and its bring-up made to fail, so the ranks must agree on the fallback to the host's process group (what bench.py relies on)
        os.environ.update({k: v for k, v in cases[0][1].items() if k == "RIPP_COMM_NO_RCCL"})
        comm = NativeComm("rccl" if os.environ.get("RIPP_COMM_NO_RCCL") else "callback")
        assert comm.transport == "callback"
        for idx, (n, env) in enumerate(cases):
            os.environ.update(env)                                  # (the library re-reads its RIPP_* overrides at every call)
            try:
                a, b, r = R.synth_g1(123, n), R.synth_g2(456, n), R.synth_fr(7, n)
                value = R.product_of_pairings_with_coeffs(a, b, r)
                job = R.SippJob(shard(a, rank, world), shard(b, rank, world), shard(r, rank, world), rank=rank, world=world)
                proof, ch, st = native_sipp_job_prove(job, value, full=(a, b, r) if rank == 0 else None)
                # second proof on the same resident shard, digest precomputed by the host
                proof2, _, _ = native_sipp_job_prove(job, value, seed_digest=R.sipp_seed_digest(a, b, r, value) if rank == 0 else None)
                job.close()
                out = {"value": value, "proof": proof, "ch": ch, "proof2": proof2, "look_items": int(st["look_items"])}
                if not env.get("RIPP_TEST_SIPP_ONLY"):
                    import orclib as o
                    aj, bj, s = o.blind_g1(a, 1), o.blind_g2(b, 2), R.synth_fr(9, n)
                    out["ip"] = native_pairing_inner_product(shard(aj, rank, world), shard(bj, rank, world))
                    out["m1"] = native_msm(shard(aj, rank, world), shard(s, rank, world), "g1")
                    out["m2"] = native_msm(shard(bj, rank, world), shard(s, rank, world), "g2")
                ret[(rank, idx)] = out
            finally:
                for k in env:
                    os.environ.pop(k, None)
        comm.close()
    finally:
        dist.destroy_process_group()


SPAWN_LOG_ROOT = os.path.join(os.path.dirname(HERE), "gpurun_out", "spawn_failures")
# what a failed START-UP of the rank processes looks like (gloo rendezvous over a TCP store on 127.0.0.1): the only failures that get a second attempt
RENDEZVOUS_ERRORS = ("Address already in use", "EADDRINUSE", "TCPStore", "DistNetworkError", "DistStoreError", "Connection reset by peer", "Connection refused",
                     "connect() timed out", "Socket Timeout", "store->get", "Timed out waiting for clients", "timed out after")


def _entry(rank, fn, logdir, *args):
    """Child side of _spawn_once: everything the rank writes to stdout / stderr (Python tracebacks, the library's messages, HIP / HSA runtime
    aborts) goes to logdir/rank<k>.log, so that a failure that happens once in a hundred runs leaves its evidence behind."""
    fd = os.open(os.path.join(logdir, f"rank{rank}.log"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    sys.stdout.flush(); sys.stderr.flush()
    os.dup2(fd, 1); os.dup2(fd, 2); os.close(fd)
    import faulthandler
    faulthandler.enable()                                   # a SIGSEGV / SIGABRT inside the library leaves the Python stack of the call in the log
    fn(rank, *args)


class SpawnFailure(Exception):
    def __init__(self, msg, logdir, exitcodes, text):
        super().__init__(msg); self.logdir, self.exitcodes, self.text = logdir, exitcodes, text


def _spawn_once(fn, world, args, timeout):
    import shutil
    import tempfile
    import time
    import torch.multiprocessing as mp
    os.makedirs(SPAWN_LOG_ROOT, exist_ok=True)
    logdir = tempfile.mkdtemp(prefix=time.strftime("%H%M%S_") + f"{fn.__name__}_w{world}_", dir=SPAWN_LOG_ROOT)
    mgr = mp.Manager(); ret = mgr.dict()
    ctx = mp.spawn(_entry, args=(fn, logdir, world, _free_port()) + tuple(args) + (ret,), nprocs=world, join=False)
    deadline = time.time() + timeout
    err = None
    try:
        while not ctx.join(timeout=5):
            if time.time() > deadline:
                err = TimeoutError(f"{world} ranks did not finish within {timeout} s: a rank hangs in a collective (logs: {logdir})")
                break
    except Exception as exc:                                # ProcessRaisedException / ProcessExitedException of torch.multiprocessing
        err = exc
    finally:
        for pr in ctx.processes:
            if pr.is_alive():
                pr.kill()
        for pr in ctx.processes:
            pr.join(10)
    if err is None:
        shutil.rmtree(logdir, ignore_errors=True)
        return dict(ret)
    # keep the evidence: exit code of every rank + its output
    codes = [pr.exitcode for pr in ctx.processes]
    tails = []
    for rank in range(world):
        try:
            tails.append(f"---- rank {rank} (exit code {codes[rank]}) ----\n" + open(os.path.join(logdir, f"rank{rank}.log"), errors="replace").read()[-3000:])
        except OSError:
            tails.append(f"---- rank {rank} (exit code {codes[rank]}): no log ----")
    text = f"{type(err).__name__}: {err}\n" + "\n".join(tails)
    with open(os.path.join(logdir, "failure.txt"), "w") as f:
        f.write(f"exit codes: {codes}\n{text}\n")
    if isinstance(err, TimeoutError):
        raise err
    raise SpawnFailure(f"{world} ranks of {fn.__name__} failed; exit codes {codes}; logs kept in {logdir}\n{text[-6000:]}", logdir, codes, text)


def _spawn(fn, world, args, timeout=900):
    """mp.spawn with a deadline: a hung collective fails the test instead of the session (the children are ended by PID, never by pattern).
    Every rank's output and exit code are kept under gpurun_out/spawn_failures/ when anything goes wrong.  ONE retry with fresh processes and a
    fresh port is granted only to a failed START-UP: an error text that names the rendezvous (RENDEZVOUS_ERRORS), no rank killed by a signal, no
    assertion, no status from the library.  A crash of a rank (SIGSEGV / SIGABRT / a GPU memory fault), a non-zero library status, a wrong result
    and a hang all fail the test at once -- the blanket retry of build round 4 would have hidden exactly those."""
    try:
        return _spawn_once(fn, world, args, timeout)
    except SpawnFailure as exc:
        by_signal = any(c is not None and c < 0 for c in exc.exitcodes)
        rendezvous = any(k in exc.text for k in RENDEZVOUS_ERRORS)
        ours = any(k in exc.text for k in ("AssertionError", "DeviceError", "InnerProductError", "[ripp]", "libripp_hip", "HSA_STATUS", "Memory access fault", "hipError"))
        if by_signal or ours or not rendezvous:
            raise
        print(f"[test_sharded_gloo] {world} ranks: rendezvous failed ({exc.logdir}) -- one retry with fresh processes", file=sys.stderr)
        return _spawn_once(fn, world, args, timeout)


CASES_CALLBACK = [(2, 2), (2, 4), (2, 64), (2, 1 << 12), (4, 4), (4, 8), (4, 64), (4, 1 << 12), (4, 1 << 14), (8, 8), (8, 16), (8, 64), (8, 1 << 12), (8, 1 << 14)]
CASES_LOOKAHEAD = [(2, 16, {"RIPP_LOOK_ITEMS": "6"}), (2, 64, {"RIPP_LOOK_ITEMS": "4"}), (2, 1 << 12, {"RIPP_LOOK_ITEMS": "6"}), (2, 1 << 14, {"RIPP_LOOK_ITEMS": "3"}),
                   (4, 64, {"RIPP_LOOK_EIGHTHS": "48"}), (4, 1 << 12, {"RIPP_LOOK_EIGHTHS": "45"}), (4, 1 << 14, {"RIPP_LOOK_EIGHTHS": "45"}),
                   (8, 128, {"RIPP_LOOK_EIGHTHS": "48"}), (8, 1 << 14, {"RIPP_LOOK_EIGHTHS": "48"}), (8, 1 << 14, {"RIPP_LOOK_EIGHTHS": "45"}), (8, 1 << 15, {"RIPP_LOOK_EIGHTHS": "29"})]
_NATIVE_BATCH = {}          # world -> (list of (n, env) cases, results of ONE spawn that ran them all, or the exception it died of)


def _case_key(n, env):
    return (n, tuple(sorted(env.items())))


def _native_batch(world):
    """Every native case of this world size (both parametrised tests) in ONE spawn; the first test that needs a result pays for it."""
    if world not in _NATIVE_BATCH:
        cases = [(n, {}) for w, n in CASES_CALLBACK if w == world] + [(n, dict(env, RIPP_TEST_SIPP_ONLY="1")) for w, n, env in CASES_LOOKAHEAD if w == world]
        try:
            got = _spawn(_native_worker, world, (cases,), timeout=1500)
        except BaseException as exc:          # every case of this world fails with the same evidence
            got = exc
        _NATIVE_BATCH[world] = (cases, got)
    return _NATIVE_BATCH[world]


def _run_native(orc, world, n, env=None, sipp_only=False, batch=True):
    env = dict(env or {})
    if sipp_only:
        env["RIPP_TEST_SIPP_ONLY"] = "1"
    if batch:
        cases, res = _native_batch(world)
        if isinstance(res, BaseException):
            raise res
        idx = [_case_key(*c) for c in cases].index(_case_key(n, env))
    else:
        res, idx = _spawn(_native_worker, world, ([(n, env)],)), 0
    got = {rank: res[(rank, idx)] for rank in range(world) if (rank, idx) in res}
    assert sorted(got) == list(range(world)), sorted(got)
    o = orc
    a, b, r = o.gen_g1(123, n), o.gen_g2(456, n), o.gen_scalars(7, n)
    value = o.product_of_pairings_with_coeffs(a, b, r)
    rc, eproof, ech = o.sipp_prove(a, b, r, value)
    assert rc == 0
    if not sipp_only:
        aj, bj, s = o.blind_g1(a, 1), o.blind_g2(b, 2), o.gen_scalars(9, n)
        eip = o.pairing_product_j(aj, bj)[1]; em1 = o.g1_to_affine(o.msm_g1_j(aj, s)[1]); em2 = o.g2_to_affine(o.msm_g2_j(bj, s)[1])
    for rank in range(world):
        g = got[rank]
        assert np.array_equal(g["value"], value), rank
        assert np.array_equal(g["proof"], eproof) and np.array_equal(g["ch"], ech), f"rank {rank}: proof differs from the oracle's"
        assert np.array_equal(g["proof2"], eproof), f"rank {rank}: second proof on the resident shard differs"
        if not sipp_only:
            assert np.array_equal(g["ip"], eip), rank
            assert np.array_equal(o.g1_to_affine(g["m1"]), em1) and np.array_equal(o.g2_to_affine(g["m2"]), em2), rank
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("world,n", CASES_CALLBACK)
def test_native_sharded_driver_callback_transport(engine, orc, world, n):
    """ripp_sipp_job_prove_sharded / ripp_*_sharded_j: the library's own round loop and collectives with 2, 4 and 8 ranks on cuda:0, the
    all-gather supplied by the host (gloo): proofs, pairing product and MSMs of EVERY rank equal the oracle's on the unsharded vectors.
    n == world: every rank starts with one element (tail gather first, log2(world) replicated rounds -- three for 8 ranks);
    n == 2 * world: one sharded round.  Reference loop: sipp/src/lib.rs:69-104."""
    _run_native(orc, world, n)


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,env", CASES_LOOKAHEAD)
def test_native_sharded_driver_with_lookahead(engine, orc, world, n, env):
    """The sharded prover with the multi-GPU look-ahead plans forced onto small statements: every rank pre-evaluates the values of rounds
    1..3 from ITS shard's round-0 blocks, reduces them with the challenges and contributes the partial GT values; folds of those rounds are
    not waited for.  RIPP_LOOK_EIGHTHS = 45 / 48: the plans `look_plan` picks for 4 / 8 ranks at n = 2^20 (rounds 1-2 + (3,l) + 5/8 of (3,r);
    rounds 1-3 in full).  Proofs of every rank equal the oracle's on the unsharded vectors."""
    got = _run_native(orc, world, n, env, sipp_only=True)
    eighths = 8 * min(6, int(env["RIPP_LOOK_ITEMS"])) if "RIPP_LOOK_ITEMS" in env else min(48, int(env["RIPP_LOOK_EIGHTHS"]))
    nl, want = n // world, 0
    for it in range((eighths + 7) // 8):                            # job_lookahead's static plan: items (1,l) (1,r) (2,l) .. on blocks of nl >> (R + 1) pairs
        qblk, frac = nl >> (it // 2 + 2), min(8, eighths - 8 * it)
        q = qblk if frac >= 8 else (qblk * frac // 8) & ~63         # a partial item is cut to a multiple of 64 pairs
        if q == 0:
            break
        want += 1
    for rank in range(world):
        assert got[rank]["look_items"] == want, (rank, got[rank]["look_items"], want)


@pytest.mark.gpu
def test_native_comm_falls_back_to_the_process_group(engine, orc):
    """bench.py asks for the library's own RCCL communicator; when that cannot be brought up on every rank the ranks agree (one all-reduce)
    to run the library's collectives through torch.distributed instead.  Forced here with RIPP_COMM_NO_RCCL on two ranks sharing cuda:0."""
    _run_native(orc, 2, 64, {"RIPP_COMM_NO_RCCL": "1"}, batch=False)


# ---------------------------------------------------------------- collective error exit
def _fail_worker(rank, world, port, n, fail_rank, fail_round, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), RIPP_RANKS_PER_DEVICE=str(world))
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "model")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import ctypes
    import torch.distributed as dist
    import ripp_amd as R
    from ripp_amd._lib import lib
    from ripp_amd.sharded import NativeComm, shard, native_sipp_job_prove
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        R.init(0)
        comm = NativeComm("callback")
        a, b, r = R.synth_g1(123, n), R.synth_g2(456, n), R.synth_fr(7, n)
        value = R.product_of_pairings_with_coeffs(a, b, r)
        job = R.SippJob(shard(a, rank, world), shard(b, rank, world), shard(r, rank, world), rank=rank, world=world)
        lib().ripp_test_inject_failure.restype = None
        if rank == fail_rank:
            lib().ripp_test_inject_failure(ctypes.c_int32(fail_rank), ctypes.c_int32(fail_round))
        err = None
        try:
            native_sipp_job_prove(job, value, full=(a, b, r) if rank == 0 else None)
        except Exception as exc:
            err = str(exc)
        # the protocol is in step again (every rank left at the same exchange): the NEXT proof on the same communicator and the same
        # resident shard succeeds -- no fresh processes needed, nothing re-executed
        proof, ch, _ = native_sipp_job_prove(job, value, full=(a, b, r) if rank == 0 else None)
        job.close(); comm.close()
        ret[rank] = {"err": err, "proof": proof, "ch": ch}
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,fail_rank,fail_round", [(2, 64, 1, 0), (4, 1 << 12, 2, 3), (8, 1 << 12, 5, 0), (8, 1 << 12, 0, 4), (8, 64, 7, 2), (4, 1 << 14, 3, 12)])
def test_collective_error_exit_no_rank_hangs(engine, orc, world, n, fail_rank, fail_round):
    """ONE rank fails locally in the fold of round r (ripp_test_inject_failure).  Every message of the protocol carries the sender's status, so
    the failing rank keeps walking the skeleton until the next exchange has told the others: EVERY rank returns an error (nothing hangs: the
    spawn has a deadline), and the next proof on the same communicator equals the oracle's.  Last case: the failing round lies in the
    replicated tail (after the gather) -- a local matter of that rank, the others finish their proof."""
    got = _spawn(_fail_worker, world, (n, fail_rank, fail_round), timeout=600)
    assert sorted(got) == list(range(world))
    sharded_rounds = (n // world).bit_length() - 1                  # rounds whose exchange is collective
    a, b, r = orc.gen_g1(123, n), orc.gen_g2(456, n), orc.gen_scalars(7, n)
    value = orc.product_of_pairings_with_coeffs(a, b, r)
    rc, eproof, ech = orc.sipp_prove(a, b, r, value)
    for rank in range(world):
        if fail_round < sharded_rounds:
            assert got[rank]["err"], f"rank {rank} returned success although rank {fail_rank} failed in round {fail_round}"
            assert ("injected failure" in got[rank]["err"]) == (rank == fail_rank), got[rank]["err"]
            if rank != fail_rank:
                assert f"rank {fail_rank} failed" in got[rank]["err"], got[rank]["err"]
        else:
            assert bool(got[rank]["err"]) == (rank == fail_rank), (rank, got[rank]["err"])
        assert np.array_equal(got[rank]["proof"], eproof) and np.array_equal(got[rank]["ch"], ech), f"rank {rank}: the proof after the failure differs"


# ---------------------------------------------------------------- communicator self-test, a rank that DIES
def _selftest_worker(rank, world, port, claim, ret):
    """ripp_comm_init_callback over gloo with rank `claim[rank]` announced to the library (no device needed: the callback transport is host code)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (os.path.dirname(HERE), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import ctypes
    import datetime
    import torch
    import torch.distributed as dist
    from ripp_amd._lib import lib
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)

        def allgather(_user, send, recv, nbytes):
            try:
                src = np.ctypeslib.as_array(ctypes.cast(send, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,))
                t = torch.from_numpy(src.copy()); outs = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(outs, t)
                np.ctypeslib.as_array(ctypes.cast(recv, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes * world,))[:] = torch.cat(outs).numpy()
                return 0
            except Exception:
                return 1
        cb = FN(allgather)
        L = lib(); L.ripp_last_error.restype = ctypes.c_char_p
        rc = L.ripp_comm_init_callback(ctypes.c_int32(claim[rank]), ctypes.c_int32(world), cb, None)
        ret[rank] = {"rc": rc, "err": L.ripp_last_error().decode() if rc else "", "world": L.ripp_comm_world()}
        L.ripp_comm_destroy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,claim", [(2, (0, 1)), (2, (0, 0)), (4, (0, 1, 3, 2))])
def test_comm_selftest_names_the_miswired_rank(world, claim):
    """ripp_comm_init[_callback] ends with ONE all-gather of (rank, world, ABI version, build fingerprint) that every rank checks: a bring-up in which two
    processes claim the same rank, or the ranks are permuted against the transport's order, is an error STRING on every rank -- not a first proof that
    hangs or is wrong.  Runs without a GPU (callback transport over gloo)."""
    got = _spawn(_selftest_worker, world, (claim,), timeout=240)
    assert sorted(got) == list(range(world))
    for rank in range(world):
        if tuple(claim) == tuple(range(world)):
            assert got[rank]["rc"] == 0 and got[rank]["world"] == world, got[rank]
        else:
            assert got[rank]["rc"] == 4 and "self-test" in got[rank]["err"] and "introduced itself as rank" in got[rank]["err"], got[rank]
            assert got[rank]["world"] == 1                       # the communicator was not kept


def _kill_worker(rank, world, port, n, kill_rank, kill_round, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), RIPP_RANKS_PER_DEVICE=str(world))
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "model")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import ctypes
    import datetime
    import time
    import torch.distributed as dist
    import ripp_amd as R
    from ripp_amd._lib import lib
    from ripp_amd.sharded import NativeComm, shard, native_sipp_job_prove
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=20))      # the callback transport's deadline is the process group's
    R.init(0)
    comm = NativeComm("callback")
    a, b, r = R.synth_g1(123, n), R.synth_g2(456, n), R.synth_fr(7, n)
    value = R.product_of_pairings_with_coeffs(a, b, r)
    job = R.SippJob(shard(a, rank, world), shard(b, rank, world), shard(r, rank, world), rank=rank, world=world)
    proof, ch, _ = native_sipp_job_prove(job, value, full=(a, b, r) if rank == 0 else None)           # a complete proof first: everything is in step
    lib().ripp_test_inject_failure.restype = None
    if rank == kill_rank:
        lib().ripp_test_inject_failure(ctypes.c_int32(kill_rank), ctypes.c_int32(1000 + kill_round))      # SIGKILL inside the library, in the fold of that round
    t0 = time.time(); err = None
    try:
        native_sipp_job_prove(job, value, full=(a, b, r) if rank == 0 else None)
    except Exception as exc:
        err = str(exc)
    ret[rank] = {"err": err, "seconds": time.time() - t0, "proof": proof}
    # no clean-up collectives: the group has lost a member (the process ends here)
    os._exit(0)


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,kill_rank,kill_round", [(2, 1 << 12, 1, 2), (4, 1 << 12, 2, 0)])
def test_killed_rank_ends_the_survivors_within_the_deadline(engine, orc, world, n, kill_rank, kill_round):
    """ONE rank is killed (SIGKILL, raised inside the library in the fold of round r) in the middle of a sharded proof.  It never sends its next block:
    the survivors' exchange fails when the transport's deadline passes (here gloo's 20 s; the RCCL transport polls its stream against
    ripp_config.comm_timeout_ms) and every survivor returns an error naming the exchange -- nothing waits for the driver's own limit."""
    import time
    import torch.multiprocessing as mp
    mgr = mp.Manager(); ret = mgr.dict()
    ctx = mp.spawn(_kill_worker, args=(world, _free_port(), n, kill_rank, kill_round, ret), nprocs=world, join=False)
    t0 = time.time()
    while any(pr.is_alive() for pr in ctx.processes) and time.time() - t0 < 150:
        time.sleep(0.5)
    alive = [pr.is_alive() for pr in ctx.processes]
    for pr in ctx.processes:
        if pr.is_alive():
            pr.kill()
    for pr in ctx.processes:
        pr.join(10)
    assert not any(alive), f"ranks still running after 150 s: {alive}"
    codes = [pr.exitcode for pr in ctx.processes]
    assert codes[kill_rank] == -9, codes
    got = dict(ret)
    rc, eproof, _ = orc.sipp_prove(orc.gen_g1(123, n), orc.gen_g2(456, n), orc.gen_scalars(7, n), orc.product_of_pairings_with_coeffs(orc.gen_g1(123, n), orc.gen_g2(456, n), orc.gen_scalars(7, n)))
    for rank in range(world):
        if rank == kill_rank:
            assert rank not in got
            continue
        assert codes[rank] == 0 and rank in got, (rank, codes)
        assert got[rank]["err"] and "all-gather" in got[rank]["err"] and f"rank {rank} of {world}" in got[rank]["err"], got[rank]["err"]
        assert got[rank]["seconds"] < 90, got[rank]["seconds"]
        assert np.array_equal(got[rank]["proof"], eproof)           # (the proof before the kill was the oracle's)


@pytest.mark.gpu
def test_bench_exits_non_zero_when_a_rank_dies(engine):
    """`bench.py --gpus 2` (ranks on cuda:0, collectives over gloo) with rank 1 killed in the middle of the timed proofs (RIPP_BENCH_KILL_RANK): the job
    ends with a non-zero exit code in well under 90 s -- never a re-exec, never a hang until the driver's limit."""
    import subprocess
    import time
    env = dict(os.environ, RIPP_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", RIPP_BENCH_KILL_RANK="1", RIPP_BENCH_DIST_TIMEOUT_S="20")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "14"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0, p.stdout[-2000:]
    assert time.time() - t0 < 90, time.time() - t0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]          # no result line from a job that lost a rank


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
_AGG_ORACLE = {}            # n -> the oracle's outputs on the (deterministic) instance of that size


def _agg_worker(rank, world, port, cases, ret):
    """One rank of the sharded GIPA / TIPP prover and of ripp_aggregate_proofs_sharded on the instances the parent prepared (cases: (path, env) pairs,
    all served by ONE process start-up / communicator); hands every output back to the parent as ret[(rank, case index)]."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), RIPP_RANKS_PER_DEVICE=str(world))
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "model")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import ripp_amd as R
    from ripp_amd._lib import AggregateProof
    from ripp_amd.sharded import NativeComm, shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        R.init(0)
        comm = NativeComm("callback")
        for idx, (path, env) in enumerate(cases):
            os.environ.update(env)
            try:
                d = np.load(path)
                sh = lambda k: shard(d[k], rank, world)
                steps, tr, (ba, bb), (ka, kb) = R.gipa_tipp_prove_sharded(sh("m_a"), sh("m_b"), sh("ck_a"), sh("ck_b"))
                out = {"steps": steps, "tr": tr, "ba": ba, "bb": bb, "ka": ka, "kb": kb}
                srs = R.SRS(d["gap"], d["hbp"], d["g_beta"], d["h_alpha"])
                got, _ = R.aggregate_proofs_sharded(srs, sh("a"), sh("b"), sh("c"))
                for k in AggregateProof.FIXED:
                    out["agg_" + k] = np.array(got.field(k))
                for k in AggregateProof.STEPS:
                    out["agg_" + k] = np.array(getattr(got, k))
                srs.close()
                ret[(rank, idx)] = out
            finally:
                for k in env:
                    os.environ.pop(k, None)
        comm.close()
    finally:
        dist.destroy_process_group()


CASES_AGG = [(2, 2, {}), (2, 8, {}), (2, 256, {}), (2, 1 << 14, {}), (4, 4, {}), (4, 8, {}), (4, 256, {}), (4, 1 << 14, {}), (8, 8, {}), (8, 16, {}), (8, 256, {}), (8, 1 << 14, {}),
             (2, 256, {"RIPP_AGG_SEQUENTIAL": "1"}), (8, 1 << 14, {"RIPP_AGG_SEQUENTIAL": "1"})]
_AGG_INSTANCE = {}          # n -> (path of the instance file, the arrays the checks need)
_AGG_BATCH = {}             # world -> (cases, results of the ONE spawn that ran them)


def _agg_instance(engine, o, n, tmpdir):
    if n not in _AGG_INSTANCE:
        import helpers as h
        m_a, m_b = o.blind_g1(engine.synth_g1(11, n), 1), o.blind_g2(engine.synth_g2(22, n), 2)
        ck_a, ck_b = o.blind_g2(engine.synth_g2(33, n), 3), o.blind_g1(engine.synth_g1(44, n), 4)
        osrs = h.make_srs(n, 0xa1fa + n, 0xbe7a + n)
        vk, pub, a, b, c = h.fake_groth16(n, 2, seed=n)
        path = os.path.join(tmpdir, f"instance_{n}.npz")
        np.savez(path, m_a=m_a, m_b=m_b, ck_a=ck_a, ck_b=ck_b, gap=osrs[0], hbp=osrs[1], g_beta=osrs[2], h_alpha=osrs[3], a=a, b=b, c=c)
        _AGG_INSTANCE[n] = (path, (m_a, m_b, ck_a, ck_b, osrs, vk, pub, a, b, c))
    return _AGG_INSTANCE[n]


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,env", CASES_AGG)
def test_sharded_gipa_and_aggregate(engine, orc, tmp_path_factory, world, n, env):
    """ripp_gipa_tipp_prove_sharded / ripp_aggregate_proofs_sharded with 2, 4 and 8 ranks (callback transport on cuda:0; n == world: every rank
    holds ONE proof): commitments of every round, transcripts, base cases, KZG openings and every member of the aggregate -- on EVERY rank --
    equal the oracle's on the unsharded vectors, and the oracle's verifier accepts.  2^14 is config 5's size (groth16_aggregation.rs:77-160).
    Default: the TIPP and TIPAWithSSM sub-proofs run SIDE BY SIDE on every rank, their k-th exchanges paired in one all-gather (CommMux, build
    round 5); RIPP_AGG_SEQUENTIAL=1: one after the other, as in build round 4 -- the same bytes.  All cases of one world size share ONE spawn."""
    import helpers as h
    from ripp_amd._lib import AggregateProof
    o = orc
    if world not in _AGG_BATCH:
        tmpdir = str(tmp_path_factory.mktemp(f"agg_w{world}"))
        cases = [(_agg_instance(engine, o, nn, tmpdir)[0], ee) for w, nn, ee in CASES_AGG if w == world]
        keys = [(nn, tuple(sorted(ee.items()))) for w, nn, ee in CASES_AGG if w == world]
        try:
            res = _spawn(_agg_worker, world, (cases,), timeout=1500)
        except BaseException as exc:
            res = exc
        _AGG_BATCH[world] = (keys, res)
    keys, res = _AGG_BATCH[world]
    if isinstance(res, BaseException):
        raise res
    idx = keys.index((n, tuple(sorted(env.items()))))
    got = {rank: res[(rank, idx)] for rank in range(world) if (rank, idx) in res}
    assert sorted(got) == list(range(world))
    m_a, m_b, ck_a, ck_b, osrs, vk, pub, a, b, c = _AGG_INSTANCE[n][1]
    if n not in _AGG_ORACLE:                                          # ONE oracle run per instance and session (6.5 s + 14 s at n = 2^14), shared by the world sizes
        rc, esteps, etr, eba, ebb, eka, ekb = o.gipa_tipp_prove(m_a, m_b, ck_a, ck_b)
        assert rc == 0
        rc, exp = o.aggregate_proofs(osrs[0], osrs[1], a, b, c)
        assert rc == 0
        _AGG_ORACLE[n] = (esteps, etr, eba, ebb, eka, ekb, exp)
    esteps, etr, eba, ebb, eka, ekb, exp = _AGG_ORACLE[n]
    rounds = n.bit_length() - 1
    for rank in range(world):
        g = got[rank]
        assert np.array_equal(g["steps"], esteps) and np.array_equal(g["tr"], etr), rank
        assert np.array_equal(o.g1_to_affine(g["ba"]), o.g1_to_affine(eba)) and np.array_equal(o.g2_to_affine(g["bb"]), o.g2_to_affine(ebb)), rank
        assert np.array_equal(o.g2_to_affine(g["ka"]), o.g2_to_affine(eka)) and np.array_equal(o.g1_to_affine(g["kb"]), o.g1_to_affine(ekb)), rank
        for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c"):
            assert np.array_equal(g["agg_" + k], exp.field(k)), (rank, k)
        for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript"):
            assert np.array_equal(g["agg_" + k], getattr(exp, k)), (rank, k)
        for k in ("agg_c", "ab_base_a", "ab_final_ck_b", "ab_opening_b", "c_base_a"):
            assert np.array_equal(o.g1_to_affine(g["agg_" + k]), o.g1_to_affine(exp.field(k))), (rank, k)
        for k in ("ab_base_b", "ab_final_ck_a", "ab_opening_a", "c_final_ck_a", "c_opening_a"):
            assert np.array_equal(o.g2_to_affine(g["agg_" + k]), o.g2_to_affine(exp.field(k))), (rank, k)
        assert np.array_equal(o.normalize_g1(g["agg_c_com_g1"][: 2 * rounds]), o.normalize_g1(exp.c_com_g1[: 2 * rounds])), rank
    # the oracle's verifier accepts what the ranks produced (rank world - 1's copy, rebuilt into the struct)
    pf = AggregateProof(n); g = got[world - 1]
    for k in AggregateProof.FIXED:
        pf.field(k)[...] = g["agg_" + k]
    for k in AggregateProof.STEPS:
        getattr(pf, k)[...] = g["agg_" + k]
    assert o.verify_aggregate_proof(h.verifier_srs(osrs), vk, pub, pf) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 8])
def test_bench_ranks_on_one_device_reports_the_hash_excluded_figures(engine, gpus):
    """`bench.py --gpus 2 / 8` end to end (self-launch through torch.distributed.run, the rank processes on cuda:0, the library's collectives over
    gloo): the JSON line keeps `value` end to end and adds what makes an N > 1 curve readable -- the time rank 0 was blocked on the statement
    hash, the rest of the step, the hash-excluded rate, the look-ahead the ranks ran in the window, the time in the per-round exchanges."""
    import json
    import subprocess
    env = dict(os.environ, RIPP_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "1", "--log-n", "17"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == gpus and out["scaling"] == "strong" and out["config"]["n"] == 1 << 17 and out["config"]["sharding"] == "index residue mod %d" % gpus
    for k in ("hash_wait_ms", "statement_hash_ms", "exchange_ms", "look_ms"):
        assert k in out["phase_ms"], k
    assert out["gpu_phase_ms"] > 0 and abs(out["gpu_phase_ms"] - (out["ms_per_step"] - out["phase_ms"]["hash_wait_ms"])) < 1e-3
    assert abs(out["value_excl_hash"] - out["config"]["n"] / (out["gpu_phase_ms"] * 1e-3)) < 1e-6 * out["value_excl_hash"]
    assert out["value"] <= out["value_excl_hash"] * (1 + 1e-9)
    assert abs(out["post_hash_ms"] - (out["ms_per_step"] - out["statement_hash_ms"])) < 2e-3 and out["statement_hash_ms"] > 0
    assert out["config"]["call"].startswith("ripp_sipp_prove_sharded on host slices") and out["value_resident"] > 0
    for k, v in out["roofline"]["int_alu"].items():                   # no fraction of a roof may exceed 1
        assert not k.startswith("frac") or 0 < v <= 1, (k, v)
    assert 0 < out["roofline"]["frac"] <= 1
    assert set(out["look_ahead"]) >= {"items", "pairs"}
    assert "cpu_baseline" not in out                                  # rank 0 at N = 1 only


@pytest.mark.gpu
def test_recorded_peer_transport_one_rank_alone_equals_the_oracle(engine, tmp_path):
    """Transport "replay" (ripp_comm_record / ripp_comm_init_replay; tools/replay_ranks.py): a two-rank proof is recorded on cuda:0, then rank 0 and
    rank 1 each prove ALONE with the peer's blocks served from the recording -- two passes each, the second one against the peer's measured gaps.
    The tool compares every recorded and every replayed proof with the CPU oracle's and fails otherwise."""
    import json
    import subprocess
    out = str(tmp_path / "replay")
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "replay_ranks.py"), "all", "--world", "2", "--log-n", "13", "--steps", "2", "--warmup", "1",
                        "--out-dir", out], capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-4000:]
    passes = json.load(open(os.path.join(out, "w2_n13_passes.json")))
    assert len(passes) == 4 and all(q["proof_equals_oracle"] and q["replay"]["exchanges_served"] > 0 for q in passes)
    assert [q["rank"] for q in passes] == [0, 1, 0, 1]
    # a GT value a rank sends is deterministic: only the plan message (timing fields) may differ from the recording, once per proof
    assert all(q["replay"]["own_blocks_differing"] <= q["steps"] + q["warmup"] for q in passes), [q["replay"] for q in passes]
    # the non-hashing rank waited for rank 0's digest in the second pass (rank 0's gaps were known by then)
    assert passes[1]["replay"]["waited_ms"] > 0
