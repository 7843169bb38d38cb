"""Host-side plumbing of the multi-GPU provers.  PRODUCTION PATH: the NATIVE drivers at the bottom of this file
(NativeComm / ReplayComm + native_sipp_prove / native_sipp_job_prove / native_pairing_inner_product / native_msm): round loop and collectives
live inside libripp_hip.so, Python only hands the RCCL id over.  The Python-level prover at the top (TorchComm, ShardedSippProver,
sharded_pairing_inner_product, sharded_msm) is the CPU TEST VEHICLE of tests/test_sharded_gloo.py: the same protocol with an injectable
oracle-backed job, so the N > 1 control flow runs on boxes without a GPU (gloo, world sizes 2 / 4 / 8).

One process per GPU, vectors sharded by index residue, partial GT products combined by all-gather + local multiply.

Sharding: global element i lives on rank i mod G at local index i div G.  Every halving round pairs (i, i + L/2);
while L/2 is a multiple of G both partners have the same residue, so the fold is 100% local and each rank simply
halves its own shard with the common challenge.  The only exchange per round is the rank's two partial Miller values
(2 x 576 B): all-gather + local multiply (RCCL has no Fq12-product reduction op); each rank folds its own 68 per-step
products first, which is valid because the f <- f^2 * L recurrence is multiplicative.  When every rank is
down to ONE element (L == G) the G remaining elements are all-gathered and the last log2(G) rounds run replicated.

The engine (device job) and the communicator are injected, so the N > 1 control flow is testable on CPU with gloo
and an oracle-backed stand-in (tests/test_sharded_gloo.py); production uses ripp_amd.api.SippJob + TorchComm.
"""
import os
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


class HipPrimitives:
    """The three device calls the sharded inner products need (ripp_amd.api over the C ABI); tests inject an oracle-backed stand-in."""

    def __init__(self):
        from . import api
        self.api = api

    def pairing_miller(self, left, right):
        import ctypes
        from ._lib import lib
        l, r = self.api._c(left, 18), self.api._c(right, 36); out = np.zeros(72, dtype=np.uint64)
        self.api._check(lib().ripp_pairing_miller_j(self.api._p(l), ctypes.c_size_t(len(l)), self.api._p(r), ctypes.c_size_t(len(r)), self.api._p(out)), len(l), len(r))
        return out

    def final_exp(self, f): return self.api.final_exponentiation(f)
    def gt_mul(self, a, b): return self.api.gt_mul(a, b)
    def msm_g1(self, bases, scalars): return self.api.MultiexponentiationInnerProductG1.inner_product(bases, scalars)
    def msm_g2(self, bases, scalars): return self.api.MultiexponentiationInnerProductG2.inner_product(bases, scalars)

    def sum_points(self, pts, cols):
        import ctypes
        from ._lib import lib
        pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, cols); out = np.zeros(cols, dtype=np.uint64)
        fn = lib().ripp_sum_g1_j if cols == 18 else lib().ripp_sum_g2_j
        self.api._check(fn(self.api._p(pts), ctypes.c_size_t(len(pts)), self.api._p(out))); return out


def sharded_pairing_inner_product(comm, left_shard, right_shard, prim=None):
    """PairingInnerProduct::inner_product (inner_products/src/lib.rs:52-75) over vectors sharded by index residue: every rank
    evaluates the Miller loops of ITS pairs, the 576-byte Miller values are all-gathered and multiplied, and the single final
    exponentiation is applied to the product (replicated: every rank returns the same GT value)."""
    prim = prim or HipPrimitives()
    parts = comm.all_gather(prim.pairing_miller(left_shard, right_shard))
    acc = parts[0]
    for p in parts[1:]:
        acc = prim.gt_mul(acc, p)
    return prim.final_exp(acc)


def sharded_msm(comm, bases_shard, scalars_shard, group="g1", prim=None):
    """MultiexponentiationInnerProduct::inner_product (inner_products/src/lib.rs:119-142) over sharded vectors: one Pippenger MSM per
    rank, all-gather of the partial sums (one projective point each), G-1 additions."""
    prim = prim or HipPrimitives()
    cols = 18 if group == "g1" else 36
    part = (prim.msm_g1 if group == "g1" else prim.msm_g2)(bases_shard, scalars_shard)
    return prim.sum_points(np.stack(comm.all_gather(np.ascontiguousarray(part, dtype=np.uint64).reshape(cols))), cols)


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


# ---------------------------------------------------------------------------------------------------------------------------------
# Native path: the round loop and the collective live INSIDE libripp_hip.so (comm_api.inc: ripp_comm_*, ripp_sipp_*_prove_sharded),
# so that a non-Python host binds one call.  torch.distributed is used here only as the RENDEZVOUS (to hand rank 0's RCCL id to the
# other ranks) or, with transport="callback", as the all-gather the library calls back into (gloo: CPU boxes / several ranks on one GPU).
class NativeComm:
    """Initialises the library's communicator for this process.  transport: "rccl" (one GPU per rank, xGMI) or "callback"."""

    def __init__(self, transport=None):
        import ctypes
        import torch.distributed as dist
        from ._lib import lib
        from . import api
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()
        if transport is None:
            transport = "rccl" if dist.get_backend() == "nccl" else "callback"
        self.transport = transport
        if transport == "rccl":
            import torch
            on_gpu = dist.get_backend() == "nccl"
            ok = 1
            try:
                ident = np.zeros(128, dtype=np.uint8)
                if os.environ.get("RIPP_COMM_NO_RCCL"):        # (tests: take the fallback below without needing a machine where RCCL really fails)
                    raise RuntimeError("RIPP_COMM_NO_RCCL is set")
                if self.rank == 0:
                    api._check(lib().ripp_comm_unique_id(api._p(ident)))
            except Exception as exc:          # librccl not loadable beside this process's HIP runtime: every rank must learn it
                ok = 0; self.rccl_error = str(exc)
            flag = torch.tensor([ok], dtype=torch.int32); flag = flag.cuda() if on_gpu else flag
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()):
                t = torch.from_numpy(ident); t = t.cuda() if on_gpu else t
                dist.broadcast(t, src=0)
                ident = t.cpu().numpy().copy()
                try:
                    api._check(lib().ripp_comm_init(api._p(ident), ctypes.c_int32(self.rank), ctypes.c_int32(self.world)))
                except Exception as exc:
                    ok = 0; self.rccl_error = str(exc)
                flag = torch.tensor([ok], dtype=torch.int32); flag = flag.cuda() if on_gpu else flag
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if not int(flag.item()):
                # The library's own communicator could not be brought up on every rank: run its collectives through the host's process
                # group instead (same all-gather + local multiply, the 576-byte partials just take the torch.distributed route).
                lib().ripp_comm_destroy()
                self.transport = transport = "callback"
                if self.rank == 0:
                    import sys
                    print("[ripp] native RCCL communicator unavailable (%s): collectives go through torch.distributed" % getattr(self, "rccl_error", "another rank failed"), file=sys.stderr)
        if transport == "callback":
            import torch
            on_gpu = dist.get_backend() == "nccl"
            FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)

            def allgather(_user, send, recv, nbytes):
                try:
                    src = np.ctypeslib.as_array(ctypes.cast(send, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,))
                    t = torch.from_numpy(src.copy()); t = t.cuda() if on_gpu else t
                    outs = [torch.empty_like(t) for _ in range(self.world)]
                    dist.all_gather(outs, t)
                    dst = np.ctypeslib.as_array(ctypes.cast(recv, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes * self.world,))
                    dst[:] = torch.cat(outs).cpu().numpy()
                    return 0
                except Exception:          # never unwind through the C frames
                    return 1
            self._cb = FN(allgather)                 # keep the trampoline alive as long as the communicator
            api._check(lib().ripp_comm_init_callback(ctypes.c_int32(self.rank), ctypes.c_int32(self.world), self._cb, None))

    def close(self):
        from ._lib import lib
        lib().ripp_comm_destroy()

    # measurement rig (include/ripp_hip.h, transport "replay"): keep the exchanges of the proofs that follow / write the last proof's to a file
    def record(self, on=True):
        from ._lib import lib
        from . import api
        api._check(lib().ripp_comm_record(1 if on else 0))

    def save_recording(self, path):
        from ._lib import lib
        from . import api
        api._check(lib().ripp_comm_recording_save(os.fsencode(path)))


class ReplayComm:
    """ripp_comm_init_replay: THIS process is rank `rank` of `world`, alone on its GPU; the peers' blocks of every all-gather come from the
    recording at `path` and arrive when the recorded gaps (where measured) say the slowest peer would have.  No torch.distributed involved."""

    transport = "replay"

    def __init__(self, rank, world, path, latency_us=0.0):
        import ctypes
        from ._lib import lib
        from . import api
        self.rank, self.world = rank, world
        api._check(lib().ripp_comm_init_replay(ctypes.c_int32(rank), ctypes.c_int32(world), os.fsencode(path), ctypes.c_double(latency_us)))

    def info(self):
        import ctypes
        from ._lib import lib
        served, differs, waited = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_double()
        lib().ripp_comm_replay_info(ctypes.byref(served), ctypes.byref(differs), ctypes.byref(waited))
        return {"exchanges_served": served.value, "own_blocks_differing": differs.value, "waited_ms": waited.value}

    def check(self):
        """ripp_comm_replay_check: own blocks whose bytes differed from the recording are compared by meaning; raises when one MEANS something else"""
        import ctypes
        from ._lib import lib
        from . import api
        differing, mismatches = ctypes.c_uint64(), ctypes.c_uint64()
        api._check(lib().ripp_comm_replay_check(ctypes.byref(differing), ctypes.byref(mismatches)))
        return {"own_blocks_bytewise_differing": differing.value, "own_blocks_semantic_mismatches": mismatches.value}

    save_recording = NativeComm.save_recording
    close = NativeComm.close


def read_recording(path):
    """Parse a recording file: (world, [(bytes_per_rank, gaps[world] (NaN = not measured), blocks (world, bytes) uint8), ...])."""
    raw = open(path, "rb").read()
    assert raw[:8] == b"RIPPREC1", path
    world, count = (int(v) for v in np.frombuffer(raw, dtype=np.uint32, count=2, offset=8))
    off, out = 16, []
    for _ in range(count):
        nbytes = int(np.frombuffer(raw, dtype=np.uint32, count=1, offset=off)[0]); off += 8
        gaps = np.frombuffer(raw, dtype=np.float64, count=world, offset=off).copy(); off += 8 * world
        blocks = np.frombuffer(raw, dtype=np.uint8, count=world * nbytes, offset=off).reshape(world, nbytes).copy(); off += world * nbytes
        out.append((nbytes, gaps, blocks))
    assert off == len(raw), path
    return world, out


def write_recording(path, world, exchanges):
    with open(path, "wb") as f:
        f.write(b"RIPPREC1"); f.write(np.array([world, len(exchanges)], dtype=np.uint32).tobytes())
        for nbytes, gaps, blocks in exchanges:
            f.write(np.array([nbytes, 0], dtype=np.uint32).tobytes()); f.write(np.asarray(gaps, dtype=np.float64).tobytes()); f.write(np.ascontiguousarray(blocks, dtype=np.uint8).tobytes())


def native_sipp_job_prove(job, value, full=None, seed_digest=None):
    """ripp_sipp_job_prove_sharded: SIPP::prove (sipp/src/lib.rs:42-106) across the library's communicator on a resident shard.
    job: api.SippJob created with this rank / world; rank 0 passes full = (a, b, r) of the whole statement or seed_digest.
    Returns (proof (2*rounds,72), challenges (rounds,4), stats)."""
    import ctypes
    from ._lib import lib, RippStats
    from . import api
    world = int(lib().ripp_comm_world())
    n = job.n_local * world; lg = n.bit_length() - 1
    value = np.ascontiguousarray(value, dtype=np.uint64).reshape(72)
    proof = np.zeros((2 * max(lg, 1), 72), dtype=np.uint64); ch = np.zeros((max(lg, 1), 4), dtype=np.uint64); st = RippStats()
    z = ctypes.c_void_p(None)
    fa = fb = fr = z
    if full is not None:
        fa_, fb_, fr_ = api._c(full[0], 12), api._c(full[1], 24), api._c(full[2], 4)
        assert len(fa_) == len(fb_) == len(fr_) == n
        fa, fb, fr = api._p(fa_), api._p(fb_), api._p(fr_)
    dg = (ctypes.c_uint8 * 32).from_buffer_copy(seed_digest) if seed_digest is not None else z
    api._check(lib().ripp_sipp_job_prove_sharded(job._h, api._p(value), fa, fb, fr, dg, api._p(proof), api._p(ch), ctypes.byref(st)))
    return proof[: 2 * lg], ch[:lg], st.as_dict()


def native_sipp_prove(a_shard, b_shard, r_shard, value, full=None, seed_digest=None):
    """ripp_sipp_prove_sharded: the one-shot form on HOST slices -- this rank's shard is uploaded inside the call, rank 0's statement hash
    starts on `full` = (a, b, r) of the whole statement before that upload.  Returns (proof, challenges, stats)."""
    import ctypes
    from ._lib import lib, RippStats
    from . import api
    a, b, r = api._c(a_shard, 12), api._c(b_shard, 24), api._c(r_shard, 4)
    assert len(a) == len(b) == len(r)
    world = int(lib().ripp_comm_world())
    n = len(a) * world; lg = n.bit_length() - 1
    value = np.ascontiguousarray(value, dtype=np.uint64).reshape(72)
    proof = np.zeros((2 * max(lg, 1), 72), dtype=np.uint64); ch = np.zeros((max(lg, 1), 4), dtype=np.uint64); st = RippStats()
    z = ctypes.c_void_p(None)
    fa = fb = fr = z
    if full is not None:
        fa_, fb_, fr_ = api._c(full[0], 12), api._c(full[1], 24), api._c(full[2], 4)
        assert len(fa_) == len(fb_) == len(fr_) == n
        fa, fb, fr = api._p(fa_), api._p(fb_), api._p(fr_)
    dg = (ctypes.c_uint8 * 32).from_buffer_copy(seed_digest) if seed_digest is not None else z
    api._check(lib().ripp_sipp_prove_sharded(api._p(a), api._p(b), api._p(r), ctypes.c_size_t(len(a)), api._p(value), fa, fb, fr, dg, api._p(proof), api._p(ch), ctypes.byref(st)))
    return proof[: 2 * lg], ch[:lg], st.as_dict()


def native_pairing_inner_product(left_shard, right_shard):
    """PairingInnerProduct::inner_product over vectors sharded by index residue, collective inside the library."""
    import ctypes
    from ._lib import lib
    from . import api
    l, r = api._c(left_shard, 18), api._c(right_shard, 36); out = np.zeros(72, dtype=np.uint64)
    api._check(lib().ripp_pairing_product_sharded_j(api._p(l), ctypes.c_size_t(len(l)), api._p(r), ctypes.c_size_t(len(r)), api._p(out)), len(l), len(r))
    return out


def native_msm(bases_shard, scalars_shard, group="g1"):
    import ctypes
    from ._lib import lib
    from . import api
    cols = 18 if group == "g1" else 36
    b, s = api._c(bases_shard, cols), api._c(scalars_shard, 4); out = np.zeros(cols, dtype=np.uint64)
    fn = lib().ripp_msm_g1_sharded_j if group == "g1" else lib().ripp_msm_g2_sharded_j
    api._check(fn(api._p(b), ctypes.c_size_t(len(b)), api._p(s), ctypes.c_size_t(len(s)), api._p(out)), len(b), len(s))
    return out
