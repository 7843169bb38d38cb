"""Randomized stress of the sharded prover (one round loop for every world size, look-ahead with shared G2 chains, pipelined tail): WORLD ranks
(2, 4 or 8) on ONE GPU (callback transport over gloo), many proofs of random statements with random look-ahead plans against the CPU oracle;
also the one-rank form with the same plans.
  python tools/stress_sharded.py <seed> <seconds> <min log n> <max log n> [world = 2]"""
import os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def worker(rank, world, port, seed, seconds, lo, hi, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), RIPP_RANKS_PER_DEVICE=str(world), OMP_NUM_THREADS=str(max(1, 16 // world)))
    import torch.distributed as dist
    import orclib as o
    import ripp_amd as R
    from ripp_amd.sharded import NativeComm, shard, native_sipp_job_prove
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R.init(0)
    comm = NativeComm("callback")
    rng = np.random.default_rng(seed)            # the same stream on both ranks: same statements, same plans
    bad = cnt = 0; t0 = time.time()
    while True:
        go = np.array([1 if time.time() - t0 < seconds else 0], dtype=np.int64)
        import torch
        t = torch.from_numpy(go); dist.broadcast(t, src=0)
        if int(t.item()) == 0: break
        lg = int(rng.integers(max(lo, world.bit_length() - 1), hi + 1)); n = 1 << lg          # n >= world: every rank holds at least one element
        sa, sb, sr = (int(x) for x in rng.integers(1, 1 << 30, 3))
        plan = int(rng.integers(0, 49))
        a, b, r = o.gen_g1(sa, n), o.gen_g2(sb, n), o.gen_scalars(sr, n)
        if rng.random() < 0.2: a[int(rng.integers(0, n))] = 0
        if rng.random() < 0.2: b[int(rng.integers(0, n))] = 0
        v = o.product_of_pairings_with_coeffs(a, b, r)
        rc, ep, ech = o.sipp_prove(a, b, r, v)
        if rng.random() < 0.7: os.environ["RIPP_LOOK_EIGHTHS"] = str(plan)
        else: os.environ.pop("RIPP_LOOK_EIGHTHS", None)
        job = R.SippJob(shard(a, rank, world), shard(b, rank, world), shard(r, rank, world), rank=rank, world=world)
        proof, ch, _ = native_sipp_job_prove(job, v, full=(a, b, r) if rank == 0 else None)
        job.close()
        ok = rc == 0 and np.array_equal(proof, ep) and np.array_equal(ch, ech)
        cnt += 1
        if not ok: bad += 1; print("MISMATCH rank", rank, n, sa, sb, sr, plan, flush=True)
    os.environ.pop("RIPP_LOOK_EIGHTHS", None)
    comm.close()
    ret[rank] = (cnt, bad)
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1; seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 1; hi = int(sys.argv[4]) if len(sys.argv) > 4 else 13
    world = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(worker, args=(world, port, seed, seconds, lo, hi, ret), nprocs=world, join=True)
    print(world, "ranks (proofs, mismatches) per rank:", dict(ret))
    # one rank, same generator of statements and plans
    import orclib as o, ripp_amd as R
    R.init(0)
    rng = np.random.default_rng(seed + 1); bad = cnt = 0; t0 = time.time()
    while time.time() - t0 < seconds / 2:
        lg = int(rng.integers(max(lo, 1), hi + 1)); n = 1 << lg
        sa, sb, sr = (int(x) for x in rng.integers(1, 1 << 30, 3))
        a, b, r = o.gen_g1(sa, n), o.gen_g2(sb, n), o.gen_scalars(sr, n)
        v = o.product_of_pairings_with_coeffs(a, b, r)
        rc, ep, _ = o.sipp_prove(a, b, r, v)
        os.environ["RIPP_LOOK_EIGHTHS"] = str(int(rng.integers(0, 49)))
        p = R.SIPP.prove(a, b, r, v)
        cnt += 1
        if not (rc == 0 and np.array_equal(p, ep)): bad += 1; print("MISMATCH one rank", n, sa, sb, sr, os.environ["RIPP_LOOK_EIGHTHS"], flush=True)
    print("one rank: proofs", cnt, "mismatches", bad)
