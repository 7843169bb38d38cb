"""Stress of the 8-rank aggregate path (the one case of the suite that failed once, unexplained, in build round 4): ripp_gipa_tipp_prove_sharded +
ripp_aggregate_proofs_sharded with WORLD ranks on ONE GPU (callback transport over gloo), in two modes that separate the two suspects:

  respawn <runs> <n>          the test's own shape: every run spawns WORLD fresh processes (torch import, gloo rendezvous, library start-up) and
                              proves once -- a start-up / rendezvous fault shows here.  Every rank's stdout / stderr and exit code are kept on failure.
  resident <seconds> <n,n,..> WORLD processes are spawned ONCE and loop over the sizes for <seconds>: hundreds of proofs through the parked TIPA
                              buffers, the auxiliary engine, the side-by-side sub-proofs (CommMux) and the callback trampoline -- a race in the
                              PRODUCT shows here.  Every output of every rank is compared with the oracle's (computed once per size by the parent).

Both run under CPU load when --load K is given (K busy oracle threads beside the ranks: the failing run of round 4 was on a busy box).
  python tools/stress_agg_ranks.py respawn 20 16384 [--world 8] [--load 8]
  python tools/stress_agg_ranks.py resident 240 256,16384 [--world 8] [--load 8]"""
import argparse
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "model")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np


def make_instance(n, path):
    """the instance of tests/test_sharded_gloo.py::test_sharded_gipa_and_aggregate and the oracle's outputs on it"""
    import helpers as h
    import orclib as o
    import ripp_amd as R
    from ripp_amd._lib import AggregateProof
    m_a, m_b = o.blind_g1(R.synth_g1(11, n), 1), o.blind_g2(R.synth_g2(22, n), 2)
    ck_a, ck_b = o.blind_g2(R.synth_g2(33, n), 3), o.blind_g1(R.synth_g1(44, n), 4)
    osrs = h.make_srs(n, 0xa1fa + n, 0xbe7a + n)
    vk, pub, a, b, c = h.fake_groth16(n, 2, seed=n)
    rc, esteps, etr, eba, ebb, eka, ekb = o.gipa_tipp_prove(m_a, m_b, ck_a, ck_b)
    assert rc == 0
    rc, exp = o.aggregate_proofs(osrs[0], osrs[1], a, b, c)
    assert rc == 0
    d = dict(m_a=m_a, m_b=m_b, ck_a=ck_a, ck_b=ck_b, gap=osrs[0], hbp=osrs[1], g_beta=osrs[2], h_alpha=osrs[3], a=a, b=b, c=c, e_steps=esteps, e_tr=etr)
    for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c"):
        d["x_" + k] = np.array(exp.field(k))
    for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript"):
        d["x_" + k] = np.array(getattr(exp, k))
    for k in ("agg_c", "ab_base_a", "ab_final_ck_b", "ab_opening_b", "c_base_a"):
        d["x_" + k] = o.g1_to_affine(exp.field(k))
    for k in ("ab_base_b", "ab_final_ck_a", "ab_opening_a", "c_final_ck_a", "c_opening_a"):
        d["x_" + k] = o.g2_to_affine(exp.field(k))
    np.savez(path, **d)


def check(R, o, d, steps, tr, got):
    if not (np.array_equal(steps, d["e_steps"]) and np.array_equal(tr, d["e_tr"])):
        return "gipa steps / transcript"
    for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c"):
        if not np.array_equal(np.array(got.field(k)), d["x_" + k]):
            return k
    for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript"):
        if not np.array_equal(np.array(getattr(got, k)), d["x_" + k]):
            return k
    for k in ("agg_c", "ab_base_a", "ab_final_ck_b", "ab_opening_b", "c_base_a"):
        if not np.array_equal(o.g1_to_affine(np.array(got.field(k))), d["x_" + k]):
            return k
    for k in ("ab_base_b", "ab_final_ck_a", "ab_opening_a", "c_final_ck_a", "c_opening_a"):
        if not np.array_equal(o.g2_to_affine(np.array(got.field(k))), d["x_" + k]):
            return k
    return None


def worker(rank, world, port, paths, seconds, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), RIPP_RANKS_PER_DEVICE=str(world), OMP_NUM_THREADS="1")
    import torch
    import torch.distributed as dist
    import orclib as o
    import ripp_amd as R
    from ripp_amd.sharded import NativeComm, shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R.init(0)
    comm = NativeComm("callback")
    inst = []
    for p in paths:
        d = dict(np.load(p))
        inst.append((d, R.SRS(d["gap"], d["hbp"], d["g_beta"], d["h_alpha"])))
    cnt = bad = 0; t0 = time.time(); it = 0
    while True:
        go = torch.tensor([1 if (seconds <= 0 and it < len(inst)) or (seconds > 0 and time.time() - t0 < seconds) else 0], dtype=torch.int64)
        dist.broadcast(go, src=0)
        if int(go.item()) == 0:
            break
        d, srs = inst[it % len(inst)]; it += 1
        sh = lambda k: shard(d[k], rank, world)
        steps, tr, _, _ = R.gipa_tipp_prove_sharded(sh("m_a"), sh("m_b"), sh("ck_a"), sh("ck_b"))
        got, _ = R.aggregate_proofs_sharded(srs, sh("a"), sh("b"), sh("c"))
        why = check(R, o, d, steps, tr, got)
        cnt += 1
        if why:
            bad += 1; print(f"MISMATCH rank {rank} iteration {it} n {len(d['a'])}: {why}", flush=True)
    for _, srs in inst:
        srs.close()
    comm.close()
    ret[rank] = (cnt, bad)
    dist.destroy_process_group()


def entry(rank, logdir, *args):
    fd = os.open(os.path.join(logdir, f"rank{rank}.log"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    sys.stdout.flush(); sys.stderr.flush(); os.dup2(fd, 1); os.dup2(fd, 2); os.close(fd)
    import faulthandler
    faulthandler.enable()
    worker(rank, *args)


def spawn(world, paths, seconds, logroot, tag):
    import torch.multiprocessing as mp
    logdir = os.path.join(logroot, tag); os.makedirs(logdir, exist_ok=True)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    mgr = mp.Manager(); ret = mgr.dict()
    ctx = mp.spawn(entry, args=(logdir, world, port, paths, seconds, ret), nprocs=world, join=False)
    deadline = time.time() + max(600, 3 * seconds)
    err = None
    try:
        while not ctx.join(timeout=5):
            if time.time() > deadline:
                err = "timeout"; break
    except Exception as exc:
        err = f"{type(exc).__name__}: {str(exc)[-1500:]}"
    for pr in ctx.processes:
        if pr.is_alive():
            pr.kill()
    for pr in ctx.processes:
        pr.join(10)
    codes = [pr.exitcode for pr in ctx.processes]
    got = dict(ret)
    ok = err is None and sorted(got) == list(range(world)) and all(got[k][1] == 0 for k in got)
    if ok:
        import shutil
        shutil.rmtree(logdir, ignore_errors=True)
    else:
        with open(os.path.join(logdir, "failure.txt"), "w") as f:
            f.write(f"error: {err}\nexit codes: {codes}\nresults: {got}\n")
    return ok, err, codes, got


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["respawn", "resident"])
    ap.add_argument("count", type=float)
    ap.add_argument("sizes")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--load", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stress_agg"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    import ripp_amd as R
    sizes = [int(s) for s in args.sizes.split(",")]
    paths = []
    # (the parent touches the GPU only through child processes: instance generation runs in one)
    import multiprocessing as pymp
    for n in sizes:
        path = os.path.join(args.out, f"instance_{n}.npz"); paths.append(path)
        if not os.path.exists(path):
            p = pymp.get_context("spawn").Process(target=_make, args=(n, path)); p.start(); p.join()
            assert p.exitcode == 0
    load = []
    if args.load:
        import subprocess
        busy = "import sys; sys.path.insert(0, %r); import orclib as o; o.lib().orc_set_num_threads(1)\nwhile True:\n a, b = o.gen_g1(1, 64), o.gen_g2(2, 64); o.pairing_product_a(a, b)" % os.path.join(ROOT, "tests")
        load = [subprocess.Popen([sys.executable, "-c", busy]) for _ in range(args.load)]
    t0 = time.time(); fails = 0; total = 0
    try:
        if args.mode == "respawn":
            for run in range(int(args.count)):
                ok, err, codes, got = spawn(args.world, paths, 0, args.out, f"respawn_{run:03d}")
                total += 1
                if not ok:
                    fails += 1; print(f"run {run}: FAILED ({err}; exit codes {codes}; {got})", flush=True)
            print(f"respawn: {total} spawns of {args.world} ranks at n = {sizes} under load {args.load}: {fails} failures in {time.time() - t0:.0f} s", flush=True)
        else:
            ok, err, codes, got = spawn(args.world, paths, args.count, args.out, "resident")
            proofs = sum(v[0] for v in got.values()); bad = sum(v[1] for v in got.values())
            print(f"resident: {args.world} ranks x {proofs // max(1, len(got))} aggregate + GIPA proofs each at n = {sizes} under load {args.load} in {time.time() - t0:.0f} s: "
                  f"{'ok' if ok else 'FAILED'} (mismatches {bad}, error {err}, exit codes {codes})", flush=True)
            fails = 0 if ok else 1
    finally:
        for p in load:
            p.kill()
    sys.exit(1 if fails else 0)


def _make(n, path):
    import ripp_amd as R
    R.init(0)
    make_instance(n, path)


if __name__ == "__main__":
    main()
