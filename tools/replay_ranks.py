#!/usr/bin/env python3
"""One rank of G at full speed on ONE GPU: recorded-peer measurement of the sharded SIPP prover.

The build's boxes have one MI355X.  G processes time-slicing it measure control flow only; this rig measures a rank's real critical
path instead.  A proof is deterministic, so the blocks a rank receives in its all-gathers are too:

  record   G processes on cuda:0 (callback transport over gloo) prove the statement once; rank 0 keeps every exchange
           (ripp_comm_record / ripp_comm_recording_save).  Every rank's proof is compared with the CPU oracle's.
  replay   ONE process is rank k of G (ripp_comm_init_replay), alone on the device: bench.py's timed call (ripp_sipp_prove_sharded on
           host slices) with the peers' blocks served from the recording, each exchange completing no earlier than the slowest peer
           whose gaps are known would have arrived (+ --latency-us per exchange).  Its own gaps / blocks go back into the file.
  all      record, then the passes  rank 0 (peers instant) -> rank 1 (sees rank 0's digest time; its gaps stand for ranks 1..G-1, which do
           the same work) -> rank 0 (sees the measured peers) -> rank 1 again: a discrete-event model of the G-rank run with one live
           rank per pass.  Writes rank{0,1}_of_{G}_{bench.json,timeline.txt} (RIPP_TRACE of the last proof) into --out-dir.

  python tools/replay_ranks.py all --world 8 [--log-n 20] [--steps 5] [--warmup 2] [--latency-us 20] [--out-dir gpurun_out/replay]
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def oracle_proof(log_n, out_dir):
    """The CPU oracle's proof of the synthetic statement (cached per out_dir: ~80 s at n = 2^20 on 16 CPUs)."""
    import numpy as np
    path = os.path.join(out_dir, f"oracle_proof_n{log_n}.npz")
    if os.path.exists(path):
        z = np.load(path); return z["value"], z["proof"], z["ch"]
    import orclib as o
    n = 1 << log_n
    a, b, r = o.gen_g1(1000, n), o.gen_g2(2000, n), o.gen_scalars(0, n)
    value = o.product_of_pairings_with_coeffs(a, b, r)
    rc, proof, ch = o.sipp_prove(a, b, r, value)
    assert rc == 0
    np.savez(path, value=value, proof=proof, ch=ch)
    return value, proof, ch


def statement(R, rank, world, n):
    nl = n // world
    a, b, r = R.synth_g1(1000, nl, first=rank, stride=world), R.synth_g2(2000, nl, first=rank, stride=world), R.synth_fr(0, nl, first=rank, stride=world)
    full = (R.synth_g1(1000, n), R.synth_g2(2000, n), R.synth_fr(0, n)) if rank == 0 else None
    return a, b, r, full


# ---- workload "aggregate": config 5 (aggregate_proofs, groth16_aggregation.rs:77-160) across the ranks ---------------------------------------------
def agg_statement(R, o, n):
    alpha, beta = o.fr_array([0xa1fa0001])[0], o.fr_array([0xbe7a0001])[0]
    return R.SRS.from_trapdoors(alpha, beta, n), R.synth_g1(101, n), R.synth_g2(202, n), R.synth_g1(303, n)


def agg_expected(log_n, out_dir):
    """the ORACLE's aggregate of the synthetic instance (cached per out_dir): the members compared bit for bit"""
    import numpy as np
    path = os.path.join(out_dir, f"oracle_aggregate_n{log_n}.npz")
    if not os.path.exists(path):
        import orclib as o
        import ripp_amd as R
        R.init(0)
        srs, a, b, c = agg_statement(R, o, 1 << log_n)
        rc, exp = o.aggregate_proofs(srs.g_alpha_powers, srs.h_beta_powers, a, b, c)
        assert rc == 0
        d = {k: np.array(exp.field(k)) for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c")}
        d.update({k: np.array(getattr(exp, k)) for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript")})
        np.savez(path, **d)
    return dict(np.load(path))


def agg_check(got, exp, who):
    import numpy as np
    for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c"):
        assert np.array_equal(np.array(got.field(k)), exp[k]), f"{who}: aggregate member {k} differs from the oracle's"
    for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript"):
        assert np.array_equal(np.array(getattr(got, k)), exp[k]), f"{who}: aggregate member {k} differs from the oracle's"


def agg_record_worker(args):
    import torch.distributed as dist
    import orclib as o
    import ripp_amd as R
    from ripp_amd.sharded import NativeComm, shard
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    exp = agg_expected(args.log_n, args.out_dir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R.init(0)
    comm = NativeComm("callback")
    srs, a, b, c = agg_statement(R, o, 1 << args.log_n)
    if rank == 0:
        comm.record(True)
    got, _ = R.aggregate_proofs_sharded(srs, shard(a, rank, world), shard(b, rank, world), shard(c, rank, world))
    agg_check(got, exp, f"recording run, rank {rank}")
    if rank == 0:
        comm.save_recording(args.rec)
    dist.barrier(); comm.close(); dist.destroy_process_group()


def agg_replay_worker(args):
    pin_cores(args)
    import torch
    import orclib as o
    import ripp_amd as R
    from ripp_amd.sharded import ReplayComm, shard
    rank, world = args.rank, args.world
    exp = agg_expected(args.log_n, args.out_dir)
    torch.cuda.set_device(0); R.init(0)
    srs, a, b, c = agg_statement(R, o, 1 << args.log_n)
    t_one = []
    for _ in range(args.warmup + args.steps):                      # the same instance unsharded, for the comparison on this box
        t0 = time.perf_counter(); got1, _ = R.aggregate_proofs(srs, a, b, c); t_one.append((time.perf_counter() - t0) * 1e3)
    agg_check(got1, exp, "unsharded")
    comm = ReplayComm(rank, world, args.rec, args.latency_us)
    sa, sb, sc = shard(a, rank, world), shard(b, rank, world), shard(c, rank, world)
    times = []
    for it in range(args.warmup + args.steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got, st = R.aggregate_proofs_sharded(srs, sa, sb, sc)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        agg_check(got, exp, f"replayed rank {rank} of {world}")
        if it >= args.warmup:
            times.append(dt)
    info = comm.info(); info.update(comm.check()); comm.close()          # check(): raises when an own block MEANS something else than the recorded one
    print(json.dumps({"what": f"aggregate_proofs, n = 2^{args.log_n}: rank {rank} of {world} alone on one MI355X, peers replayed (instant, + {args.latency_us} us per exchange)",
                      "ms_per_call_sharded": sum(times) / len(times), "ms_all": [round(t, 2) for t in times],
                      "ms_per_call_unsharded_same_box": sum(t_one[args.warmup:]) / args.steps, "replay": info, "members_equal_oracle": True}), flush=True)


def record_worker(args):
    """One of the G recording processes (all on cuda:0, gloo carries the all-gather)."""
    import numpy as np
    import torch.distributed as dist
    import ripp_amd as R
    from ripp_amd.sharded import NativeComm, native_sipp_prove
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n = 1 << args.log_n
    value, eproof, ech = oracle_proof(args.log_n, args.out_dir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R.init(0)
    comm = NativeComm("callback")
    a, b, r, full = statement(R, rank, world, n)
    if rank == 0:
        comm.record(True)
    proof, ch, st = native_sipp_prove(a, b, r, value, full=full)
    assert np.array_equal(proof, eproof) and np.array_equal(ch, ech), f"recording run: rank {rank}'s proof differs from the oracle's"
    if rank == 0:
        comm.save_recording(args.rec)
    dist.barrier(); comm.close(); dist.destroy_process_group()


def pin_cores(args):
    """--cores N: this rank may use N host cores only (what it gets on a node where 8 ranks share the CPUs); the library's workers then compete among themselves"""
    if args.cores > 0:
        avail = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, set(avail[:args.cores]))
        if args.cores < 8:
            os.environ.pop("RIPP_HOT_WORKERS", None)          # polling workers need cores of their own: leave the choice to the library (it counts the CPUs it may use)


def replay_worker(args):
    """Rank k of G alone on the GPU: the timed call of bench.py with recorded peers; prints one JSON line."""
    pin_cores(args)
    import numpy as np
    import torch
    import ripp_amd as R
    from ripp_amd.sharded import ReplayComm, native_sipp_prove, read_recording
    rank, world = args.rank, args.world
    n = 1 << args.log_n
    value, eproof, ech = oracle_proof(args.log_n, args.out_dir)
    torch.cuda.set_device(0); R.init(0)
    comm = ReplayComm(rank, world, args.rec, args.latency_us)
    a, b, r, full = statement(R, rank, world, n)
    times, stats_all = [], []
    for it in range(args.warmup + args.steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        proof, ch, st = native_sipp_prove(a, b, r, value, full=full)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert np.array_equal(proof, eproof) and np.array_equal(ch, ech), f"replayed rank {rank} of {world}: proof differs from the oracle's"
        if it >= args.warmup:
            times.append(dt * 1e3); stats_all.append(st)
        print(f"[replay] rank {rank}/{world} proof {it}: {dt * 1e3:.1f} ms", file=sys.stderr, flush=True)
    comm.save_recording(args.rec)
    info = comm.info(); info.update(comm.check())                  # raises when an own block MEANS something else than the recorded one: the measurement would be void
    _, ex = read_recording(args.rec)
    comm.close()
    ms = sum(times) / len(times); st = stats_all[-1]
    hash_ms = sum(s["statement_hash_ms"] + s["statement_hash_wait_ms"] for s in stats_all) / len(stats_all)
    out = {"what": f"rank {rank} of {world}, alone on one MI355X, peers replayed from a recording (tools/replay_ranks.py)", "n": n, "rank": rank, "world": world,
           "steps": args.steps, "warmup": args.warmup, "latency_us_per_exchange": args.latency_us, "host_cores": len(os.sched_getaffinity(0)),
           "ms_per_step": ms, "ms_per_step_median": sorted(times)[len(times) // 2], "ms_per_step_all": [round(t, 3) for t in times],
           "value_if_this_rank_is_the_slowest": n / (ms * 1e-3), "unit": "pairs/s",
           "statement_hash_ms": round(hash_ms, 3) if rank == 0 else None, "post_hash_ms": ms - hash_ms if rank == 0 else None,
           "exchange_ms_per_step": sum(s["exchange_ms"] for s in stats_all) / len(stats_all),
           "look_ahead": {"items": int(st["look_items"]), "pairs": int(st["look_pairs"])},
           "phase_ms": {k: round(v, 3) for k, v in st.items() if k.endswith("_ms")},
           "replay": info, "proof_equals_oracle": True,
           "gaps_ms_last_proof": {"this_rank": [round(float(g[rank]), 3) for _, g, _ in ex],
                                  "peers_modelled": [[None if g[w] != g[w] else round(float(g[w]), 3) for w in range(world)] for _, g, _ in ex][:3]}}
    print(json.dumps(out), flush=True)


def run_record(args, world, rec):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RIPP_RANKS_PER_DEVICE=str(world),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "_record", "--log-n", str(args.log_n), "--out-dir", args.out_dir, "--rec", rec, "--workload", args.workload], env=env))
    deadline = time.time() + 900
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=max(1, deadline - time.time())))
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            raise SystemExit("recording run hangs")
    if any(rcs):
        raise SystemExit(f"recording run failed: exit codes {rcs}")


def run_replay(args, world, rank, rec, tag):
    cmd = [sys.executable, os.path.abspath(__file__), "_replay", "--log-n", str(args.log_n), "--out-dir", args.out_dir, "--rec", rec, "--world", str(world),
           "--rank", str(rank), "--steps", str(args.steps), "--warmup", str(args.warmup), "--latency-us", str(args.latency_us), "--workload", args.workload, "--cores", str(args.cores)]
    env = dict(os.environ, RIPP_TRACE="1", OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if p.returncode:
        sys.stderr.write(p.stderr[-4000:]); raise SystemExit(f"replay of rank {rank}/{world} failed ({p.returncode})")
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line); res["pass"] = tag
    base = os.path.join(args.out_dir, f"rank{rank}_of_{world}")
    json.dump(res, open(base + "_bench.json", "w"), indent=1)
    # the timeline of the LAST proof: everything the engine traced after the last "[replay] ... proof" marker but one
    lines = p.stderr.splitlines()
    marks = [i for i, ln in enumerate(lines) if ln.startswith("[replay]")]
    start = marks[-2] + 1 if len(marks) >= 2 else 0
    with open(base + "_timeline.txt", "w") as f:
        f.write(f"# rank {rank} of {world}, n = 2^{args.log_n}, pass '{tag}', RIPP_TRACE of the last of {args.warmup + args.steps} proofs; {res['ms_per_step']:.1f} ms per step\n")
        f.write("\n".join(lines[start:]) + "\n")
    print(f"rank {rank}/{world} [{tag}]: {res['ms_per_step']:.1f} ms/step, post-hash {res['post_hash_ms']}, look-ahead {res['look_ahead']}, waited in exchanges {res['replay']['waited_ms'] / (args.warmup + args.steps):.1f} ms/proof", flush=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["all", "_record", "_replay"])
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--latency-us", type=float, default=20.0, help="fixed cost added to every replayed exchange (a small RCCL all-gather over xGMI: ~15-30 us)")
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "gpurun_out", "replay"))
    ap.add_argument("--rec", default=None)
    ap.add_argument("--workload", choices=["sipp", "aggregate"], default="sipp", help="aggregate: config 5 (use --log-n 14); one pass of rank 0 with instant peers (the ranks are symmetric)")
    ap.add_argument("--passes", type=int, default=2, help="rank 0 / rank 1 rounds of the fixed-point iteration")
    ap.add_argument("--cores", type=int, default=0, help="host cores the live rank may use (0 = all): 2 = its share of a 16-core node with 8 ranks")
    ap.add_argument("--sweep-latency-us", default="", help="comma-separated per-exchange latencies: after the passes, rank 0 is replayed once more at each of them")
    args = ap.parse_args()
    os.makedirs(args.out_dir, exist_ok=True)
    if args.mode == "_record":
        return agg_record_worker(args) if args.workload == "aggregate" else record_worker(args)
    if args.mode == "_replay":
        return agg_replay_worker(args) if args.workload == "aggregate" else replay_worker(args)
    import numpy as np
    from ripp_amd.sharded import read_recording, write_recording
    if args.workload == "aggregate":
        world = args.world
        rec = os.path.join(args.out_dir, f"agg_w{world}_n{args.log_n}.rec")
        p = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tools')!r}); import replay_ranks as r; r.agg_expected({args.log_n}, {args.out_dir!r})"])
        assert p.returncode == 0
        run_record(args, world, rec)
        w, ex = read_recording(rec)
        write_recording(rec, world, [(nb, np.full(world, np.nan), bl) for nb, _, bl in ex])
        cmd = [sys.executable, os.path.abspath(__file__), "_replay", "--log-n", str(args.log_n), "--out-dir", args.out_dir, "--rec", rec, "--world", str(world), "--rank", "0",
               "--steps", str(args.steps), "--warmup", str(args.warmup), "--latency-us", str(args.latency_us), "--workload", "aggregate"]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        if p.returncode:
            sys.stderr.write(p.stderr[-4000:]); raise SystemExit("aggregate replay failed")
        res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]); res["exchanges_per_call"] = len(ex)
        json.dump(res, open(os.path.join(args.out_dir, f"aggregate_rank0_of_{world}.json"), "w"), indent=1)
        print(f"aggregate n = 2^{args.log_n}, rank 0 of {world}: {res['ms_per_call_sharded']:.1f} ms sharded ({len(ex)} exchanges) against {res['ms_per_call_unsharded_same_box']:.1f} ms on one GPU", flush=True)
        return
    oracle_proof(args.log_n, args.out_dir)
    world = args.world
    rec = os.path.join(args.out_dir, f"w{world}_n{args.log_n}.rec")
    t0 = time.time()
    run_record(args, world, rec)
    w, ex = read_recording(rec)
    assert w == world
    # the recording processes time-sliced ONE device: their gaps mean nothing
    write_recording(rec, world, [(nb, np.full(world, np.nan), bl) for nb, _, bl in ex])
    print(f"recorded {len(ex)} exchanges of a {world}-rank proof at n = 2^{args.log_n} in {time.time() - t0:.0f} s (every rank's proof equals the oracle's)", flush=True)
    summary = []
    for it in range(args.passes):
        summary.append(run_replay(args, world, 0, rec, "peers instant" if it == 0 else f"peers as measured in pass {it}"))
        res1 = run_replay(args, world, 1, rec, f"rank 0 as measured in pass {it + 1}")
        summary.append(res1)
        # ranks 2..G-1 do the same work as rank 1 on other residues of the same statement: its gaps stand for theirs
        w, ex = read_recording(rec)
        for _, gaps, _ in ex:
            gaps[2:] = gaps[1]
        write_recording(rec, world, ex)
    json.dump(summary, open(os.path.join(args.out_dir, f"w{world}_n{args.log_n}_passes.json"), "w"), indent=1)
    if args.sweep_latency_us:
        # the fixed per-exchange cost is the one free parameter of the model: rank 0 against the measured peers at each value (its own gaps are rewritten each time,
        # the peers' stay).  Then ONE more run with the rank confined to 2 host cores -- its share of a 16-core node that hosts 8 ranks.
        sweep = []
        base_lat, base_cores = args.latency_us, args.cores
        for lat in [float(x) for x in args.sweep_latency_us.split(",")]:
            args.latency_us = lat
            res = run_replay(args, world, 0, rec, f"latency sweep: {lat:g} us per exchange")
            sweep.append({k: res[k] for k in ("latency_us_per_exchange", "host_cores", "ms_per_step", "ms_per_step_all", "statement_hash_ms", "post_hash_ms", "exchange_ms_per_step", "replay")})
        args.latency_us = base_lat
        for cores in (8, 4, 2):
            args.cores = cores
            res = run_replay(args, world, 0, rec, f"{cores} host cores for the rank")
            sweep.append({k: res[k] for k in ("latency_us_per_exchange", "host_cores", "ms_per_step", "ms_per_step_all", "statement_hash_ms", "post_hash_ms", "exchange_ms_per_step", "replay")})
        args.cores = base_cores
        json.dump({"what": f"rank 0 of {world}, n = 2^{args.log_n}, peers as measured; per-exchange latency swept, then the rank confined to 8 / 4 / 2 host cores", "runs": sweep},
                  open(os.path.join(args.out_dir, f"w{world}_n{args.log_n}_latency_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
