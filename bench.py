#!/usr/bin/env python3
"""bench.py -- SIPP prover pairing-products/sec at n = 2^20 on BLS12-381 (BASELINE.json's metric).

One "step" = one complete SIPP::prove (sipp/src/lib.rs:42-106: statement hash, a_i <- r_i a_i, log2 n rounds of
two pairing products + Fiat-Shamir + two folds) over a synthetic statement of n random-looking G1 x G2 pairs that
is already resident in HBM when the timed region starts.  `value` = n / t_step  (pairs per second, whole job).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 20]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
          (one rank per GPU, RCCL; STRONG scaling: n is fixed, shards are index residues mod N)

Extra objects on the JSON line: "roofline" (dominant kernel, algorithmic HBM bytes / measured launch time vs the
8 TB/s peak -- this path is integer-ALU bound) and "cpu_baseline" (the CPU oracle = a C/OpenMP
restatement of the reference algorithm, timed on this box's host cores on a bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_PAIR = 288          # one affine G1 (96 B) + one affine G2 (192 B) read once
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Integer-ALU roofs for the 381-bit Montgomery product of the dominant kernel (12 x 32-bit limbs: 288 v_mad_u64_u32 per product):
#   "mad_issue": the hardware's measured v_mad_u64_u32 issue rate, 34.7 T lane-MAD/s (profiles/r01_ubench_valu_rates.txt) / 288
#   "multiplier": the multiplier's own measured chip rate (every MAD is followed by the v_addc_co_u32 that captures its carry),
#                 profiles/r02_fpbench_production.txt.  (The carry-free 14 x 28-bit form of the fold kernels and the field VM, fq28.hpp,
#                 reaches 78.5 G products/s: profiles/r02_fqbench.txt.)
MAD_ISSUE_PEAK_G = 34.72e3 / 288
FP_MUL_PEAK_G = 59.96


def csrc_sha256():
    """Fingerprint of the kernel sources: a committed PMC profile is only quoted on the bench line when it was taken on THIS code."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ripp_amd", "csrc")
    for dirpath, _, files in sorted(os.walk(d)):
        for f in sorted(files):
            if f.endswith((".hip", ".hpp", ".inc")):
                h.update(f.encode()); h.update(open(os.path.join(dirpath, f), "rb").read())
    return h.hexdigest()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N FRESH rank processes through torch.distributed.run (this process has not
    touched the GPU -- nothing is re-exec'd after HIP initialisation) and relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--log-n", str(args.log_n), "--cpu-log-n", str(args.cpu_log_n)]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line, flush=True)
    sys.exit(p.returncode if p.returncode or line else 1)


def cpu_baseline(log_n_sample):
    """Time the CPU oracle's SIPP prover on all host cores for a bounded sample (same generator, smaller n)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib as o
    o.lib().orc_set_num_threads(o.effective_cpus())          # = the cgroup CPU quota of this box (16 on the GPU pool)
    n = 1 << log_n_sample
    a, b, r = o.gen_g1(1000, n), o.gen_g2(2000, n), o.gen_scalars(0, n)
    value = o.product_of_pairings_with_coeffs(a, b, r)
    t0 = time.perf_counter()
    rc, _, _ = o.sipp_prove(a, b, r, value)
    dt = time.perf_counter() - t0
    assert rc == 0
    out = {"value": n / dt, "unit": "pairs/s", "cores": int(o.lib().orc_num_threads()), "kind": "port",
           "sample": f"oracle sipp_prove, n=2^{log_n_sample}, same synthetic generator (seeds 1000/2000/0), one run of {dt:.2f} s"}
    # the per-core figure BASELINE.md section 2 asks for: the same prover on ONE thread, on a smaller sample of the same statement
    n1 = 1 << min(log_n_sample, 13)
    o.lib().orc_set_num_threads(1)
    v1 = o.product_of_pairings_with_coeffs(a[:n1], b[:n1], r[:n1])
    t0 = time.perf_counter()
    rc, _, _ = o.sipp_prove(a[:n1], b[:n1], r[:n1], v1)
    d1 = time.perf_counter() - t0
    o.lib().orc_set_num_threads(o.effective_cpus())
    assert rc == 0
    out["one_thread"] = {"value": n1 / d1, "unit": "pairs/s", "cores": 1, "sample": f"oracle sipp_prove, n=2^{n1.bit_length() - 1}, one run of {d1:.2f} s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--cpu-log-n", type=int, default=18, help="log2 size of the CPU-baseline sample (0 disables)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            self_launch(args)                                 # does not return
        args.gpus = world
    n = 1 << args.log_n

    import numpy as np
    import torch
    import ripp_amd as R
    from ripp_amd.sharded import NativeComm, native_sipp_job_prove

    # test hooks (used on 1-GPU boxes to exercise the N > 1 control flow): all ranks on device 0, gloo transport
    single_dev = bool(os.environ.get("RIPP_BENCH_SINGLE_DEVICE"))
    if single_dev:
        local_rank = 0
        os.environ.setdefault("RIPP_RANKS_PER_DEVICE", str(world))       # the look-ahead plan prices the hash window per DEVICE: all ranks' work lands on this one
    backend = os.environ.get("RIPP_BENCH_BACKEND", "gloo" if single_dev else "nccl")    # RCCL needs one device per rank
    if world > 1 and not single_dev and torch.cuda.device_count() < world:
        sys.exit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} device(s) visible (RIPP_BENCH_SINGLE_DEVICE=1 runs all ranks on device 0 over gloo, for control-flow tests only)")
    torch.cuda.set_device(local_rank)
    R.init(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)     # rendezvous + barriers; "nccl" IS RCCL on ROCm
        # the proof's collectives run INSIDE libripp_hip.so: its own RCCL communicator over xGMI (id handed over through torch.distributed),
        # or -- single-device test mode -- an all-gather callback over gloo
        comm = NativeComm("rccl" if backend == "nccl" else "callback")
    else:
        dist = None
        comm = None

    # ---- synthetic statement: a_i = (1000+i) G1, b_i = (2000+i) G2, r_i from SplitMix64(0) ----
    # every rank generates its shard on its own GPU; rank 0 additionally holds the full statement on the host because
    # the prover hashes ALL of it (sipp/src/lib.rs:56-59).  `value` (the claimed product) is part of the statement.
    if world == 1:
        a, b, r = R.synth_g1(1000, n), R.synth_g2(2000, n), R.synth_fr(0, n)
        full = (a, b, r)
        value = R.product_of_pairings_with_coeffs(a, b, r)
    else:
        nl = n // world
        a, b, r = R.synth_g1(1000, nl, first=rank, stride=world), R.synth_g2(2000, nl, first=rank, stride=world), R.synth_fr(0, nl, first=rank, stride=world)
        value = np.zeros(72, dtype=np.uint64); full = None
        if rank == 0:
            full = (R.synth_g1(1000, n), R.synth_g2(2000, n), R.synth_fr(0, n))
            value = R.product_of_pairings_with_coeffs(*full)
        t = torch.from_numpy(value.view(np.int64).copy()); t = t.cuda() if backend == "nccl" else t
        dist.broadcast(t, src=0); value = t.cpu().numpy().view(np.uint64)

    job = R.SippJob(a, b, r, rank=rank, world=world)      # statement (shard) now resident in HBM

    def one_step():
        if world == 1:
            proof, ch, st = job.prove(value)              # hashing overlaps the first kernels inside the engine
            return proof, st
        # ripp_sipp_job_prove_sharded: rank 0 hashes the full statement on a host thread of the library while all ranks run round 0
        proof, ch, st = native_sipp_job_prove(job, value, full=full if rank == 0 else None)
        return proof, st

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        proof, st = one_step()
    times, stats, ref_proof = [], None, None
    for _ in range(args.steps):
        fence(); t0 = time.perf_counter()
        proof, st = one_step()
        fence(); times.append(time.perf_counter() - t0)
        stats = st
        if ref_proof is None:
            ref_proof = proof
        assert np.array_equal(proof, ref_proof), "non-deterministic proof"
    total = sum(times)
    if dist is not None:
        tt = torch.tensor([total], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); total = float(tt.item())
    ms_per_step = total / args.steps * 1e3

    # the call SURVEY.md section 8(d) defines the metric on: ripp_sipp_prove from HOST slices (upload of the 336 MB statement inside the call,
    # the statement hash started on the caller's buffers before it).  Reported beside `value` (resident statement), never as `value`.
    host_slices_ms = None
    if world == 1:
        job.close(); job = None
        ts = []
        for _ in range(2):
            t0 = time.perf_counter(); p2 = R.SIPP.prove(a, b, r, value); ts.append(time.perf_counter() - t0)
            assert np.array_equal(p2, ref_proof), "one-shot proof differs from the resident-statement proof"
        host_slices_ms = min(ts) * 1e3

    if rank == 0:
        # dominant kernel of the path on this rank, from HIP events recorded on the engine's own stream
        # the kernels the engine launches for throughput-sized products: the carry-free twins unless switched off
        no_fq = bool(os.environ.get("RIPP_NO_FQ"))
        lp_name = "k_line_products" if no_fq or int(os.environ.get("RIPP_LP_FQ_MIN", "0")) > (1 << 19) else "k_line_products_q"
        ml_name = "k_miller_lines" if no_fq or int(os.environ.get("RIPP_ML_FQ_MIN", "0")) > (1 << 19) else "k_miller_lines_q"
        k_lines = (stats["kernel_miller_lines_ms_sum"], stats["kernel_miller_lines_launches"], stats["pairs_lines"], ml_name)
        k_prod = (stats["kernel_line_products_ms_sum"], stats["kernel_line_products_launches"], stats["pairs_products"], lp_name)
        dom = max(k_lines, k_prod, key=lambda k: k[0])
        achieved = (dom[2] * ALG_BYTES_PER_PAIR) / (dom[0] * 1e-3) / 1e9 if dom[0] > 0 else 0.0
        # HBM bytes per launch of that kernel from rocprofv3 PMC passes (FETCH_SIZE x2 correction + WRITE_SIZE, separate passes, this
        # workload; tools/pmc_traffic.py).  Quoted ONLY when the committed profile was taken on exactly these kernel sources
        # (csrc_sha256 recorded in the profile); null otherwise -- it is a counter measurement, not something a timed run can see.
        traffic = None
        try:
            if args.log_n == 20 and world == 1:
                prof = json.load(open(os.path.join(ROOT, "profiles", "traffic_current.json")))
                if prof.get("csrc_sha256") == csrc_sha256():
                    tr = prof["kernels"].get(dom[3])
                    traffic = tr["bytes_per_launch"] if tr else None
        except Exception:
            traffic = None
        out = {
            "metric": "SIPP prover pairing-products/sec at n=2^%d BLS12-381" % args.log_n,
            "value": n / (ms_per_step * 1e-3), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "sipp_prove (all log2 n rounds, Blake2s Fiat-Shamir)", "curve": "BLS12-381", "n": n,
                       "sharding": "index residue mod %d" % world, "inputs": "a_i=(1000+i)G1, b_i=(2000+i)G2, r_i=SplitMix64(0) 254-bit"},
            "roofline": {"bound": "hbm", "kernel": dom[3], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom[2] * ALG_BYTES_PER_PAIR / max(dom[1], 1),
                         "avg_launch_ms": dom[0] / max(dom[1], 1), "launches_per_step": dom[1], "pairs_per_step": dom[2],
                         # the roof that actually binds: 381-bit Montgomery products on the VALU.  Algorithmic Fp products per pair of the
                         # kernel (k_line_products: 68 sparse mul_by_014 of 13 Fp2 = 39 Fp products; k_miller_lines: 63 doubling steps of 25
                         # + 5 addition steps of 41) against the multiplier's measured chip rate (profiles/r02_fpbench_production.txt).
                         "int_alu": (lambda fpm: {"unit": "G Fp-mul/s", "achieved": dom[2] * fpm / (dom[0] * 1e-3) / 1e9 if dom[0] > 0 else 0.0,
                                                  "peak": MAD_ISSUE_PEAK_G, "peak_kind": "hardware v_mad_u64_u32 issue rate / 288 MADs per product",
                                                  "frac": (dom[2] * fpm / (dom[0] * 1e-3) / 1e9 / MAD_ISSUE_PEAK_G) if dom[0] > 0 else 0.0,
                                                  "multiplier_peak": FP_MUL_PEAK_G,
                                                  "frac_of_multiplier": (dom[2] * fpm / (dom[0] * 1e-3) / 1e9 / FP_MUL_PEAK_G) if dom[0] > 0 else 0.0,
                                                  # (multiplier_peak is the standalone 12 x 32-bit multiplier of profiles/r02_fpbench_production.txt: the carry-free kernels need fewer
                                                  #  instructions per product and may exceed it.)  The ceiling of a kernel that runs at 2 waves per SIMD: a lone wave issues one
                                                  #  v_mad_u64_u32 per 3.99 ns, two waves one per 2.11 ns, eight one per 1.89 ns (profiles/r01_ubench_valu_rates.txt).
                                                  "occupancy_ceiling": {"waves_per_simd": 2, "frac_of_issue_roof": round(1.89 / 2.11, 3)},
                                                  "fp_products_per_pair": fpm})(68 * 39 if dom[3].startswith("k_line_products") else 63 * 25 + 5 * 41),
                         "note": "integer-ALU bound (381-bit Montgomery arithmetic, ~5e3 Fp products per 288 input bytes); see DESIGN.md"},
            "phase_ms": {k: round(v, 3) for k, v in stats.items() if k.endswith("_ms")},
        }
        # What the statement hash hides: the prover cannot draw its first challenge before rank 0 has hashed the whole
        # statement with a sequential Blake2s.  hash_wait_ms = time rank 0 was BLOCKED on that hash (everything the GPUs could do without a
        # challenge -- scaling, round 0, fold tables, the look-ahead of rounds 1..k -- was done by then); gpu_phase_ms = the rest of the step;
        # value_excl_hash = n / gpu_phase_ms -- the figure that scales with the number of GPUs.  `value` stays end to end.
        hd = R.statement_hash_times()
        out["phase_ms"]["hash_wait_ms"] = round(stats["hash_ms"], 3)
        out["phase_ms"]["statement_hash_ms"] = round(hd[0] + hd[1], 3)
        out["gpu_phase_ms"] = ms_per_step - stats["hash_ms"]
        out["value_excl_hash"] = n / ((ms_per_step - stats["hash_ms"]) * 1e-3)
        # The look-ahead FILLS the window (hash_wait_ms -> 0 by design), so the figure that shows what more GPUs buy is the time the proof needs
        # AFTER the digest exists: post_hash_ms = step - duration of the statement hash (last step's hash; DESIGN.md section 6 has the model).
        out["post_hash_ms"] = ms_per_step - (hd[0] + hd[1])
        out["look_ahead"] = {"items": int(stats["look_items"]), "pairs": int(stats["look_pairs"]), "order": "(1,l) (1,r) (2,l) (2,r) (3,l) (3,r)"}
        if host_slices_ms is not None:
            out["host_slices_ms"] = host_slices_ms
            out["value_host_slices"] = n / (host_slices_ms * 1e-3)
        if world == 1 and args.cpu_log_n > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_log_n)
        print(json.dumps(out), flush=True)
    if job is not None:
        job.close()
    if dist is not None:
        dist.barrier(); comm.close(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
