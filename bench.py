#!/usr/bin/env python3
"""bench.py -- SIPP prover pairing-products/sec at n = 2^20 on BLS12-381 (BASELINE.json's metric).

One "step" = one complete SIPP::prove (sipp/src/lib.rs:42-106: statement hash, a_i <- r_i a_i, log2 n rounds of
two pairing products + Fiat-Shamir + two folds) over a synthetic statement of n random-looking G1 x G2 pairs,
called the way SURVEY.md section 8(d) defines the metric: `ripp_sipp_prove` on HOST slices (the statement in host
memory; the upload happens inside the call, the statement hash starts on the caller's buffers before it).
`value` = n / t_step (pairs per second, whole job).  `value_resident` is the same proof from a statement already
resident in HBM (ripp_sipp_job_prove), timed over the same number of steps right after.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 20]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
          (one rank per GPU, RCCL; STRONG scaling: n is fixed, shards are index residues mod N; the step is
           ripp_sipp_prove_sharded on every rank's host shard)

Extra objects on the JSON line: "roofline" (dominant kernel, algorithmic HBM bytes / measured launch time vs the
8 TB/s peak -- this path is integer-ALU bound, see DESIGN.md -- with the multiply-adds the kernel executes against the
hardware's issue rate under "int_alu") and "cpu_baseline" (the CPU oracle = a C/OpenMP restatement of the reference
algorithm, timed on this box's host cores AT THE HEADLINE SIZE; rank 0, N = 1 only).
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
# Integer roof of the dominant kernels: 32 x 32 + 64 -> 64 multiply-adds (v_mad_u64_u32), counted from the code, per pair:
#   k_line_products_q  68 lines x 3 lanes x 5 488 (12 lazily reduced six-product sums of 196 each + their 196-MAD reductions, fq_line_products.hpp)
#   k_miller_lines_q   63 doubling steps x 9 800 (6 Fp2 squares of 784, 3 Fp2 products of 1 176, 2 Fp2 x Fp of 784) + 5 addition steps x 16 072 (fq_miller.hpp)
#   k_line_products    68 x 3 x 8 064, k_miller_lines (63 x 25 + 5 x 41) Fp products x 288: the 12 x 32-bit forms (BLS12-377, RIPP_NO_FQ), every MAD followed by a carry capture
#   k_line_products_k  68 lines x 6 lanes x 2 156 (per Fp2 output 9 products + 2 reductions: Karatsuba inside the lazily reduced sums, fq_line_products_k.hpp; build round 6, BLS12-381)
MADS_PER_PAIR = {"k_line_products_q": 68 * 3 * 5488, "k_miller_lines_q": 63 * 9408 + 5 * 16072, "k_line_products_k": 68 * 6 * 2156,
                 "k_line_products": 68 * 3 * 8064, "k_miller_lines": (63 * 25 + 5 * 41) * 288}
MAD_ISSUE_PEAK_T = 34.72          # T lane-MAD/s, the hardware's measured v_mad_u64_u32 issue rate (profiles/r01_ubench_valu_rates.txt)
# what an isolated multiplier chain reaches, in the same unit: carry-free 14 x 28-bit product 78.5 G/s x 392 MADs (profiles/r03_fqbench.txt);
# 12 x 32-bit product 59.96 G/s x 288 MADs (profiles/r03_fpbench_production.txt)
MULTIPLIER_PEAK_T = {"q": 78.5e-3 * 392, "32": 59.96e-3 * 288}


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


def cpu_baseline(log_n_sample, statement=None):
    """Time the CPU oracle's SIPP prover on all host cores -- by default at the headline size itself (n = 2^20: ~85 s on the pool's 16-CPU
    quota), as the reference's harness times every size itself (sipp/examples/scaling-ipp.rs:57-82) -- and on ONE thread at n = 2^14."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orclib as o
    o.lib().orc_set_num_threads(o.effective_cpus())          # = the cgroup CPU quota of this box (16 on the GPU pool)
    n = 1 << log_n_sample
    if statement is not None and len(statement[0]) >= n:     # the engine's synthetic statement (the same generator: __graft_entry__.smoke() asserts it) and its value
        a, b, r = (np.ascontiguousarray(v[:n]) for v in statement[:3])
        value = statement[3] if len(statement[0]) == n else o.product_of_pairings_with_coeffs(a, b, r)
    else:
        a, b, r = o.gen_g1(1000, n), o.gen_g2(2000, n), o.gen_scalars(0, n)
        value = o.product_of_pairings_with_coeffs(a, b, r)
    t0 = time.perf_counter()
    rc, eproof, _ = o.sipp_prove(a, b, r, value)
    dt = time.perf_counter() - t0
    assert rc == 0
    out = {"_proof": eproof if statement is not None and len(statement[0]) == n else None,      # (the oracle's proof of the bench statement itself: compared with the GPU's below)
           "value": n / dt, "unit": "pairs/s", "cores": int(o.lib().orc_num_threads()), "kind": "port",
           "sample": f"oracle sipp_prove, n=2^{log_n_sample} (the headline statement itself when 20), same synthetic generator (seeds 1000/2000/0), one run of {dt:.2f} s"}
    # the per-core figure BASELINE.md section 2 asks for: the same prover on ONE thread, on a smaller sample of the same statement
    n1 = 1 << min(log_n_sample, 14)
    o.lib().orc_set_num_threads(1)
    v1 = o.product_of_pairings_with_coeffs(a[:n1], b[:n1], r[:n1])
    t0 = time.perf_counter()
    rc, _, _ = o.sipp_prove(a[:n1], b[:n1], r[:n1], v1)
    d1 = time.perf_counter() - t0
    o.lib().orc_set_num_threads(o.effective_cpus())
    assert rc == 0
    out["one_thread"] = {"value": n1 / d1, "unit": "pairs/s", "cores": 1, "sample": f"oracle sipp_prove, n=2^{n1.bit_length() - 1}, one run of {d1:.2f} s"}
    return out


def config_figures(R, np, with_oracle):
    """BASELINE.json's configs 2 and 3 beside the headline (outside its timed region): PairingInnerProduct::inner_product at n = 2^16 on Jacobian inputs
    (SURVEY.md section 8d's "raw P1 throughput") and the G1 / G2 MultiexponentiationInnerProduct at n = 2^20, each on HOST slices (upload inside), best of 3
    after one warm-up, with the algorithmic bytes of section 8(d) against the HBM peak and the CPU oracle's time for the same call on this box."""
    from ripp_amd import api
    o = None
    if with_oracle:                     # the CPU leg: the oracle supplies the randomised Jacobian representatives and the expected values
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orclib as o
    one = api._fp_one()

    def jac1(p): return np.ascontiguousarray(np.concatenate([p, np.tile(one, (len(p), 1))], axis=1))                                             # (x, y, 1)
    def jac2(p): return np.ascontiguousarray(np.concatenate([p, np.tile(np.concatenate([one, np.zeros(6, dtype=np.uint64)]), (len(p), 1))], axis=1))
    out = {}
    n2 = 1 << 16
    a2, b2 = R.synth_g1(1000, n2), R.synth_g2(2000, n2)
    aj, bj = (o.blind_g1(a2, 1), o.blind_g2(b2, 1)) if o else (jac1(a2), jac2(b2))

    def best(fn, reps=3):
        fn(); ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); v = fn(); ts.append(time.perf_counter() - t0)
        return min(ts), v
    t, got = best(lambda: R.PairingInnerProduct.inner_product(aj, bj))
    fig = {"ms": t * 1e3, "pairs_per_s": n2 / t, "algorithmic_GBps": n2 * (144 + 288) / t / 1e9, "frac": n2 * (144 + 288) / t / 1e9 / HBM_PEAK_GBS,
           "call": "ripp_pairing_product_j, n = 2^16 Jacobian inputs (144 + 288 B per pair), host slices"}
    if with_oracle:
        t0 = time.perf_counter(); rc, exp = o.pairing_product_j(aj, bj); fig["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
        fig["equals_oracle"] = bool(rc == 0 and np.array_equal(got, exp))
    out["pairing_product_2p16"] = fig
    n3 = 1 << 20
    a, b, r = R.synth_g1(1000, n3), R.synth_g2(2000, n3), R.synth_fr(0, n3)
    for name, bases, ip, norm, per_term, shape in (("msm_g1_2p20", jac1(a), R.MultiexponentiationInnerProductG1, R.normalize_batch_g1, 144 + 32, (1, 12)),
                                                   ("msm_g2_2p20", jac2(b), R.MultiexponentiationInnerProductG2, R.normalize_batch_g2, 288 + 32, (1, 24))):
        t, got = best(lambda: ip.inner_product(bases, r))
        fig = {"ms": t * 1e3, "terms_per_s": n3 / t, "algorithmic_GBps": n3 * per_term / t / 1e9, "frac": n3 * per_term / t / 1e9 / HBM_PEAK_GBS,
               "call": "ripp_%s_j, n = 2^20 Jacobian bases (%d B per term with its scalar), host slices" % (name[:6], per_term)}
        if with_oracle:
            t0 = time.perf_counter()
            exp = (o.g1_to_affine(o.msm_g1_a(a, r)) if name == "msm_g1_2p20" else o.g2_to_affine(o.msm_g2_a(b, r))).reshape(shape)
            fig["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
            fig["equals_oracle"] = bool(np.array_equal(norm(got), exp))
        out[name] = fig
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--cpu-log-n", type=int, default=20, help="log2 size of the CPU-baseline run (default: the headline size, ~85 s on 16 CPUs; 0 disables)")
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
    from ripp_amd.sharded import NativeComm, native_sipp_job_prove, native_sipp_prove

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
        import datetime
        # a rank that DIES must end the job, not hang it: the process group's collectives (and through them the callback transport) give up after this long;
        # the library's own RCCL exchanges poll against ripp_config.comm_timeout_ms (60 s).  A failing rank exits non-zero -- nothing here re-executes a process.
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("RIPP_BENCH_DIST_TIMEOUT_S", "120"))))     # rendezvous + barriers; "nccl" IS RCCL on ROCm
        # the proof's collectives run INSIDE libripp_hip.so: its own RCCL communicator over xGMI (id handed over through torch.distributed),
        # or -- single-device test mode -- an all-gather callback over gloo
        comm = NativeComm("rccl" if backend == "nccl" else "callback")
    else:
        dist = None
        comm = None

    # ---- synthetic statement: a_i = (1000+i) G1, b_i = (2000+i) G2, r_i from SplitMix64(0) ----
    # every rank generates its shard on its own GPU; rank 0 additionally holds the full statement on the host because
    # the prover hashes ALL of it (sipp/src/lib.rs:56-59).  `value` (the claimed product) is part of the statement.
    # The host copies are ordinary (pageable) numpy arrays -- what a caller of the trait surface hands over; the HIP runtime pins them for the upload
    # and caches the registration.  RIPP_BENCH_PINNED=1 places them in hipHostMalloc'ed memory instead (section 8d's wording): measured SLOWER on the
    # pool's boxes, 457.4 against 443.8 ms per proof on the same box (build round 4; the A/B record itself was not kept: re-run with RIPP_BENCH_PINNED=1 to repeat it) -- the CPU side of the proof, the
    # sequential hash and the serialisation workers, reads the statement more slowly from that mapping (hash 319.6 against 316.6 ms) and so does
    # everything behind it.
    def pinned(arr):
        if not os.environ.get("RIPP_BENCH_PINNED"):
            return arr
        t = torch.empty(arr.shape, dtype=torch.int64, pin_memory=True)
        v = t.numpy().view(np.uint64); v[...] = arr
        keep.append(t)
        return v
    keep = []
    if world == 1:
        a, b, r = (pinned(v) for v in (R.synth_g1(1000, n), R.synth_g2(2000, n), R.synth_fr(0, n)))
        full = (a, b, r)
        value = R.product_of_pairings_with_coeffs(a, b, r)
    else:
        nl = n // world
        a, b, r = (pinned(v) for v in (R.synth_g1(1000, nl, first=rank, stride=world), R.synth_g2(2000, nl, first=rank, stride=world), R.synth_fr(0, nl, first=rank, stride=world)))
        value = np.zeros(72, dtype=np.uint64); full = None
        if rank == 0:
            full = tuple(pinned(v) for v in (R.synth_g1(1000, n), R.synth_g2(2000, n), R.synth_fr(0, n)))
            value = R.product_of_pairings_with_coeffs(*full)
        t = torch.from_numpy(value.view(np.int64).copy()); t = t.cuda() if backend == "nccl" else t
        dist.broadcast(t, src=0); value = t.cpu().numpy().view(np.uint64)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, warmup, steps, ref_proof):
        """W untimed + exactly K timed steps, each bracketed by barrier + synchronize; (sum of the K times, MAX over ranks; per-step times; last stats; proof)"""
        for _ in range(warmup):
            proof, st = step()
        times, stats_all = [], []
        for _ in range(steps):
            fence(); t0 = time.perf_counter()
            proof, st = step()
            fence(); times.append(time.perf_counter() - t0)
            stats_all.append(st)
            if ref_proof is None:
                ref_proof = proof
            assert np.array_equal(proof, ref_proof), "non-deterministic proof"
        total = sum(times)
        if dist is not None:
            tt = torch.tensor([total] + times, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu"); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            total = float(tt[0].item()); times = [float(x) for x in tt[1:].tolist()]
        return total, times, stats_all, ref_proof

    # ---- the timed region: the section 8(d) call.  HOST slices in, proof bytes out; upload of the statement (shard) inside the call
    kill_rank = int(os.environ.get("RIPP_BENCH_KILL_RANK", "-1"))       # test hook (tests/test_sharded_gloo.py): this rank kills itself in round 3 of its second proof
    host_calls = [0]

    def host_step():
        host_calls[0] += 1
        if world > 1 and rank == kill_rank and host_calls[0] == 2:
            import ctypes
            from ripp_amd._lib import lib as _lib
            _lib().ripp_test_inject_failure.restype = None
            _lib().ripp_test_inject_failure(ctypes.c_int32(rank), ctypes.c_int32(1003))
        if world == 1:
            proof, ch, st = R.SIPP.prove_one_shot(a, b, r, value)       # ripp_sipp_prove: hashing starts on the caller's buffers, overlaps the upload and the first kernels
        else:
            proof, ch, st = native_sipp_prove(a, b, r, value, full=full if rank == 0 else None)      # ripp_sipp_prove_sharded
        return proof, st
    total, times, stats_all, ref_proof = timed(host_step, args.warmup, args.steps, None)
    ms_per_step = total / args.steps * 1e3
    ms_median = sorted(times)[len(times) // 2] * 1e3 if len(times) % 2 else 0.5 * (sorted(times)[len(times) // 2 - 1] + sorted(times)[len(times) // 2]) * 1e3
    stats = stats_all[-1]
    hash_ms_steps = [st["statement_hash_ms"] + st["statement_hash_wait_ms"] for st in stats_all]        # THIS call's hash, per step (rank 0)

    # ---- the same proof from a statement (shard) already resident in HBM: ripp_sipp_job_prove[_sharded]; same number of steps
    job = R.SippJob(a, b, r, rank=rank, world=world)

    def resident_step():
        if world == 1:
            proof, ch, st = job.prove(value)
        else:
            proof, ch, st = native_sipp_job_prove(job, value, full=full if rank == 0 else None)
        return proof, st
    total_res, times_res, stats_res, _ = timed(resident_step, 1, args.steps, ref_proof)
    resident_ms = total_res / args.steps * 1e3
    job.close(); job = None

    if rank == 0:
        # dominant kernel of the path on this rank, from HIP events recorded on the engine's own stream
        # the kernels the engine launches for throughput-sized products: the carry-free twins unless switched off
        no_fq = bool(os.environ.get("RIPP_NO_FQ"))
        lp_name = "k_line_products" if no_fq or int(os.environ.get("RIPP_LP_FQ_MIN", "0")) > (1 << 19) else "k_line_products_q" if os.environ.get("RIPP_NO_LP_KARA") else "k_line_products_k"
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
        mads = MADS_PER_PAIR[dom[3]]
        mad_rate_t = dom[2] * mads / (dom[0] * 1e-3) / 1e12 if dom[0] > 0 else 0.0            # T multiply-adds per second inside the kernel
        carry_free = dom[3].endswith(("_q", "_k"))
        mult_peak_t = MULTIPLIER_PEAK_T["q" if carry_free else "32"]
        hash_ms = sum(hash_ms_steps) / len(hash_ms_steps)
        out = {
            "metric": "SIPP prover pairing-products/sec at n=2^%d BLS12-381" % args.log_n,
            "value": n / (ms_per_step * 1e-3), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "sipp_prove (all log2 n rounds, Blake2s Fiat-Shamir)", "curve": "BLS12-381", "n": n,
                       "sharding": "index residue mod %d" % world, "inputs": "a_i=(1000+i)G1, b_i=(2000+i)G2, r_i=SplitMix64(0) 254-bit",
                       "call": ("ripp_sipp_prove" if world == 1 else "ripp_sipp_prove_sharded") + " on host slices (%s), upload inside the timed call" % ("pinned" if os.environ.get("RIPP_BENCH_PINNED") else "pageable")},
            # the step is bound by the HOST's sequential Blake2s of the 336 B x n statement, whose speed differs by a few per cent from box to box: read `value` next to it
            "statement_hash_ms": round(hash_ms, 3), "ms_per_step_median": ms_median, "value_median": n / (ms_median * 1e-3),
            "ms_per_step_all": [round(t * 1e3, 3) for t in times],
            # the hash of the SAME calls, step by step: a slow step is a slow hash (the core's clock / a neighbour on its L3), not a slow GPU -- read the two lists side by side
            "statement_hash_ms_all": [round(h, 3) for h in hash_ms_steps],
            "post_hash_ms_all": [round(t * 1e3 - h, 3) for t, h in zip(times, hash_ms_steps)],
            "value_resident": n / (resident_ms * 1e-3), "ms_per_step_resident": resident_ms,
            # `bound` / `achieved` / `peak` / `frac` are the byte roofline the bench contract asks for; `binding` names the roof that actually limits the kernel
            "roofline": {"bound": "hbm", "binding": "int_alu", "kernel": dom[3], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom[2] * ALG_BYTES_PER_PAIR / max(dom[1], 1),
                         "avg_launch_ms": dom[0] / max(dom[1], 1), "launches_per_step": dom[1], "pairs_per_step": dom[2],
                         # the roof that actually binds: multiply-adds on the VALU.  `achieved` = the MADs the kernel EXECUTES (counted from its code,
                         # MADS_PER_PAIR above) per second of its launches; `peak` = the hardware's measured MAD issue rate; `multiplier_peak` = what an
                         # isolated product chain of the same limb form reaches (it pays for its non-MAD instructions too).  The ceiling of a kernel that
                         # runs at 2 waves per SIMD: a lone wave issues one v_mad_u64_u32 per 3.99 ns, two waves one per 2.11 ns, eight one per 1.89 ns.
                         "int_alu": {"unit": "T MAD/s", "achieved": mad_rate_t, "peak": MAD_ISSUE_PEAK_T, "peak_kind": "measured v_mad_u64_u32 issue rate of the chip",
                                     "frac": mad_rate_t / MAD_ISSUE_PEAK_T, "mads_per_pair": mads,
                                     "multiplier_peak": mult_peak_t, "multiplier_kind": "isolated %s Montgomery product chain, MADs/s" % ("14 x 28-bit carry-free" if carry_free else "12 x 32-bit"),
                                     "frac_of_multiplier": mad_rate_t / mult_peak_t,
                                     # the same launches priced at build round 5's multiply-add count for this stage (k_line_products_q: 1 119 552 per pair): the
                                     # Karatsuba form executes 21 % fewer multiply-adds per pair, so its EXECUTED rate is lower at a higher pair rate -- compare this one across rounds
                                     "achieved_at_r05_count": (dom[2] * MADS_PER_PAIR["k_line_products_q"] / (dom[0] * 1e-3) / 1e12) if dom[3] == "k_line_products_k" and dom[0] > 0 else None,
                                     "ns_per_pair": dom[0] * 1e6 / dom[2] if dom[2] else None,
                                     "occupancy_ceiling": {"waves_per_simd": 2, "frac_of_issue_roof": round(1.89 / 2.11, 3)}},
                         "note": "integer-ALU bound (381-bit Montgomery arithmetic, ~1.8 M multiply-adds per 288 input bytes); see DESIGN.md"},
            "phase_ms": {k: round(v, 3) for k, v in stats.items() if k.endswith("_ms")},
        }
        # What the statement hash hides: the prover cannot draw its first challenge before rank 0 has hashed the whole
        # statement with a sequential Blake2s.  hash_wait_ms = time rank 0 was BLOCKED on that hash (everything the GPUs could do without a
        # challenge -- scaling, round 0, fold tables, the look-ahead of rounds 1..k -- was done by then); gpu_phase_ms = the rest of the step;
        # value_excl_hash = n / gpu_phase_ms -- the figure that scales with the number of GPUs.  `value` stays end to end.
        hash_wait = sum(st["hash_ms"] for st in stats_all) / len(stats_all)
        out["phase_ms"]["hash_wait_ms"] = round(hash_wait, 3)
        out["gpu_phase_ms"] = ms_per_step - hash_wait
        out["value_excl_hash"] = n / ((ms_per_step - hash_wait) * 1e-3)
        # The look-ahead FILLS the window (hash_wait_ms -> 0 by design), so the figure that shows what more GPUs buy is the time the proof needs
        # AFTER the digest exists: post_hash_ms = step - duration of the statement hash OF THE SAME CALLS (ripp_stats.statement_hash_ms, averaged
        # over the timed steps; DESIGN.md section 6 has the model).
        out["post_hash_ms"] = ms_per_step - hash_ms
        out["post_hash_ms_resident"] = resident_ms - sum(st["statement_hash_ms"] + st["statement_hash_wait_ms"] for st in stats_res) / len(stats_res)
        out["look_ahead"] = {"items": int(stats["look_items"]), "pairs": int(stats["look_pairs"]), "order": "(1,l) (1,r) (2,l) (2,r) (3,l) (3,r)"}
        out["miller_loops_per_s"] = 2 * (n - 1) / (ms_per_step * 1e-3)            # the 2 (n - 1) Miller loops of the reference's prover per second of the step
        if world == 1 and args.log_n == 20 and not os.environ.get("RIPP_BENCH_NO_CONFIGS"):
            out.update(config_figures(R, np, args.cpu_log_n > 0))
        parity_ok = True
        if world == 1 and args.cpu_log_n > 0:
            cb = cpu_baseline(args.cpu_log_n, (a, b, r, value))
            eproof = cb.pop("_proof")
            if eproof is not None:          # the CPU baseline ran the oracle's prover on the bench statement: a live full-size parity check in every default run
                parity_ok = bool(np.array_equal(eproof, ref_proof))
                cb["proof_equals_gpu"] = parity_ok
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
        if not parity_ok:
            sys.exit("bench.py: the GPU proof differs from the CPU oracle's proof of the same statement")
    if dist is not None:
        dist.barrier(); comm.close(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
