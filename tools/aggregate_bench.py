"""aggregate_proofs on n = 2^14 synthetic (A, B, C) triples, one MI355X next to the CPU oracle
(16-core quota) on the same inputs; GT / Fr members compared bit for bit, group members after normalisation."""
import sys, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, orclib as o, ripp_amd as R
R.init(0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 14
n = 1 << logn
alpha, beta = o.fr_array([0xa1fa0001])[0], o.fr_array([0xbe7a0001])[0]
t = time.perf_counter(); srs = R.SRS.from_trapdoors(alpha, beta, n); t_srs = time.perf_counter() - t
a, b, c = R.synth_g1(101, n), R.synth_g2(202, n), R.synth_g1(303, n)
ts = []
for _ in range(3):
    t = time.perf_counter(); got, stats = R.aggregate_proofs(srs, a, b, c); ts.append(time.perf_counter() - t)
vs = srs.get_verifier_key()
t = time.perf_counter()
ok_ab = R.TIPA_TIPP.verify_with_srs_shift(vs, [got.field("com_a"), got.field("com_b"), got.field("ip_ab")],
                                          dict(steps=got.ab_com_steps, base_a=got.field("ab_base_a"), base_b=got.field("ab_base_b"), final_ck_a=got.field("ab_final_ck_a"),
                                               final_ck_b=got.field("ab_final_ck_b"), opening_a=got.field("ab_opening_a"), opening_b=got.field("ab_opening_b")), got.field("r"))
ok_c = R.TIPAWithSSM.verify_with_structured_scalar_message(vs, (got.field("com_c"), got.field("agg_c")), got.field("r"),
                                                           dict(com_gt=got.c_com_gt, com_g1=got.c_com_g1, base_a=got.field("c_base_a"), final_ck_a=got.field("c_final_ck_a"), opening_a=got.field("c_opening_a")))
t_verify = time.perf_counter() - t
out = {"n": n, "gpu_aggregate_s": min(ts), "gpu_runs_s": ts, "gpu_proofs_per_s": n / min(ts), "gpu_srs_setup_s": t_srs, "gpu_tipa_verifiers_s": t_verify,
       "gpu_verifiers_accept": bool(ok_ab and ok_c), "gpu_stats_ms": {k: round(v, 1) for k, v in stats.items() if k.endswith("_ms") and v}}
if "--no-cpu" not in sys.argv:
    t = time.perf_counter(); rc, exp = o.aggregate_proofs(srs.g_alpha_powers, srs.h_beta_powers, a, b, c); tc = time.perf_counter() - t
    same = all(np.array_equal(got.field(k), exp.field(k)) for k in ("com_a", "com_b", "com_c", "ip_ab", "r", "ab_kzg_c", "c_base_b", "c_kzg_c")) and \
        all(np.array_equal(getattr(got, k), getattr(exp, k)) for k in ("ab_com_steps", "ab_transcript", "c_com_gt", "c_transcript"))
    same = same and np.array_equal(R.normalize_batch_g2(got.field("ab_opening_a").reshape(1, 36)), o.normalize_g2(np.ascontiguousarray(exp.field("ab_opening_a").reshape(1, 36))))
    same = same and np.array_equal(R.normalize_batch_g2(got.field("c_opening_a").reshape(1, 36)), o.normalize_g2(np.ascontiguousarray(exp.field("c_opening_a").reshape(1, 36))))
    out.update({"cpu_oracle_aggregate_s": tc, "cpu_threads": o.effective_cpus(), "cpu_proofs_per_s": n / tc, "speedup": tc / min(ts), "bit_exact_vs_oracle": bool(same and rc == 0)})
print(json.dumps(out, indent=1))
