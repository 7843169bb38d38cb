# Copy the summaries collected by tools/run_profiles.sh <tag> (gpurun_out/prof_<tag>/) into profiles/ under the round's names.
TAG=${1:-r06}; R=${2:-r06}
S=gpurun_out/prof_$TAG
cp $S/kstats/k_kernel_stats.csv profiles/${R}_bench_kernel_stats_rocprofv3.csv
cp $S/bench_n1.json profiles/${R}_bench_n1.json
cp $S/fpbench.txt profiles/${R}_fpbench_production.txt
cp $S/fqbench.txt profiles/${R}_fqbench.txt
cp $S/invbench.txt profiles/${R}_invbench.txt
[ -s $S/lpbench.txt ] && cp $S/lpbench.txt profiles/${R}_lpbench.txt
[ -s $S/fqgroup.txt ] && cp $S/fqgroup.txt profiles/${R}_fqgroup.txt
cp $S/hbm_traffic.csv profiles/${R}_hbm_traffic_pmc.csv
cp $S/traffic.json profiles/traffic_current.json
cp $S/msm_sweep.txt profiles/${R}_msm_2p20.txt
cp $S/msm_hbm_traffic.csv profiles/${R}_msm_2p20_hbm_traffic_pmc.csv
cp $S/msm_kstats/k_kernel_stats.csv profiles/${R}_msm_2p20_kernel_stats_rocprofv3.csv
cp $S/sq_counters.csv profiles/${R}_sq_counters_pmc.csv
ls -la profiles | tail -30
# the sweeps of tools/run_final.sh <tag> (gpurun_out/final_<tag>/), when present
F=gpurun_out/final_$TAG
if [ -d $F ]; then
  cp $F/ipp-mi355x-hip.csv profiles/${R}_scaling_ipp_mi355x.csv
  cp $F/ipp-cpu-oracle.csv profiles/${R}_scaling_ipp_cpu_oracle_16thr.csv
  cp $F/c377/ipp-mi355x-hip-bls12_377.csv profiles/${R}_scaling_ipp_bls12_377_mi355x.csv
  cp $F/c377/ipp-cpu-oracle-bls12_377.csv profiles/${R}_scaling_ipp_bls12_377_cpu_oracle_16thr.csv
  cp $F/aggregate_2p14.json profiles/${R}_aggregate_2p14.json
  for G in 2 4 8; do [ -s $F/bench_n${G}_single_device_gloo.json ] && cp $F/bench_n${G}_single_device_gloo.json profiles/${R}_bench_n${G}_single_device_gloo.json; done
  [ -s $F/post_hash_timeline.txt ] && cp $F/post_hash_timeline.txt profiles/${R}_post_hash_timeline.txt
  for f in stress_world4 stress_world8 stress_tail; do [ -s $F/$f.txt ] && tail -3 $F/$f.txt > profiles/${R}_$f.txt; done
  [ -s $F/poly_commit_bench.csv ] && cp $F/poly_commit_bench.csv profiles/${R}_poly_commit_bench.csv
  [ -s $F/bench_n1_derated.json ] && cp $F/bench_n1_derated.json profiles/${R}_bench_n1_derated.json
  [ -s $F/plan_derate_ab.txt ] && cp $F/plan_derate_ab.txt profiles/${R}_plan_derate_ab.txt
  [ -s $F/replay/w8_n20_latency_sweep.json ] && cp $F/replay/w8_n20_latency_sweep.json profiles/${R}_replay_latency_sweep.json
  [ -s $F/replay/aggregate_rank0_of_8.json ] && cp $F/replay/aggregate_rank0_of_8.json profiles/${R}_aggregate_2p14_sharded_replay.json
  for G in 2 4 8; do for K in 0 1; do for X in bench.json timeline.txt; do [ -s $F/replay/rank${K}_of_${G}_$X ] && cp $F/replay/rank${K}_of_${G}_$X profiles/${R}_rank${K}_of_${G}_$X; done; done; [ -s $F/replay/w${G}_n20_passes.json ] && cp $F/replay/w${G}_n20_passes.json profiles/${R}_replay_w${G}_passes.json; done
fi
