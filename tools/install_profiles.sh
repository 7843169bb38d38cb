# Copy the summaries collected by tools/run_profiles.sh <tag> (gpurun_out/prof_<tag>/) into profiles/ under the round's names.
TAG=${1:-r02c}; R=${2:-r02}
S=gpurun_out/prof_$TAG
cp $S/kstats/k_kernel_stats.csv profiles/${R}_bench_kernel_stats_rocprofv3.csv
cp $S/bench_n1.json profiles/${R}_bench_n1.json
cp $S/fpbench.txt profiles/${R}_fpbench_production.txt
cp $S/fqbench.txt profiles/${R}_fqbench.txt
cp $S/invbench.txt profiles/${R}_invbench.txt
cp $S/hbm_traffic.csv profiles/${R}_hbm_traffic_pmc.csv
cp $S/traffic.json profiles/traffic_current.json
cp $S/msm_sweep.txt profiles/${R}_msm_2p20.txt
cp $S/msm_hbm_traffic.csv profiles/${R}_msm_2p20_hbm_traffic_pmc.csv
cp $S/msm_kstats/k_kernel_stats.csv profiles/${R}_msm_2p20_kernel_stats_rocprofv3.csv
cp $S/sq_counters.csv profiles/${R}_sq_counters_pmc.csv
ls -la profiles | tail -30
