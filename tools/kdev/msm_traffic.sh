cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/msmt; rm -rf $O; mkdir -p $O
python3 tools/msm_sweep.py 20 > $O/sweep.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f -o f --output-format csv -- python3 tools/msm_sweep.py 20 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w -o w --output-format csv -- python3 tools/msm_sweep.py 20 > /dev/null 2>&1
python3 tools/pmc_traffic.py $O/f $O/w $O/t.csv $O/t.json | head -4; cat $O/sweep.txt | tail -1
rm -rf $O
