cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/kdev/lat16.py 16 20
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/lat16 -o lat --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/kdev/lat16.py 16 20 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/lat16
