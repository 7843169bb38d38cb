# ms_per_step + phases of bench.py under the given environment: bash tools/kdev/bench_ms.sh [ENV=VAL ...]
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
timeout 300 python3 bench.py --steps 4 --warmup 1 --cpu-log-n 0 2>/tmp/bench_ms.err | tail -1 > /tmp/bench_ms.json
python3 - <<PY
import json
try:
    d = json.load(open("/tmp/bench_ms.json")); print("$*", "%.1f ms" % d["ms_per_step"], {k: round(v, 1) for k, v in d["phase_ms"].items() if v})
except Exception as ex:
    print("$*", "FAILED", ex); print(open("/tmp/bench_ms.err").read()[-600:])
PY
