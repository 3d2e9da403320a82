#!/bin/bash
# Development probe (GPU box): the driver's command and the long walk, $1 repeats each, quiet extras. Output lines: shape, fps, latency.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
QUIET="--cpu-sample 0 --multi-gof 0 --quality 0 --sweep 0 --walk-frames 0 --fanout-gofs 0 --steady-steps 0"
for rep in $(seq 1 ${1:-2}); do for shape in "--steps 20 --warmup 5" "--steps 256 --warmup 48"; do
  timeout -k 10 300 python3 $R/bench.py $shape $QUIET > $O/qab.json 2> $O/qab.err || { tail -5 $O/qab.err; exit 2; }
  python3 - "$shape" $O/qab.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "fps", d["value"], "ms/step", d["ms_per_step"], flush=True)
PY
done; done
