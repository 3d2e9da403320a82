#!/bin/bash
# Development probe (GPU box): throughput with the loop filters fused (decoder k_loopfilter: RBT_FUSED_LF=1; encoder en_sao_ctb: default) and with the separate in-place
# deblocking launches (decoder: default; encoder: RBT_FUSED_ENC_LF=0). No profiler; every variant three times, interleaved. $1 = steps (20: the driver's command), $2 = warmup
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
QUIET="--cpu-sample 0 --multi-gof 0 --quality 0 --sweep 0 --walk-frames 0 --fanout-gofs 0 --steady-steps 0"
for rep in 1 2 3; do for v in "1 1" "0 1" "0 0"; do
  set -- $v; export RBT_FUSED_LF=$1 RBT_FUSED_ENC_LF=$2
  timeout -k 10 200 python3 $R/bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} $QUIET > $O/lfp.json 2> $O/lfp.err || exit 2
  python3 - "$1" "$2" $O/lfp.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("decoder fused", sys.argv[1], "encoder fused", sys.argv[2], "fps", d["value"], {k: round(v, 1) for k, v in d["roofline"]["kernel_ms"].items()}, flush=True)
PY
done; done
