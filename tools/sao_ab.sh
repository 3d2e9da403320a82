#!/bin/bash
# SAO by CTB (round 4) against the per-sample kernel: the same bench command, interleaved. Output: gpurun_out/sao_ab.txt
out=gpurun_out/sao_ab.txt; : > $out
B="python bench.py --steps 20 --warmup 5 --cpu-sample 0 --quality 0 --walk-frames 0 --fanout-gofs 0 --multi-gof 0 --sweep 0"
run() { label=$1; shift
  line=$(env "$@" timeout -k 10 500 $B 2>/dev/null | tail -1)
  python - "$label" "$line" >> $out <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2]); print(sys.argv[1], "driver", d["value"], "steady", d["steady_state_fps_256"], "kernel_ms", d["roofline"]["kernel_ms"], "span", d["host_ms"]["job_gpu_span"])
except Exception as e: print(sys.argv[1], "FAILED", e, sys.argv[2][:200])
PY
}
for rep in 1 2; do run "sao-by-ctb" RBT_X=0; run "sao-per-sample" RBT_SAO_PER_SAMPLE=1; done
cat $out
