#!/bin/bash
# Long walks with 3 / 4 / 5 GOFs per job at 16 jobs in flight (the arena of a GOF went from 4.5 to under 3 GB in round 4, so 64 and 80 GOFs in flight fit the 288 GB),
# and 12 x 5 / 12 x 6. Output: gpurun_out/job_shape_sweep.txt
out=gpurun_out/job_shape_sweep.txt; : > $out
B="python bench.py --warmup 16 --cpu-sample 0 --quality 0 --walk-frames 0 --fanout-gofs 0 --multi-gof 0 --sweep 0 --steady-steps 0"
run() { # label, steps, gofs per job, in flight
  line=$(timeout -k 10 400 $B --steps $2 --gofs-per-job $3 --in-flight $4 2>gpurun_out/job_shape_sweep.err | tail -1)
  python - "$1" "$line" >> $out <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2]); c = d["config"]
    print(sys.argv[1], "fps", d["value"], "ms/gof", d["ms_per_step"], "arena MB/GOF", c["arena_MB_per_gof"], "in flight", c["gofs_in_flight"], "span", d["host_ms"]["job_gpu_span"], "kernel_ms", d["roofline"]["kernel_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e, sys.argv[2][:300])
PY
}
run "16x3" 384 3 16
run "16x4" 384 4 16
run "16x5" 400 5 16
run "12x5" 360 5 12
run "16x3" 384 3 16
run "16x4" 384 4 16
cat $out
