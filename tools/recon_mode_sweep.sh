#!/bin/bash
# Reconstruction per level: per-diagonal launches / neighbour flags (the round-3 rule by depth) against the ready-queue launch at several widths.
# Same bench command each time (driver's command + the 256-step steady state), interleaved repeats. Output: gpurun_out/recon_mode_sweep.txt
out=gpurun_out/recon_mode_sweep.txt; : > $out
B="python bench.py --steps 20 --warmup 5 --cpu-sample 0 --quality 0 --walk-frames 0 --fanout-gofs 0 --multi-gof 0 --sweep 0"
run() { # label, env...
  label=$1; shift
  line=$(env "$@" timeout -k 10 500 $B 2>/dev/null | tail -1)
  python - "$label" "$line" >> $out <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    print(sys.argv[1], "driver", d["value"], "steady", d["steady_state_fps_256"], "recon_ms", d["roofline"]["kernel_ms"]["reconstruct+loopfilter"], "parse_ms", d["roofline"]["kernel_ms"]["cabac_parse"], "span", d["host_ms"]["job_gpu_span"])
except Exception as e:
    print(sys.argv[1], "FAILED", e, sys.argv[2][:200])
PY
}
for rep in 1 2; do
  run "round3-rule" RBT_RECON_QUEUE=0
  run "queue-100" RBT_RECON_QUEUE=1
  run "queue-60" RBT_RECON_QUEUE=1 RBT_RECON_QUEUE_WIDTH=60
  run "queue-200" RBT_RECON_QUEUE=1 RBT_RECON_QUEUE_WIDTH=200
  [ $rep = 1 ] && run "queue-400" RBT_RECON_QUEUE=1 RBT_RECON_QUEUE_WIDTH=400
  [ $rep = 1 ] && run "flags-always" RBT_RECON_LEVEL=1
done
cat $out
