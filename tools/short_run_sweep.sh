# The driver times `bench.py --steps 20 --warmup 5`: a run that is all ramp-up and drain. Job shapes for it (GOFs per job, jobs in flight), headline leg only.
# usage (GPU box): bash tools/short_run_sweep.sh ["G D" ...]
[ $# -eq 0 ] && set -- "0 16" "2 16" "3 16" "4 16" "3 8" "5 16"
for cfg in "$@"; do
  set -- $cfg
  python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --gofs-per-job $1 --in-flight $2 --cpu-sample 0 --multi-gof 0 --sweep 0 --quality 0 --walk-frames 0 --fanout-gofs 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G D', '$1', '$2', 'steps', d['steps'], d['value'], d['ms_per_step'], d['config'].get('jobs_in_flight'), d['config'].get('gofs_per_job'))" >> gpurun_out/short.log
done
cat gpurun_out/short.log
