# Job shapes (GOFs per job, jobs in flight) for the 300-frame sequence walk and its container form (bench.py legs sequence_walk / container), RBT_WALK_SHAPE="G,D".
[ $# -eq 0 ] && set -- "1,10" "2,5" "3,4" "4,3" "5,2" "2,8"
for shape in "$@"; do
  RBT_WALK_SHAPE=$shape python bench.py --steps 4 --warmup 2 --cpu-sample 0 --multi-gof 0 --sweep 0 --quality 0 --fanout-gofs 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['sequence_walk']; print('shape $shape walk', w['value'], w['stitched_equals_unsharded'], 'container', w['container']['value'], w['container']['video_units_equal_walk'])" >> gpurun_out/walk.log
done
cat gpurun_out/walk.log
