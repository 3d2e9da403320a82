"""Development probe: decodes the attribute sub-bitstream of the benchmark GOF's first point-cloud frame with the profiling build (make -C rabbit-transcoding_amd librbt_prof.so;
RBT_LIB_PATH=.../librbt_prof.so) - the parser prints its cycle stamps per slice. GPU box only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rbt_lib
R = rbt_lib.module()
gs = rbt_lib.module_file("gof_shard")
ctx = R.Context(device=0)
s = gs.split_pairs(open(os.path.join(ROOT, "tests", "golden", "hm_r5_1280x1280_f32_attr.annexb"), "rb").read())[0]
dec = ctx.decode(s)
print("decoded", dec[0].shape, "fail", dec[5], flush=True)
