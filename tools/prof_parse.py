"""Cycle breakdown of the slice parser on the largest slice of the benchmark fixture (the attribute IDR of point-cloud frame 0):
RBT_LIB_PATH=rabbit-transcoding_amd/librbt_prof.so python tools/prof_parse.py   (on the GPU box; build with make -C rabbit-transcoding_amd librbt_prof.so)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rbt_lib
R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
man = json.load(open(os.path.join(ROOT, "tests/golden/hm_r5_manifest.json")))["1280x1280_f32"]
kind = sys.argv[1] if len(sys.argv) > 1 else "attr"
s = gs.split_pairs(open(os.path.join(ROOT, "tests/golden", man["streams"][kind]["file"]), "rb").read())[0]
ctx = R.Context(device=0)
ctx.decode(s, verify_md5=False)
