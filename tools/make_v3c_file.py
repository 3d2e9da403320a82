"""Writes a V3C sample stream file around the committed HM-like fixture GOF (tests/golden/hm_r5_*.annexb) repeated as a sequence of N point-cloud frames
(gof_shard.make_sequence / wrap_v3c; placeholder parameter set and atlas units) - a stand-in for a real longdress_r5.bin to hand to `bench.py --v3c-input`
or `rbt_pipeline --v3c`. Host only (librbt.so's stream conversions need no GPU).   python tools/make_v3c_file.py out.bin [frames=300]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rbt_lib
R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
man = json.load(open(os.path.join(ROOT, "tests", "golden", "hm_r5_manifest.json")))["1280x1280_f32"]
streams = {k: open(os.path.join(ROOT, "tests", "golden", v["file"]), "rb").read() for k, v in man["streams"].items()}
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
seq = gs.make_sequence([streams["occ"], streams["geo"], streams["attr"]], frames, 32)
data = gs.wrap_v3c(R, seq)
open(sys.argv[1], "wb").write(data)
print(f"{sys.argv[1]}: {len(data)} bytes, {len(seq)} GOFs, {frames} point-cloud frames; {R.v3c_stats(data)}")
