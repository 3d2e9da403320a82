"""GPU box, under rocprofv3 --pmc (tools/ablate_run.sh): one GOF of the CTC fixture through a measurement build of the library (tools/ablate.sh, RBT_ABLATE mask argv[1]).
argv[2] = decode (geometry and attribute sub-bitstreams decoded, nothing else: the reconstruction masks leave the pictures wrong) | transcode (R5 -> R3; the analysis masks
change decisions only, the output stays a valid stream)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RBT_LIB_PATH"] = os.path.join(ROOT, "rabbit-transcoding_amd", "ablate", "librbt_a%s.so" % sys.argv[1])
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rbt_lib
R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
man = json.load(open(os.path.join(ROOT, "tests", "golden", "hm_r5_manifest.json")))["1280x1280_f32_ctc"]
streams = {k: open(os.path.join(ROOT, "tests", "golden", man["streams"][k]["file"]), "rb").read() for k in ("occ", "geo", "attr")}
ctx = R.Context(device=0)
if sys.argv[2] == "decode":
    for k in ("geo", "attr"): ctx.decode(streams[k], verify_md5=False)
else:
    out = ctx.transcode_gof([streams[k] for k in ("occ", "geo", "attr")], gs.rate_params(R, 3))
    print("bytes out", [len(o) for o in out])
ctx.close()
