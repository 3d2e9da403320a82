"""What the decoded-picture-hash options cost on one full-size GOF (blocking call): md5_sei (hash SEI in the output) and verify_md5 (check of the input's hash SEI).
On the GPU box: python tools/md5_cost.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rbt_lib
R = rbt_lib.module()
ctx = R.Context(device=0)
man = json.load(open(os.path.join(ROOT, "tests", "golden", "hm_r5_manifest.json")))["1280x1280_f32"]
s = {k: open(os.path.join(ROOT, "tests", "golden", v["file"]), "rb").read() for k, v in man["streams"].items()}
streams = [s["occ"], s["geo"], s["attr"]]
for md5_sei, verify in ((0, 0), (1, 0), (0, 1), (1, 1)):
    P = R.StreamParams
    params = [P(0, 8, 4, 5, -1, md5_sei, verify), P(1, 24, 4, 5, -1, md5_sei, verify), P(19, 32, 4, 5, -1, md5_sei, verify)]
    ctx.transcode_gof(streams, params)
    t0 = time.perf_counter(); out = ctx.transcode_gof(streams, params); dt = time.perf_counter() - t0
    st = ctx.stats()
    print(f"md5_sei={md5_sei} verify_md5={verify}: {1000 * dt:.1f} ms per GOF (gpu {st['gpu_ms']:.1f}, d2h {st['d2h_ms']:.1f}, host pack {st['host_pack_ms']:.1f}), {sum(len(o) for o in out)} bytes")
