"""Decode / transcode latency of a wavefront stream in x265's form (one slice segment per picture, CTB rows behind entry point offsets), which the
decoder cuts into one parse task per row on the host. The stream is made by the oracle's encoder (rows_per_slice=-2) from the fixture's attribute
pictures, so this script lives under tests/ (the oracle is test infrastructure). On the GPU box:
  python tests/wavefront_entry_bench.py; RBT_WPP_PARALLEL=0 python tests/wavefront_entry_bench.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rbt_lib
import oracle_lib as O
R = rbt_lib.module()
ctx = R.Context(device=0)
src = open(os.path.join(ROOT, "tests/golden/hm_r5_1280x1280_f32_attr.annexb"), "rb").read()
N = int(os.environ.get("PICTURES", "16"))
yuv, w, h, bd, _, _ = ctx.decode(src, verify_md5=False)
fs = w * h * 3 // 2
pics = np.asarray(yuv)[:N]
t0 = time.perf_counter(); bs, rec = O.encode(pics, w, h, 10, 32, gop=2, log2_ctb=5, rows_per_slice=-2); te = time.perf_counter() - t0
best = 1e9
for it in range(5):
    t0 = time.perf_counter(); d = ctx.decode(bs, verify_md5=False); best = min(best, time.perf_counter() - t0)
d = ctx.decode(bs, verify_md5=True)
same = bool(np.array_equal(np.asarray(d[0]).reshape(-1), np.asarray(rec).reshape(-1)))
t2 = 1e9
for it in range(3):
    t0 = time.perf_counter(); o2 = ctx.transcode_substream(bs, R.RBT_VIDEO_ATTRIBUTE, 42, log2_ctb=5, rows_per_slice=-1, md5_sei=0); t2 = min(t2, time.perf_counter() - t0)
print(f"RBT_WPP_PARALLEL={os.environ.get('RBT_WPP_PARALLEL', '1')}: {len(bs)} bytes, {N} pictures in {len(O.slice_headers(bs))} slice segments (oracle encode {te:.1f} s); "
      f"decode {1000 * best:.1f} ms (hash SEI: {d[4]} checked, {d[5]} failed, == oracle reconstruction: {same}); transcode of it {1000 * t2:.1f} ms, {len(o2)} bytes")
