"""Decode latency of a wavefront stream (this library's own R3 output of the fixture's attribute sub-bitstream: 64 pictures of 1280x1280, one dependent
slice segment per CTB row) - one wave per CTB row (default) against one wave per slice (RBT_WPP_PARALLEL=0). On the GPU box:
  python tools/wavefront_decode_bench.py; RBT_WPP_PARALLEL=0 python tools/wavefront_decode_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rbt_lib
R = rbt_lib.module()
ctx = R.Context(device=0)
src = open(os.path.join(ROOT, "tests/golden/hm_r5_1280x1280_f32_attr.annexb"), "rb").read()
out = ctx.transcode_substream(src, R.RBT_VIDEO_ATTRIBUTE, 32, log2_ctb=5, rows_per_slice=-1, md5_sei=1)
best = 1e9
for it in range(5):
    t0 = time.perf_counter(); d = ctx.decode(out, verify_md5=False); best = min(best, time.perf_counter() - t0)
d = ctx.decode(out, verify_md5=True)
t0 = time.perf_counter(); o2 = ctx.transcode_substream(out, R.RBT_VIDEO_ATTRIBUTE, 42, log2_ctb=5, rows_per_slice=-1, md5_sei=0); t2 = time.perf_counter() - t0
print(f"RBT_WPP_PARALLEL={os.environ.get('RBT_WPP_PARALLEL', '1')}: {len(out)} bytes, 64 pictures; decode {1000 * best:.1f} ms (hash SEI: {d[4]} checked, {d[5]} failed); R3 -> R1 transcode of it {1000 * t2:.1f} ms, {len(o2)} bytes")
