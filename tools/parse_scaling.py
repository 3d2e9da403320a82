"""Development probe (GPU box): how long the merged entropy-decoding launch of ONE job takes as the job carries more copies of the attribute sub-bitstream of the benchmark
GOF (64 slices each, 32 of them IDR slices of ~150 KB): waves in flight vs duration, nothing else on the GPU. python tools/parse_scaling.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import rbt_lib
R = rbt_lib.module()
ctx = R.Context(device=0)
attr = open(os.path.join(ROOT, "tests", "golden", "hm_r5_1280x1280_f32_attr.annexb"), "rb").read()
geo = open(os.path.join(ROOT, "tests", "golden", "hm_r5_1280x1280_f32_geo.annexb"), "rb").read()
P = R.StreamParams
ctx.set_depth(1)
for n in (1, 2, 4, 8, 12, 16, 20):
    streams = [bytes(bytearray(attr)) for _ in range(n)] + [bytes(bytearray(geo))]           # distinct buffers (identical ones are decoded once); + one geometry stream: several pipelines, merged launch
    params = [P(R.RBT_VIDEO_ATTRIBUTE, 32, 4, 5, -1, 0, 0) for _ in range(n)] + [P(R.RBT_VIDEO_GEOMETRY, 24, 4, 5, -1, 0, 0)]
    try:
        ctx.transcode_gof(streams, params)
        t0 = time.perf_counter(); ctx.transcode_gof(streams, params); dt = time.perf_counter() - t0
        st = ctx.stats()
        print(f"attribute streams {n:2d}: parser waves {64 * n + 64:5d}  parse launch {st['k_parse_ms']:7.1f} ms  recon {st['k_recon_ms']:7.1f}  encode {st['k_encode_ms']:7.1f}  call {1000 * dt:7.1f} ms", flush=True)
    except Exception as e:
        print(n, "failed:", e, flush=True); break
    ctx.trim()
# beyond one wave per SIMD (1024): several such jobs side by side
ctx.set_depth(4)
streams = [bytes(bytearray(attr)) for _ in range(12)]; params = [P(R.RBT_VIDEO_ATTRIBUTE, 32, 4, 5, -1, 0, 0) for _ in range(12)]
for jobs in (1, 2, 3, 4):
    for rep in range(2):
        t0 = time.perf_counter()
        hs = [ctx.submit_gof(streams, params) for _ in range(jobs)]
        sts = []
        for h in hs:
            ctx.wait_gof(h); sts.append(ctx.stats())
        dt = time.perf_counter() - t0
    print(f"jobs {jobs}: parser waves {768 * jobs:5d}  parse launches " + " ".join(f"{s_['k_parse_ms']:6.1f}" for s_ in sts) + "  recon " + " ".join(f"{s_['k_recon_ms']:6.1f}" for s_ in sts) + f"  all {1000 * dt:7.1f} ms", flush=True)
