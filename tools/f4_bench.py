"""Development probe: the benchmark GOF with and without occupancy-aware coding (rbt_stream_params.occupancy_rd) at several pipeline depths: ms per GOF and the
kernel group timings of the last job (rbt_get_stats). GPU box only.   python tools/f4_bench.py [depths...]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rbt_lib

R = rbt_lib.module()
gs = rbt_lib.module_file("gof_shard")
ctx = R.Context(device=0)
streams = [open(os.path.join(ROOT, "tests", "golden", f"hm_r5_1280x1280_f32_{k}.annexb"), "rb").read() for k in ("occ", "geo", "attr")]
depths = [int(x) for x in sys.argv[1:]] or [1, 4, 16]
for rd in (0, 1):
    params = gs.rate_params(R, 3, occupancy_rd=rd)
    for d in depths:
        ctx.set_depth(d)
        n = max(8, 4 * d)
        for rep in range(2):                       # first pass: arenas
            q = []; t0 = time.perf_counter()
            for i in range(n):
                if len(q) == d: outs = ctx.wait_gof(q.pop(0))
                q.append(ctx.submit_gof(streams, params))
            while q: outs = ctx.wait_gof(q.pop(0))
            dt = time.perf_counter() - t0
        st = ctx.stats()
        print(f"occupancy_rd {rd} depth {d:2d}: {1000 * dt / n:7.2f} ms/GOF {32 * n / dt:7.1f} fps  out {sum(len(o) for o in outs)} B | " +
              " ".join(f"{k[2:-3]} {v:.1f}" for k, v in st.items() if k.startswith("k_")) + f" | gpu {st['gpu_ms']:.1f} total {st['total_ms']:.1f}", flush=True)
