#!/usr/bin/env python3
"""What the HIP runtime itself allocates on the device while librbt runs (scratch memory of the hardware queues, code objects, signals): free HBM before the first
job against free HBM after jobs of the headline's shape have run on all 16 queues and every arena has been handed back (rbt_trim). The HBM reserve of dev_alloc
(csrc/rbt_kernels.hip, RBT_HBM_RESERVE_MB) must cover it. Prints one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import rbt_lib  # noqa: E402


def probe(full_size=True):
    R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
    ctx = R.Context(device=0)
    m0 = ctx.device_memory()
    if full_size:
        man = json.load(open(os.path.join(ROOT, "tests", "golden", "hm_r5_manifest.json")))["1280x1280_f32_ctc"]
        gof = [gs.first_pictures(open(os.path.join(ROOT, "tests", "golden", man["streams"][k]["file"]), "rb").read(), n) for k, n in (("occ", 4), ("geo", 8), ("attr", 8))]
    else:
        import v3c_synth as V
        gof = V.gof_streams(256, 256, 2, 3)
    P = gs.rate_params(R, 3)
    peak = 0
    for depth in (16, 4, 1):                     # every stream-to-queue mapping the library uses
        ctx.set_depth(depth)
        for occ_rd in (0, 1):
            jobs = [ctx.submit_gof(gof, gs.rate_params(R, 3, occupancy_rd=occ_rd)) for _ in range(depth)]
            peak = max(peak, ctx.device_memory()["in_use"])
            for j in jobs: ctx.wait_gof(j)
    ctx.decode(gof[1]); ctx.trim()
    m1 = ctx.device_memory()
    out = {"free_before_MB": m0["free"] >> 20, "free_after_trim_MB": m1["free"] >> 20, "runtime_keeps_MB": (m0["free"] - m1["free"]) >> 20, "cached_after_trim_MB": m1["cached"] >> 20,
           "reserve_MB": m1["reserve"] >> 20, "arenas_at_peak_MB": peak >> 20, "total_MB": m1["total"] >> 20}
    ctx.close()
    return out


if __name__ == "__main__":
    print(json.dumps(probe("--small" not in sys.argv)))
