"""Child process of tests/test_gpu_memory.py (RBT_HBM_RESERVE_MB is read once per process): with the reserve set close to the free memory a full-size job cannot fit ->
RBT_ERR_NOMEM, no abort; the same context then runs a small job correctly. Prints OK."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O
import rbt_lib
import v3c_synth as V

R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ctx = R.Context(device=0)
m = ctx.device_memory()
assert m["reserve"] == int(os.environ["RBT_HBM_RESERVE_MB"]) << 20, m
man = json.load(open(os.path.join(GOLD, "hm_r5_manifest.json")))["1280x1280_f32_ctc"]
big = [gs.first_pictures(open(os.path.join(GOLD, man["streams"][k]["file"]), "rb").read(), n) for k, n in (("occ", 16), ("geo", 32), ("attr", 32))]     # ~2 GB of arenas
ctx.set_depth(4)
for attempt in range(2):                                      # twice: the failure leaves nothing behind that would change the second outcome
    j = ctx.submit_gof(big, gs.rate_params(R, 3))
    try:
        ctx.wait_gof(j)
        print("FAIL: the job fitted", ctx.device_memory()); sys.exit(1)
    except R.RbtError as e:
        assert e.code == -5 and "allocation" in str(e), str(e)
    assert ctx.device_memory()["in_use"] == 0
ctx.trim()
small = V.gof_streams(128, 128, 2, 9)
out = ctx.transcode_gof(small, gs.rate_params(R, 3))
assert out == O.transcode_data(small, [(0, 8, 4, 5, -1, 0), (1, 24, 4, 5, -1, 0), (19, 32, 4, 5, -1, 0)])
# a V3C walk at a depth the memory does not hold: bounded by rbt_device_memory, same bytes
seq = [V.gof_streams(256, 256, 2, 20 + g) for g in range(6)]
data = V.sample_stream([u for g, s in enumerate(seq) for u in V.gof_units(s, 30 + g)], 3)
ctx.set_depth(16)
assert ctx.transcode_v3c(data, 24, 32, gofs_per_job=0) == O.v3c_transcode(data, 24, 32, 4)
print("OK", json.dumps(ctx.device_memory()))
