"""Randomised soak of the asynchronous job API (not collected by pytest): random GOFs (sizes, CTB sizes, slice structures, QPs, with and
without the input MD5 check) walked through rbt_submit_gof / rbt_wait_gof at random depths and collected in random order; every output
must equal the oracle's. Host emulation by default, SWEEP_GPU=1 on an MI355X: python tests/sweep_async.py"""
import sys, os; sys.path.insert(0, 'tests')
import numpy as np, rbt_lib, oracle_lib as O, synth
R = rbt_lib.module()
ctx = R.Context(device=0) if os.environ.get("SWEEP_GPU") else R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
r = np.random.default_rng(11)
P = R.StreamParams
pool = []
for it in range(12):
    w = int(r.choice([64, 128, 192, 256])); h = int(r.choice([64, 96, 128, 160])); h -= h % 32
    n_pc = int(r.choice([1, 2, 3])); seed = int(r.integers(1, 10000))
    geo, attr, occ = synth.make_gof(w, h, n_pc, seed)
    lc = int(r.choice([4, 5, 6])); rows = int(r.choice([0, 1, 2, -1, -1])); qg = int(r.choice([20, 24, 32, 40])); qa = int(r.choice([27, 32, 42]))
    lcin = int(r.choice([4, 5, 6])); rin = int(r.choice([0, 1, 3, -1])); vmd5 = int(r.random() < 0.25)
    ins = [O.encode(occ, w // 2, h // 2, 8, 8, gop=1, lossless=1, i_qp_offset=0, log2_ctb=lcin, rows_per_slice=rin)[0],
           O.encode(geo, w, h, 10, 16, gop=2, log2_ctb=lcin, rows_per_slice=rin)[0], O.encode(attr, w, h, 10, 22, gop=2, log2_ctb=lcin, rows_per_slice=rin)[0]]
    ps = [P(0, 8, 4, lc, rows, 1, vmd5), P(1, qg, 4, lc, rows, 1, vmd5), P(19, qa, 4, lc, rows, 1, vmd5)]
    exp = [O.transcode_substream(ins[0], 0, 8, log2_ctb=lc, rows_per_slice=rows), O.transcode_substream(ins[1], 1, qg, log2_ctb=lc, rows_per_slice=rows),
           O.transcode_substream(ins[2], 19, qa, log2_ctb=lc, rows_per_slice=rows)]
    pool.append((ins, ps, exp))
bad = 0; n = 0
for depth in (1, 2, 3, 4, 5, 6, 8, 11, 16):
    ctx.set_depth(depth)
    q = []
    for step in range(3 * depth + 5):
        if len(q) == depth or (q and r.random() < 0.2):
            k = int(r.integers(0, len(q)))                    # collect a random job, not necessarily the oldest
            job, want = q.pop(k); got = ctx.wait_gof(job); n += 1
            if got != want: bad += 1; print("MISMATCH depth", depth, "step", step)
        ins, ps, exp = pool[int(r.integers(0, len(pool)))]
        q.append((ctx.submit_gof(ins, ps), exp))
    while q:
        job, want = q.pop(int(r.integers(0, len(q)))); got = ctx.wait_gof(job); n += 1
        if got != want: bad += 1; print("MISMATCH depth", depth, "drain")
print("done jobs", n, "bad", bad)
