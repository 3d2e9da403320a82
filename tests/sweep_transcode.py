"""Randomised parity sweep (not collected by pytest): random sizes / CTB sizes / slice structures / QPs / input encoders (RBT-E1, the HM-like mode) / occupancy-aware
coding / presets through rbt_transcode_gof vs the oracle's transcodeData. Host emulation by default, SWEEP_GPU=1 on an MI355X: python tests/sweep_transcode.py
(SWEEP_N cases, SWEEP_SEED)."""
import sys, os; sys.path.insert(0, 'tests')
import numpy as np, rbt_lib, oracle_lib as O, synth
R = rbt_lib.module()
ctx = R.Context(device=0) if os.environ.get("SWEEP_GPU") else R.Context(lib_path=rbt_lib.HOSTEMU_LIB)
bad = 0
r = np.random.default_rng(int(os.environ.get("SWEEP_SEED", "7")))
for it in range(int(os.environ.get("SWEEP_N", "36"))):
    w = int(r.choice([64, 96, 128, 192, 256])); h = int(r.choice([64, 96, 128, 160]))
    n_pc = int(r.choice([1, 2])); seed = int(r.integers(1, 10000))
    geo, attr, occ = synth.make_gof(w, h, n_pc, seed)
    if r.random() < 0.2:      # noise instead of maps: every mode, every level size
        geo = r.integers(0, 1024, geo.shape).astype(np.uint16); attr = r.integers(300, 700, attr.shape).astype(np.uint16)
    lc = int(r.choice([4, 5, 6])); rows = int(r.choice([0, 1, 2, -1, -1])); qg = int(r.choice([20, 24, 28, 32, 40])); qa = int(r.choice([27, 32, 37, 42]))
    lcin = int(r.choice([4, 5, 6])); rin = int(r.choice([0, 1, 3, -1]))
    if r.random() < 0.3 and w % 16 == 0 and h % 16 == 0:
        sg, _ = O.encode_hm(geo, w, h, 10, 16); sa, _ = O.encode_hm(attr, w, h, 10, 22)
    else:
        sg, _ = O.encode(geo, w, h, 10, 16, gop=2, log2_ctb=lcin, rows_per_slice=rin)
        sa, _ = O.encode(attr, w, h, 10, 22, gop=2, log2_ctb=lcin, rows_per_slice=rin)
    so, _ = O.encode(occ, w // 2, h // 2, 8, 8, gop=1, lossless=1, i_qp_offset=0, log2_ctb=lcin, rows_per_slice=rin)
    f4 = int(r.random() < 0.4); pg = int(r.random() < 0.3); pa = int(r.random() < 0.3); prec = 4 if r.random() < 0.85 else 2
    P = R.StreamParams
    params = [P(0, 8, prec, lc, rows, 1, 0, 0, 0), P(1, qg, prec, lc, rows, 1, 0, f4, pg), P(19, qa, prec, lc, rows, 1, 0, f4, pa)]
    outs = ctx.transcode_gof([so, sg, sa], params)
    exp = O.transcode_data([so, sg, sa], [(p.video_type, p.qp, p.occupancy_precision, p.log2_ctb, p.ctb_rows_per_slice, p.md5_sei, p.occupancy_rd, p.preset) for p in params])
    ok = all(a == b for a, b in zip(outs, exp))
    if not ok: bad += 1; print("MISMATCH", it, w, h, n_pc, seed, lc, rows, qg, qa, lcin, rin, f4, pg, pa, prec, [a == b for a, b in zip(outs, exp)], flush=True)
print("done bad", bad)
