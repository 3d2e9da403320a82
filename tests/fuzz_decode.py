"""Robustness sweep (not collected by pytest): damaged streams through the host build of the kernel bodies under AddressSanitizer. A damaged stream may decode to
garbage or be rejected, but no table read or write may leave its buffer - on the GPU that would be a fault that can take the whole node down.
  g++ ... -fsanitize=address -DRBT_HOSTEMU -o /tmp/librbt_hostemu_asan.so (see tests/hostemu/Makefile for the sources)
  ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) [RBT_FUZZ_TRANSCODE=1 | RBT_FUZZ_V3C=1] python tests/fuzz_decode.py [first_seed] [n_seeds]
RBT_FUZZ_V3C=1: damaged V3C sample streams (unit sizes, unit headers, video payloads) through rbt_v3c_index / rbt_transcode_v3c."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O, rbt_lib, synth
R = rbt_lib.module()
ctx = R.Context(lib_path=os.environ.get("RBT_FUZZ_LIB", "/tmp/librbt_hostemu_asan.so"))
first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 200
m = synth.make_maps(192, 128, 9)
streams = [O.encode_hm(m["geo"], 192, 128, 10, 16, p_qp_offset=-3)[0], O.encode_hm(m["attr"], 192, 128, 10, 22)[0],
           O.encode(np.zeros((5, 96 * 64 * 3 // 2), np.uint16), 96, 64, 10, qp=30, gop=2, stress_seed=7, log2_ctb=0)[0],
           O.encode(np.zeros((5, 128 * 80 * 3 // 2), np.uint16), 128, 80, 8, qp=30, gop=2, stress_seed=12, log2_ctb=0)[0],
           O.encode(np.zeros((5, 96 * 64 * 3 // 2), np.uint16), 96, 64, 10, qp=30, gop=2, stress_seed=13, log2_ctb=0)[0],      # wavefront rows + dependent slice segments
           O.encode(np.zeros((5, 96 * 64 * 3 // 2), np.uint16), 96, 64, 10, qp=30, gop=2, stress_seed=17, log2_ctb=0)[0],      # wavefront rows, entry points
           O.encode(m["geo"], 192, 128, 10, 24, gop=2, rows_per_slice=-1)[0]]                                                   # RBT-E1 wavefront mode
caught = ok = 0
if os.environ.get("RBT_FUZZ_V3C"):
    import v3c_synth as V
    units = V.gof_units(V.gof_streams(64, 64, 1, 3), 1, aux=True) + V.gof_units(V.gof_streams(96, 64, 1, 4), 2)
    base = [V.sample_stream(units, p) for p in (1 + 1, 3, 4, 8)]
    for seed in range(first, first + n):
        r = np.random.default_rng(seed)
        s = bytearray(base[seed % 4])
        heads = [0]                                             # byte positions of the stream header, the unit sizes and the unit headers
        for u in R.v3c_index(bytes(s), ctx.L): heads += list(range(u["offset"] - (s[0] >> 5) - 1, u["offset"] + 4))
        for _ in range(int(r.integers(1, 6))):
            k = int(r.choice(heads)) if seed % 3 else int(r.integers(0, len(s)))
            s[k] = int(r.integers(0, 256)) if seed % 2 else s[k] ^ (1 << int(r.integers(0, 8)))
        if seed % 5 == 0: s = s[: int(r.integers(1, len(s)))]
        try:
            ctx.transcode_v3c(bytes(s), 24, 32, occupancy_precision=4 if seed % 7 else 2)
            ok += 1
        except R.RbtError:
            caught += 1
    print(f"v3c seeds {first}..{first + n - 1}: {caught} rejected, {ok} transcoded to something, no memory error")
    sys.exit(0)
for seed in range(first, first + n):
    r = np.random.default_rng(seed)
    s = bytearray(streams[seed % len(streams)])
    mode = seed % 5
    lo = 0 if mode == 4 else len(s) // 10                       # mode 4 also damages parameter sets and slice headers
    for k in r.integers(lo, len(s) - 4, int(r.integers(1, 40))):
        if mode == 3: s[int(k)] = 0xFF
        elif mode == 2: s[int(k)] ^= 1 << int(r.integers(0, 8))
        else: s[int(k)] = int(r.integers(0, 256))
    if mode == 1: s = s[: int(r.integers(len(s) // 3, len(s)))]
    try:
        if os.environ.get("RBT_FUZZ_TRANSCODE"): ctx.transcode_substream(bytes(s), R.RBT_VIDEO_ATTRIBUTE, 32, md5_sei=0, rows_per_slice=-1 if seed % 2 else 1)     # the chained pipeline: the encoder runs on whatever the decoder left
        else: ctx.decode(bytes(s), verify_md5=False)
        ok += 1
    except R.RbtError:
        caught += 1
print(f"seeds {first}..{first + n - 1}: {caught} rejected, {ok} decoded to something, no memory error")
