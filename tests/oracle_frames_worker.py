"""Worker of the full-size parity tests (tests/test_gpu_fullsize.py): transcodes the point-cloud frames `start, start + step, ...` of the sample file with the CPU oracle
(oracle_transcode_data on one frame's occupancy picture, geometry I/P pair and attribute I/P pair: the frames of a GOF are closed GOPs in all three sub-bitstreams, so the
per-frame outputs concatenate to the whole stream's) and writes them to an .npz. Started as a fresh process: it never touches the GPU."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O

src, dst, start, step, gq, aq, rows, occ_rd = sys.argv[1], sys.argv[2], *[int(x) for x in sys.argv[3:9]]
z = np.load(src)
out = {}
for k in range(start, int(z["n"]), step):
    o, g, a = O.transcode_data([z[f"o{k}"].tobytes(), z[f"g{k}"].tobytes(), z[f"a{k}"].tobytes()], [(0, 8, 4, 5, rows, 0, 0), (1, gq, 4, 5, rows, 0, occ_rd), (19, aq, 4, 5, rows, 0, occ_rd)])
    out[f"o{k}"], out[f"g{k}"], out[f"a{k}"] = np.frombuffer(o, np.uint8), np.frombuffer(g, np.uint8), np.frombuffer(a, np.uint8)
np.savez(dst, **out)
