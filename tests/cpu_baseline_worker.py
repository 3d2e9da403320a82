"""Worker of bench.py's cpu_baseline leg (multi-core row): transcodes the point-cloud frames `start, start + step, ...` of the
sample file with the CPU oracle (one frame = its occupancy picture, its geometry I/P pair and its attribute I/P pair: frames of a
GOF are independent of each other) and prints the seconds it spent. Started as a fresh process: it never touches the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O

path, start, step = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 1
z = np.load(path)
n = int(z["n"])
O.lib()
t0 = time.perf_counter()
done = 0
for k in range(start, n, step):
    O.transcode_substream(z[f"o{k}"].tobytes(), 0, 8, md5_sei=0, rows_per_slice=rows)
    O.transcode_substream(z[f"g{k}"].tobytes(), 1, 24, md5_sei=0, rows_per_slice=rows)
    O.transcode_substream(z[f"a{k}"].tobytes(), 19, 32, md5_sei=0, rows_per_slice=rows)
    done += 1
print(f"{done} {time.perf_counter() - t0:.4f}")
