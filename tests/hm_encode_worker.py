"""Worker of tests/test_gpu_fullsize.py: codes one piece of a synthetic GOF with the oracle's HM-like encoder in the CTC stream structure (tests/golden/make_hm_gof.py _encode)
and writes it to a file. argv: kind (occ | geo | attr), seed, first point-cloud frame, frames, width, height, output path. A fresh process: it never touches the GPU."""
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "golden"))
import make_hm_gof
import synth

kind, seed, first, n, w, h, dst = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
geo, attr, occ = synth.make_gof_maps(w, h, first + n, seed)
frames = {"geo": geo[2 * first:2 * (first + n)], "attr": attr[2 * first:2 * (first + n)], "occ": occ[first:first + n]}[kind]
ww, hh = (w // 2, h // 2) if kind == "occ" else (w, h)
open(dst, "wb").write(make_hm_gof._encode((kind, frames, ww, hh, 1, first * (1 if kind == "occ" else 2))))
