"""The unit-level reference-sample map of the CTB chains (csrc/rbt_recon.h rc_nb_map / rc_nb_source / rc_z_before, round 4) against the sample-by-sample rule of H.265
8.4.4.2.2, compiled as host code (tests/hostemu/nb_map_check.cpp): every block size, luma and chroma units, masks of one run, several runs, none."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_unit_map_equals_the_sample_rule():
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "nb_map_check")
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-o", exe, os.path.join(HERE, "hostemu", "nb_map_check.cpp")], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
