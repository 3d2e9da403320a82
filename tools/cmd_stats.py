"""What the decoder's command lists hold (CPU, host emulation build: tests/hostemu RBT_HOSTEMU_CMD_STATS=1): transform units of the first n pictures of a sub-bitstream of
the CTC fixture by size, prediction kind and coded planes - the census behind the per-block instruction counts of profiles/r04_ablate.txt.
  python tools/cmd_stats.py geo|attr [n_pictures=4]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["RBT_LIB_PATH"] = os.path.join(ROOT, "tests", "hostemu", "librbt_hostemu.so")
os.environ["RBT_HOSTEMU_CMD_STATS"] = "1"
import rbt_lib
R = rbt_lib.module(); gs = rbt_lib.module_file("gof_shard")
man = json.load(open(os.path.join(ROOT, "tests", "golden", "hm_r5_manifest.json")))["1280x1280_f32_ctc"]
kind = sys.argv[1]; npic = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s = open(os.path.join(ROOT, "tests", "golden", man["streams"][kind]["file"]), "rb").read()
R.Context(device=0).decode(gs.first_pictures(s, npic), verify_md5=False)      # the histogram is printed at exit (stderr)
