"""The loop filters have two forms each: in-place deblocking launches + SAO (default), or - decoder, RBT_FUSED_LF=1 - one launch through LDS tiles and - encoder,
RBT_FUSED_ENC_LF=1 - deblocking inside the SAO kernel. Every combination must give the oracle's bytes; the switches are read once per
process, so each runs in a worker (tests/lf_modes_worker.py)."""
import os
import subprocess
import sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
MODES = [("1", "1"), ("0", "1"), ("1", "0")]          # the default (both unfused) is what every other test runs


def run(kind, dec, enc):
    env = dict(os.environ, RBT_FUSED_LF=dec, RBT_FUSED_ENC_LF=enc)
    r = subprocess.run([sys.executable, os.path.join(HERE, "lf_modes_worker.py"), kind], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dec,enc", MODES)
def test_loop_filter_modes_host_build(dec, enc):
    run("hostemu", dec, enc)


@pytest.mark.gpu
@pytest.mark.parametrize("dec,enc", MODES)
def test_loop_filter_modes_gpu(dec, enc):
    run("gpu", dec, enc)
