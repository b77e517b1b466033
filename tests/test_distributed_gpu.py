"""The multi-GPU orchestration (gst-plugin-rs_amd/distributed.py) with the REAL pieces on the one-GPU box: per-rank compute
= the HIP band kernels through the C ABI, collective = RCCL (torch.distributed backend "nccl", world_size 1), in a fresh
process like a bench.py worker (tests/rccl_world1_worker.py).  The world-2 logic is covered on CPU by
tests/test_distributed_cpu.py (gloo); this test makes sure the RCCL communicator is really initialised by this code and
that HIP kernels and all-reduces are ordered correctly on one stream.  Results are compared with the oracle here."""
import json
import os
import subprocess
import sys

import pytest

from tests import frames
from tests import oracle_binding as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def worker_result(gpu):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, r.stdout[-3000:]
    return json.loads(line[-1][len("RESULT "):])


def test_rccl_communicator_is_initialised(worker_result):
    assert worker_result["backend"] == "nccl" and worker_result["rccl_ranks"] == 1


def test_videocompare_sharded_hip_bands_over_rccl(worker_result):
    """band kernels -> ONE all-reduce of n_pads x 64 sums -> bits + Hamming on every rank."""
    w, h = 1920, 1080
    a = frames.random_frame(0x5EED0001, w, h)
    b = a.copy()
    b[::3, 0:w * 4:16] ^= 0x3C
    pads = [a, b, 255 - a, a]
    hs = [orc.blockhash(f, w, h, w * 4, "RGBA")[1] for f in pads]
    want = [float(orc.hamming(hs[0], x)) for x in hs[1:]]
    assert worker_result["videocompare"] == want and want[2] == 0.0


def test_ssim_and_colordetect_sharded_over_rccl(worker_result):
    w, h = 320, 240
    a = frames.random_frame(0x5EED0002, w, h)
    b = a.copy()
    b[5::7, 3:w * 4:11] ^= 0x15
    rc, want, _ = orc.ssim_distance(a, b, w, h, w * 4, w * 4, "RGBA")
    assert rc == 0 and worker_result["ssim"] == pytest.approx(want, rel=1e-9, abs=1e-12)
    rc, want_pal = orc.colordetect_palette(a, "RGBA", 10, 5)
    assert rc >= 0 and worker_result["palette"] == [int(x) for x in want_pal]
