"""-m gpu: the RCCL code paths of the data-parallel step and of the row-sharded ranking, executed on the one GPU of the test
box in a 1-rank "nccl" process group with SRFRD_FORCE_EXCHANGE=1 (tools/nccl_single_rank.py, a child process).  Every other
data-parallel test runs the gloo emulation (tests/test_gpu_dp.py, tests/test_dp_gloo.py); this one executes
reduce_scatter_tensor, the in-place all_gather_into_tensor, the asynchronous statistics all-reduce, HIP-graph capture with the
collectives inside and ranker._all_gather's nccl arm - each form bit-equal to the single-rank step."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_arms_in_a_group_of_one_equal_the_single_path():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_single_rank.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, f"no report (rc {r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    rep = json.loads(lines[-1])
    assert r.returncode == 0 and rep["ok"], json.dumps(rep, indent=1)
    assert rep["backend"] == "nccl" and rep["world"] == 1
    for key in ("sharded/graph", "sharded/eager", "allreduce/graph", "allreduce/eager"):
        f = rep["forms"][key]
        assert f["native"] and f["loss_bit_equal"] and f["weights_bit_equal"], (key, f)
    # which graph form ran is reported, not asserted: "one" (collectives captured) where torch + RCCL allow it
    assert rep["forms"]["sharded/graph"]["graph_form"] in ("one", "split")
