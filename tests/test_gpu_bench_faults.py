"""bench.py at N > 1 with a rank that dies or stops responding inside a secondary leg: rank 0 must still
print the ONE JSON line -- primary figures intact, an error field naming what happened -- inside the legs' deadline, and leave.  Two ranks on
ONE GPU over gloo (bench.py's GVQA_BENCH_ONE_DEVICE / GVQA_BENCH_BACKEND hook), started WITHOUT torch.distributed.run -- its agent would
tear rank 0 down when rank 1 exits, which is not what is under test."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(fault, deadline_s=25):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GVQA_BENCH_ONE_DEVICE="1", GVQA_BENCH_BACKEND="gloo", GVQA_BENCH_SECONDARY_DEADLINE_S=str(deadline_s), GVQA_BENCH_TEST_FAULT=fault)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                                       "--no-cpu-baseline", "--no-pmc", "--no-extras"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    return procs


@pytest.mark.parametrize("fault", ["1:exit", "1:hang"])
def test_bench_line_survives_a_rank_lost_inside_a_secondary_leg(fault):
    deadline = 25
    t0 = time.time()
    procs = _launch(fault, deadline)
    try:
        out0, err0 = procs[0].communicate(timeout=240)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    took = time.time() - t0
    lines = [l for l in out0.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, (out0[-2000:], err0[-2000:])
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["ms_per_step"] > 0 and res["rccl_ranks_seen"]["all_reduce_of_ones"] == 2.0
    assert res.get("secondary_leg_errors"), res.keys()                  # the error field: what the lost rank cost the line
    keys = list(res)
    assert keys.index("rccl_ranks_seen") < keys.index("config") < keys.index("roofline")      # what the scaling record needs comes first
    assert procs[0].returncode == 0
    assert took < 120 + deadline, took                                  # model set-up + primary region + the legs' deadline, not a collective's 30-minute timeout
