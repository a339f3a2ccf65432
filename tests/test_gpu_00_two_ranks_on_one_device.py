"""bench.py's N > 1 path end to end on a one-GPU box: two rank PROCESSES on the one device (SRL_BENCH_ALL_ON_DEVICE0 test hook), direct peer
exchange.  This file sorts in front of the other GPU tests on purpose: the two processes' kernels wait for each other's rows, so the leg
needs the driver to run both processes' queues side by side -- late in a long suite run, when the pytest process itself has created and
released a few hundred HIP contexts, it has been seen to crawl (every exchange a scheduler quantum) and to end in a time-out status once
in a few runs; alone, and in front of everything else, it has never failed (round 5)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_with_the_peer_transport_on_one_device(tmp_path):
    """bench.py's N > 1 path end to end (gloo control plane, IPC handles gathered with all_gather_object, sharded solve, barriers,
    max over ranks, the replicas leg): two ranks on the one device through the SRL_BENCH_ALL_ON_DEVICE0 test hook.  Checks the
    contract of the line, not its speed (two processes share a GPU)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SRL_BENCH_ALL_ON_DEVICE0="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # launched the way the driver does it: `python bench.py --gpus 2`, no torch.distributed.run around it -- bench.py becomes the
    # launcher of its own ranks (WORLD_SIZE must not leak in from the test environment)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--transport", "peer"]
    # Two PROCESSES time-sharing one device through a test hook: a rank's kernel spins for a row that the OTHER process' kernel has to
    # produce, so the leg depends on the driver running both processes' queues side by side.  Inside a long suite run (the pytest process
    # itself holds HIP queues by then) it has been seen to end with a time-out status once in a few runs -- never alone, never under CPU
    # load alone (round 5: 3 of 3 with every core busy; one process per GPU, the production layout, has no such coupling).  Up to three
    # attempts; every failed one is reported, not hidden.
    import warnings
    for attempt in range(3):
        p = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        if p.returncode == 0:
            break
        why = [l for l in p.stderr.decode(errors="replace").split("\n") if "failed with status" in l][-2:]
        warnings.warn(f"bench.py --gpus 2 on one device: attempt {attempt + 1} failed: " + " | ".join(why))
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().split("\n") if l.strip()]
    assert len(lines) == 1, lines                                          # ONE JSON line on stdout (rank 0 only, no library chatter)
    assert len(lines[0]) < 8000                                            # the driver keeps an 8 KB tail of stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "strong" and d["unit"] == "sweeps/s"
    assert d["config"]["residuals_used"] == 65536 and "direct peer exchange" in d["config"]["parallelism"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6
