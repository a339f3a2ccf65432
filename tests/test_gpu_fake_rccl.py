"""The RCCL form of the sharded pass with N > 1 ranks on a one-GPU box.

srl_build_residuals on a context with a communicator sequences (csrc/srl_capi.cpp): fused pass of the rank's shard into a device-side
mailbox -> ncclAllReduce of 50 doubles -> publish kernel; or, when the ordered cut of optimize.cpp:107 can trigger: count kernel ->
ncclAllGather of the per-rank counts -> reduce kernel (budget and mode from the counts of the ranks before it) -> ncclAllReduce ->
publish.  Real RCCL refuses two ranks on one device, so until an N-GPU node runs it that sequencing is executed here through a
stand-in for librccl (tests/fake_rccl: ranks are processes meeting in shared memory, collectives stream-ordered through host
callbacks, sums in rank order), selected explicitly with srl_comm_set_library.  Every rank must end with the single-context result:
counts, stop keypoint, status equal; sums to 1e-12 and BIT-identical across ranks.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import sr_livo_amd as srl
from sr_livo_amd import capi

pytestmark = pytest.mark.gpu

INT_MAX = 2**31 - 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def fake_lib():
    if not os.path.exists(FAKE):
        subprocess.run(["make", "-C", os.path.dirname(FAKE)], check=True, capture_output=True)
    return FAKE


_SCRIPT = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import sr_livo_amd as srl
from sr_livo_amd import capi
rank, G, d, max_res, fused, fake = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
srl.comm_set_library(fake)                               # explicit: this process's nccl* entry points come from the stand-in
assert srl.comm_backend_info()[1] == 99901, srl.comm_backend_info()
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "golden_small.npz"))
ctx = srl.Context(0)
ctx.map_upload(g["map_keys"], g["map_counts"], g["map_xyz"])
ctx.set_fused_reduce(fused)
idp = os.path.join(d, "uid.bin")
if rank == 0:
    uid = srl.Context.comm_unique_id()
    open(idp + ".tmp", "wb").write(uid); os.replace(idp + ".tmp", idp)
t0 = time.time()
while not os.path.exists(idp):
    if time.time() - t0 > 120: raise SystemExit(3)
    time.sleep(0.02)
ctx.comm_init_rank(G, rank, open(idp, "rb").read())
ci = ctx.comm_info()                                     # what bench.py prints for every --gpus N line (srl_comm_info)
assert ci["transport_used"] == "rccl" and ci["nranks"] == G and ci["rank"] == rank and ci["ranks_seen"] == G, ci
ctx.sweep_upload(g["raw"])                               # keeps this rank's contiguous point range (srl_shard_range)
f = capi.make_frame(g["q_pred"], g["t_pred"], g["t_last"])
for _ in range(4):                                       # back-to-back passes: the collectives of pass k + 1 queue behind those of pass k
    neq, rc = ctx.build_residuals(f, srl.default_opts(max_num_residuals=max_res))
np.savez(os.path.join(d, f"out{rank}.npz"), HtH=np.array(neq.HtH), Hth=np.array(neq.Hth), loss=neq.loss_sum, n=neq.num_residuals, last=neq.last_visited,
         pk=neq.sum_candidates, ok=neq.success)
ctx.close()
print("OK", rank, flush=True)
"""


def _ranks(tmp_path, G, max_res, fused, fake):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", _SCRIPT, ROOT, str(r), str(G), str(tmp_path), str(max_res), str(int(fused)), fake],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(G)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append((p.returncode, o.decode(errors="replace")[-2000:]))
    assert all(rc == 0 for rc, _ in outs), outs
    return [np.load(tmp_path / f"out{r}.npz") for r in range(G)]


def _single(golden, max_res):
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        ctx.sweep_upload(golden["raw"])
        neq, _rc = ctx.build_residuals(capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"]), srl.default_opts(max_num_residuals=max_res))
        return neq
    finally:
        ctx.close()


@pytest.mark.parametrize("G,max_res,fused", [(2, INT_MAX, 1), (2, 600, 1), (2, 37, 0), (2, -1, 1), (4, INT_MAX, 0), (4, 600, 1), (4, 37, 1), (4, -1, 0),
                                             (8, INT_MAX, 1), (8, 600, 0), (8, 37, 1), (8, -1, 1)])
def test_rccl_sequencing_with_n_ranks_through_the_stand_in(golden, tmp_path, fake_lib, G, max_res, fused):
    # G processes time-slice ONE GPU and meet in host callbacks of the stand-in: once in ~10 full suite runs a rank of the 8-process case did
    # not reach a collective within the stand-in's bound.  One repetition, reported as a warning with the first attempt's output -- a
    # defect of the sequencing itself fails twice.
    try:
        res = _ranks(tmp_path, G, max_res, fused, fake_lib)
        _check_ranks(golden, res, G, max_res)
        return
    except AssertionError as first:
        import warnings
        warnings.warn(f"stand-in RCCL run with {G} processes repeated after: {str(first)[:1500]}")
        for f in tmp_path.glob("*"):
            f.unlink()
    res = _ranks(tmp_path, G, max_res, fused, fake_lib)
    _check_ranks(golden, res, G, max_res)


def _check_ranks(golden, res, G, max_res):
    ref = _single(golden, max_res)
    for r in range(G):
        assert int(res[r]["n"]) == ref.num_residuals and int(res[r]["last"]) == ref.last_visited and int(res[r]["ok"]) == ref.success
        assert int(res[r]["pk"]) == ref.sum_candidates or max_res != INT_MAX          # candidates visited: every keypoint when nothing cuts
        assert rel(res[r]["HtH"], np.array(ref.HtH)) < 1e-12 and rel(res[r]["Hth"], np.array(ref.Hth)) < 1e-12
        assert np.array_equal(res[r]["HtH"], res[0]["HtH"]) and np.array_equal(res[r]["Hth"], res[0]["Hth"]) and float(res[r]["loss"]) == float(res[0]["loss"])


_SCRIPT_ARMED = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import sr_livo_amd as srl
from sr_livo_amd import capi
rank, G, d, fake, expire = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6])
srl.comm_set_library(fake)
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "golden_small.npz"))
rng = np.random.default_rng(11)
raw = np.concatenate([g["raw"] + rng.normal(0, 2e-3, g["raw"].shape) for _ in range(2 * G)])     # 4 096 keypoints per rank: sixteen-wave workgroups, fused
ctx = srl.Context(0)
ctx.map_upload(g["map_keys"], g["map_counts"], g["map_xyz"])
idp = os.path.join(d, "uid.bin")
if rank == 0:
    uid = srl.Context.comm_unique_id()
    open(idp + ".tmp", "wb").write(uid); os.replace(idp + ".tmp", idp)
t0 = time.time()
while not os.path.exists(idp):
    if time.time() - t0 > 120: raise SystemExit(3)
    time.sleep(0.02)
ctx.comm_init_rank(G, rank, open(idp, "rb").read())
ctx.sweep_upload(raw)
f = capi.make_frame(g["q_pred"], g["t_pred"], g["t_last"])
opts = srl.default_opts(max_num_residuals=2**31 - 1)
ctx.set_armed_launch(1)
for _ in range(3):
    ctx.build_residuals(f, opts)
default_armed = ctx.arm_stats()["armed"]                 # ranks share the device (identities all-gathered at srl_comm_init_rank): nothing armed
ctx.set_armed_launch(2)
if expire:
    ctx.set_arm_linger(host_linger_us=3.6e9, kernel_linger_us=200.0)
s0 = ctx.arm_stats()
res = []
for k in range(8):
    if expire and k == 4 and rank == 0:
        time.sleep(0.05)                                 # rank 0's waiting launch gives up; the record it leaves carries the time-out flag
    neq, rc = ctx.build_residuals(f, opts)
    res.append(np.array(neq.HtH))
s1 = ctx.arm_stats()
ctx.disarm()
np.savez(os.path.join(d, f"out{rank}.npz"), HtH=np.stack(res), n=neq.num_residuals, pk=neq.sum_candidates, default_armed=default_armed,
         fired=s1["fired"] - s0["fired"], armed=s1["armed"] - s0["armed"], cancelled=s1["cancelled"] - s0["cancelled"])
ctx.close()
print("OK", rank, flush=True)
"""


@pytest.mark.parametrize("expire", [0, 1])
def test_rccl_passes_are_armed_behind_the_collective(golden, tmp_path, fake_lib, expire):
    """the fused pass of a sharded sweep behind RCCL arms its successor too -- the launch is enqueued behind this pass's
    all-reduce and publish kernel and fired through the pose box.  Same sums as un-armed passes, bit-identical across ranks and passes.
    expire = 1: rank 0's waiting launch gives up on the device; it contributes an empty record with the time-out flag to the all-reduce, so
    BOTH ranks repeat that pass (a rank repeating alone would leave the collectives of the ranks out of step)."""
    G = 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", _SCRIPT_ARMED, ROOT, str(r), str(G), str(tmp_path), fake_lib, str(expire)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(G)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append((p.returncode, o.decode(errors="replace")[-2000:]))
    assert all(rc == 0 for rc, _ in outs), outs
    res = [np.load(tmp_path / f"out{r}.npz") for r in range(G)]
    rng = np.random.default_rng(11)
    raw = np.concatenate([golden["raw"] + rng.normal(0, 2e-3, golden["raw"].shape) for _ in range(2 * G)])
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        ctx.sweep_upload(raw)
        ref, _rc = ctx.build_residuals(capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"]), srl.default_opts(max_num_residuals=INT_MAX))
    finally:
        ctx.close()
    for r in range(G):
        assert int(res[r]["default_armed"]) == 0
        assert int(res[r]["n"]) == ref.num_residuals and int(res[r]["pk"]) == ref.sum_candidates
        assert int(res[r]["armed"]) >= 8 and int(res[r]["fired"]) >= (6 if expire else 7), {k: int(res[r][k]) for k in ("armed", "fired", "cancelled")}
        for k in range(8):
            # (the repeated pass of the expiry case runs un-fused: another summation order, same value to 1e-12)
            assert rel(res[r]["HtH"][k], np.array(ref.HtH)) < 1e-12
            assert np.array_equal(res[r]["HtH"][k], res[0]["HtH"][k])
            if not (expire and k == 4):
                assert np.array_equal(res[r]["HtH"][k], res[r]["HtH"][0])


def test_the_stand_in_can_only_be_chosen_before_the_first_communicator_call():
    """one RCCL instance per process: once resolved (here: the process's real RCCL, through a 1-rank communicator id) it stays"""
    code = ("import sys; sys.path.insert(0, %r); import sr_livo_amd as srl\n"
            "srl.Context.comm_unique_id()\n"
            "try:\n    srl.comm_set_library(%r); print('ACCEPTED')\nexcept srl.SrlError: print('REFUSED')\n") % (ROOT, FAKE)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFUSED" in out.stdout, out.stdout + out.stderr
