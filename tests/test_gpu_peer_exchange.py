"""Direct peer exchange of the sharded sum (srl_peer_export / srl_peer_attach, include/srlivo_hip.h): every rank's finishing
workgroup stores its row into every rank's inbox and adds the rows it received in rank order -- no RCCL call, still one
kernel per pass.  What the sum must reproduce is the reference's single loop over ALL keypoints (optimize.cpp:68-110, the
sums of :235,239), ordered cut included (optimize.cpp:107 across ordered point-range shards).

A 1-GPU box offers two stand-ins for G GPUs: G contexts of one process (threads; the peers' inboxes are plain device pointers)
and G processes on the same device (the inboxes travel as HIP IPC handles -- the very path G GPUs of a node take).  The
in-process form stops at G = 2: a process owns 4 hardware queues by default (GPU_MAX_HW_QUEUES), each context uses two
streams, and a spinning exchange kernel that shares a hardware queue with the kernel it waits for can only time out.
Processes have their own queues, as ranks on G GPUs do: G = 2, 4, 8 run that way."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import sr_livo_amd as srl
from sr_livo_amd import capi

pytestmark = pytest.mark.gpu
INT_MAX = 2**31 - 1


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _single(golden, raw, max_res, passes=1):
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        ctx.sweep_upload(raw)
        f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
        for _ in range(passes):
            neq, _rc = ctx.build_residuals(f, srl.default_opts(max_num_residuals=max_res))
        return neq
    finally:
        ctx.close()


def _peer_threads(golden, raw, G, max_res, passes=3, fused=True, skip_rank=None, deadline_ms=None):
    """G contexts on device 0, one thread each; returns the per-rank results of the LAST pass (or the exception per rank)."""
    ctxs = [srl.Context(0) for _ in range(G)]
    ptrs = [c.peer_export()[1] for c in ctxs]
    out, errors = [None] * G, [None] * G
    start = threading.Barrier(G)

    def worker(rank):
        ctx = ctxs[rank]
        try:
            ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
            ctx.peer_attach(G, rank, local_ptrs=ptrs)
            if deadline_ms is not None:
                ctx.peer_set_deadline_ms(deadline_ms)
            ctx.set_fused_reduce(1 if fused else 0)
            ctx.sweep_upload(raw)
            f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
            start.wait()
            if rank == skip_rank:
                return
            for _ in range(passes):
                neq, _rc = ctx.build_residuals(f, srl.default_opts(max_num_residuals=max_res))
            out[rank] = (neq, ctx.sweep_shard())
        except Exception as e:  # noqa: BLE001
            errors[rank] = e
            try:
                start.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for c in ctxs:
        c.close()
    return out, errors


@pytest.mark.parametrize("G", [2])
@pytest.mark.parametrize("max_res", [INT_MAX, 600, 37, -1])
@pytest.mark.parametrize("fused", [True, False])
def test_peer_exchange_reproduces_the_single_context_pass(golden, G, max_res, fused):
    ref = _single(golden, golden["raw"], max_res)
    out, errors = _peer_threads(golden, golden["raw"], G, max_res, fused=fused)
    assert not any(errors), errors
    n_total = len(golden["raw"])
    for r in range(G):
        neq, (b, n, tot) = out[r]
        assert tot == n_total and b == n_total * r // G and n == n_total * (r + 1) // G - b        # contiguous point ranges
        assert neq.num_residuals == ref.num_residuals and neq.success == ref.success and neq.last_visited == ref.last_visited
        assert neq.nan_error == ref.nan_error
        if max_res == INT_MAX:
            assert neq.sum_candidates == ref.sum_candidates
        assert rel(np.array(neq.HtH), np.array(ref.HtH)) < 1e-12 and rel(np.array(neq.Hth), np.array(ref.Hth)) < 1e-12
        assert abs(neq.loss_sum - ref.loss_sum) <= 1e-12 * abs(ref.loss_sum)
        # rows are added in rank order on every rank: the SAME bits everywhere (the filters of all ranks stay identical)
        assert np.array_equal(np.array(neq.HtH), np.array(out[0][0].HtH)) and np.array_equal(np.array(neq.Hth), np.array(out[0][0].Hth))
        assert neq.loss_sum == out[0][0].loss_sum


def test_peer_exchange_many_passes_and_a_headline_sized_sweep():
    """64k keypoints over 2 logical ranks (32k each: the FUSED pass, exchange inside the association kernel), 40 passes back to
    back: slot parity and tags survive a long run (a rank may be one exchange ahead of a peer, never two)."""
    from sr_livo_amd import synth
    n_kp, map_pts, pattern, seed = synth.CONFIGS["HEADLINE"]
    cands, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    ref_ctx = srl.Context(0)
    ref_ctx.map_insert(cands)
    keys, counts, xyz = ref_ctx.map_download()
    ref_ctx.close()
    g = dict(map_keys=keys, map_counts=counts, map_xyz=xyz, q_pred=sw["q_pred"], t_pred=sw["t_pred"], t_last=sw["t_last"])
    ref = _single(g, sw["raw"], INT_MAX)
    out, errors = _peer_threads(g, sw["raw"], 2, INT_MAX, passes=40)
    assert not any(errors), errors
    for r in range(2):
        neq = out[r][0]
        assert neq.num_residuals == ref.num_residuals and neq.sum_candidates == ref.sum_candidates
        assert rel(np.array(neq.HtH), np.array(ref.HtH)) < 1e-12
        assert np.array_equal(np.array(neq.HtH), np.array(out[0][0].HtH))


def _big_raw(golden, seed, copies=4):
    """the golden sweep `copies` times over with a little noise: 4 096 keypoints per rank of two -- sixteen-wave workgroups, the fused
    pass that is armed (the golden sweep alone gives a rank 1 024 keypoints: four-wave workgroups, never armed)"""
    rng = np.random.default_rng(seed)
    return np.concatenate([golden["raw"] + rng.normal(0, 2e-3, golden["raw"].shape) for _ in range(copies)])


def _peer_threads_body(golden, G, body):
    """G contexts on device 0 (map uploaded, peers attached), one thread each running body(ctx, rank) -> result"""
    ctxs = [srl.Context(0) for _ in range(G)]
    ptrs = [c.peer_export()[1] for c in ctxs]
    out, errors = [None] * G, [None] * G
    start = threading.Barrier(G)

    def worker(rank):
        ctx = ctxs[rank]
        try:
            ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
            ctx.peer_attach(G, rank, local_ptrs=ptrs)
            start.wait()
            out[rank] = body(ctx, rank)
        except Exception as e:  # noqa: BLE001
            errors[rank] = e
            try:
                start.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for c in ctxs:
        c.close()
    return out, errors


def test_an_expired_armed_launch_repeats_its_pass_on_the_same_exchange(golden):
    """rank 0's armed launch gives up on the device (kernel-side bound 200 us) while its host sleeps; the host then fires
    it, learns that nobody was listening and launches the pass again -- with the tag of the SAME exchange: the launch that left never
    stored a row, and rank 1 is still polling for that one.  (One exchange ahead, both sides would spin until the deadline.)"""
    import time
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
    raw = _big_raw(golden, 5)
    ref = _single(golden, raw, INT_MAX)

    def body(ctx, rank):
        ctx.peer_set_deadline_ms(8000)
        ctx.set_armed_launch(2)                    # (two contexts of one process: the default policy would not arm)
        ctx.set_arm_linger(host_linger_us=3.6e9, kernel_linger_us=200.0)
        ctx.sweep_upload(raw)
        ctx.build_residuals(f, opts)
        s0 = ctx.arm_stats()
        assert s0["armed"] >= 1
        if rank == 0:
            time.sleep(0.05)                        # the launch armed by the pass above expires meanwhile
        t0 = time.perf_counter()
        neq, _ = ctx.build_residuals(f, opts)
        dt = time.perf_counter() - t0
        neq2, _ = ctx.build_residuals(f, opts)     # ... and the exchange after it is in step again
        s1 = ctx.arm_stats()
        ctx.disarm()
        return neq, neq2, s1["expired"] - s0["expired"], dt, ctx.peer_stats()

    out, errors = _peer_threads_body(golden, 2, body)
    assert not any(errors), errors
    assert out[0][2] == 1 and out[1][2] == 0, (out[0][2], out[1][2])
    assert out[0][3] < 2.0 and out[1][3] < 2.0, (out[0][3], out[1][3])          # nobody waited for a deadline
    for r in range(2):
        for neq in out[r][:2]:
            assert neq.num_residuals == ref.num_residuals and neq.sum_candidates == ref.sum_candidates
            assert rel(np.array(neq.HtH), np.array(ref.HtH)) < 1e-12
            assert np.array_equal(np.array(neq.HtH), np.array(out[0][0].HtH)) and np.array_equal(np.array(neq.Hth), np.array(out[0][0].Hth))
        assert out[r][4][1] == 0                                                # session alive


def test_a_sharded_stream_keeps_its_armed_launch_across_the_swap(golden):
    """in the sharded stream every srl_sweep_swap cancelled the waiting launch (999 of 1 000) -- the launch carried no
    alternate sweep buffer on a sharded context.  It does now: every rank prefetches and swaps its own point range, the launch armed
    behind the last pass of sweep k is fired with SRL_ARM_ALT as the first pass of sweep k + 1.  Sums equal the single-context pass on
    the same sweep, bit-identical across the ranks and to the same stream without armed launches."""
    sweeps = [_big_raw(golden, 77 + j) for j in range(3)]
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
    refs = [_single(golden, sw, INT_MAX) for sw in sweeps]
    rounds, passes = 9, 2

    def make_body(armed):
        def body(ctx, rank):
            pins = []
            for sw in sweeps:
                p = srl.PinnedArray(sw.shape); p.array[:] = sw; pins.append(p)
            ctx.set_armed_launch(2 if armed else 0)
            ctx.sweep_prefetch(pins[0].array); ctx.sweep_swap()
            res = []
            s0 = ctx.arm_stats()
            for k in range(rounds):
                ctx.sweep_prefetch(pins[(k + 1) % 3].array)
                for _ in range(passes):
                    neq, _ = ctx.build_residuals(f, opts)
                res.append(neq)
                ctx.solve_end()
                ctx.sweep_swap()
            s1 = ctx.arm_stats()
            ctx.disarm()
            for p in pins:
                p.close()
            return res, {k: s1[k] - s0[k] for k in s0}
        return body

    plain, errors = _peer_threads_body(golden, 2, make_body(False))
    assert not any(errors), errors
    got, errors = _peer_threads_body(golden, 2, make_body(True))
    assert not any(errors), errors
    for r in range(2):
        res, st = got[r]
        # every pass but the very first fires a waiting launch -- the first pass of every swapped-in sweep included
        assert st["fired"] >= rounds * passes - 1 and st["cancelled"] <= 1 and st["expired"] == 0, st
        for k, neq in enumerate(res):
            ref = refs[k % 3]
            assert neq.num_residuals == ref.num_residuals and neq.sum_candidates == ref.sum_candidates, (r, k)
            assert rel(np.array(neq.HtH), np.array(ref.HtH)) < 1e-12
            assert np.array_equal(np.array(neq.HtH), np.array(got[0][0][k].HtH))
            assert np.array_equal(np.array(neq.HtH), np.array(plain[r][0][k].HtH)) and np.array_equal(np.array(neq.Hth), np.array(plain[r][0][k].Hth))


def test_ranks_that_share_a_device_do_not_arm_by_default(golden):
    """srl_peer_export leaves the device's identity behind the inbox rows, srl_peer_attach compares: peers on ONE device (this test
    arrangement; production is one process per GPU) must not arm launches under the default policy -- a waiting launch holds a workgroup
    per compute unit that the other rank's kernel needs."""
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])

    def body(ctx, rank):
        ctx.sweep_upload(raw)
        ctx.set_armed_launch(1)
        for _ in range(4):
            ctx.build_residuals(f, opts)
        s1 = ctx.arm_stats()
        ctx.set_armed_launch(2)
        for _ in range(4):
            ctx.build_residuals(f, opts)
        s2 = ctx.arm_stats()
        ctx.disarm()
        return s1, s2

    raw = _big_raw(golden, 9)
    out, errors = _peer_threads_body(golden, 2, body)
    assert not any(errors), errors
    for r in range(2):
        assert out[r][0]["armed"] == 0, out[r]
        assert out[r][1]["armed"] >= 4, out[r]                              # (the passes ARE eligible: forced, they arm -- once more for a pass repeated for a late row)


def test_a_second_session_does_not_see_the_rows_of_the_first(golden):
    """Detach, export again (the export resets the inbox), attach again with the ranks SWAPPED and another sweep: exchange tags
    start at 1 again, so a stale row of the first session would be taken for a fresh one if the reset were missing."""
    ctxs = [srl.Context(0) for _ in range(2)]
    f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    try:
        for c in ctxs:
            c.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        results = []
        for session, (raw, order) in enumerate(((golden["raw"], (0, 1)), (golden["raw"][: len(golden["raw"]) // 2], (1, 0)))):
            ptrs = [c.peer_export()[1] for c in ctxs]                      # every rank exports (= resets) before anybody attaches
            out = [None, None]

            def worker(i):
                rank = order[i]
                c = ctxs[i]
                c.peer_attach(2, rank, local_ptrs=[ptrs[order.index(0)], ptrs[order.index(1)]])
                c.sweep_upload(raw)
                for _ in range(3 + session):
                    out[i] = c.build_residuals(f, opts)[0]

            ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            ref = _single(golden, raw, INT_MAX)
            for i in range(2):
                assert out[i] is not None and out[i].num_residuals == ref.num_residuals
                assert rel(np.array(out[i].HtH), np.array(ref.HtH)) < 1e-12
            assert np.array_equal(np.array(out[0].HtH), np.array(out[1].HtH))
            results.append(np.array(out[0].HtH))
            with pytest.raises(srl.SrlError):
                ctxs[0].peer_export()                                      # not while peers are attached
            for c in ctxs:
                c.peer_detach()
        assert not np.array_equal(results[0], results[1])
    finally:
        for c in ctxs:
            c.close()


def test_a_missing_peer_is_an_error_not_a_hang(golden):
    """Rank 1 of 2 never calls: rank 0 re-polls until its deadline (srl_peer_set_deadline_ms) and srl_build_residuals returns SRL_ERR_COMM."""
    out, errors = _peer_threads(golden, golden["raw"], 2, INT_MAX, passes=1, skip_rank=1, deadline_ms=1200)
    assert out[0] is None and isinstance(errors[0], srl.SrlError) and errors[0].status == capi.SRL_ERR_COMM, (out, errors)
    assert "peer" in str(errors[0])


def test_a_failed_session_stays_failed_until_it_is_started_again(golden):
    """After a row never arrived the rank must not run ahead into the slot its peers may still be reading: every further pass of the
    session fails at once (no second bounded spin), and export + attach on every rank starts a working session on the same contexts."""
    import time
    ctxs = [srl.Context(0) for _ in range(2)]
    f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    try:
        for c in ctxs:
            c.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        ptrs = [c.peer_export()[1] for c in ctxs]
        for r, c in enumerate(ctxs):
            c.peer_attach(2, r, local_ptrs=ptrs)
            c.peer_set_deadline_ms(500)
            c.sweep_upload(golden["raw"])
        with pytest.raises(srl.SrlError) as e1:
            ctxs[0].build_residuals(f, opts)                               # rank 1 never calls
        assert e1.value.status == capi.SRL_ERR_COMM
        t0 = time.perf_counter()
        with pytest.raises(srl.SrlError) as e2:
            ctxs[0].build_residuals(f, opts)
        assert e2.value.status == capi.SRL_ERR_COMM and "session" in str(e2.value) and time.perf_counter() - t0 < 0.2
        for c in ctxs:
            c.peer_detach()
        ptrs = [c.peer_export()[1] for c in ctxs]
        out = [None, None]

        def worker(r):
            ctxs[r].peer_attach(2, r, local_ptrs=ptrs)
            ctxs[r].sweep_upload(golden["raw"])
            out[r] = ctxs[r].build_residuals(f, opts)[0]

        ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        ref = _single(golden, golden["raw"], INT_MAX)
        assert out[0] is not None and out[1] is not None and out[0].num_residuals == ref.num_residuals == out[1].num_residuals
        assert np.array_equal(np.array(out[0].HtH), np.array(out[1].HtH))
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("fused", [True, False])
def test_a_late_rank_is_not_a_failure(golden, fused):
    """Rank 1 starts its pass 2.5 s late -- longer than the bounded spin of the kernel that waits for its row.  Rank 0 must not fail and
    must not run ahead: it repeats the pass with the same exchange tags (srl_peer_stats counts the repeats) until the row is there; both
    ranks return SRL_OK with the single-context result, and the passes behind it run in lock step as if nothing had happened."""
    import time
    ctxs = [srl.Context(0) for _ in range(2)]
    f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    out, errors = [None, None], [None, None]
    try:
        ptrs = [c.peer_export()[1] for c in ctxs]
        for r, c in enumerate(ctxs):
            c.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
            c.peer_attach(2, r, local_ptrs=ptrs)
            c.peer_set_deadline_ms(30_000)
            c.set_fused_reduce(1 if fused else 0)
            c.sweep_upload(golden["raw"])

        def worker(r):
            try:
                if r == 1:
                    time.sleep(2.5)
                for _ in range(4):
                    out[r] = ctxs[r].build_residuals(f, opts)[0]
            except Exception as e:  # noqa: BLE001
                errors[r] = e

        ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert errors == [None, None], errors
        ref = _single(golden, golden["raw"], INT_MAX)
        for r in range(2):
            assert out[r].num_residuals == ref.num_residuals and rel(np.array(out[r].HtH), np.array(ref.HtH)) < 1e-12
        assert np.array_equal(np.array(out[0].HtH), np.array(out[1].HtH)) and np.array_equal(np.array(out[0].Hth), np.array(out[1].Hth))
        rep0, failed0 = ctxs[0].peer_stats()
        rep1, failed1 = ctxs[1].peer_stats()
        assert rep0 >= 1 and not failed0 and not failed1, (rep0, rep1)
    finally:
        for c in ctxs:
            c.close()


def test_a_rank_that_gives_up_poisons_the_session_for_the_others(golden):
    """Rank 0 (deadline 0.6 s) gives the session up while rank 1 (deadline 60 s) is away.  Rank 1's first pass still completes -- rank 0's
    rows of that exchange had been delivered --, its second finds no row, and instead of re-polling for a minute it sees the poison word
    rank 0 left in its inbox and returns SRL_ERR_COMM within the kernel's bounded spin."""
    import time
    ctxs = [srl.Context(0) for _ in range(2)]
    f = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
    opts = srl.default_opts(max_num_residuals=INT_MAX)
    try:
        ptrs = [c.peer_export()[1] for c in ctxs]
        for r, c in enumerate(ctxs):
            c.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
            c.peer_attach(2, r, local_ptrs=ptrs)
            c.peer_set_deadline_ms(600 if r == 0 else 60_000)
            c.sweep_upload(golden["raw"])
        with pytest.raises(srl.SrlError) as e0:
            ctxs[0].build_residuals(f, opts)
        assert e0.value.status == capi.SRL_ERR_COMM and "deadline" in str(e0.value)
        assert ctxs[0].peer_stats()[1]
        neq, _ = ctxs[1].build_residuals(f, opts)                          # exchange 1: rank 0's rows are there
        assert neq.num_residuals > 0
        t0 = time.perf_counter()
        with pytest.raises(srl.SrlError) as e1:
            ctxs[1].build_residuals(f, opts)                               # exchange 2: nobody is coming
        dt = time.perf_counter() - t0
        assert e1.value.status == capi.SRL_ERR_COMM and "another rank" in str(e1.value), str(e1.value)
        assert dt < 5.0, dt
        assert ctxs[1].peer_stats()[1]
    finally:
        for c in ctxs:
            c.close()


def test_peer_attach_refuses_a_second_transport(golden):
    ctx = srl.Context(0)
    other = srl.Context(0)
    try:
        ptrs = [ctx.peer_export()[1], other.peer_export()[1]]
        ctx.comm_set_host_callbacks(2, 0, lambda a: None, lambda m: [m, 0])
        with pytest.raises(srl.SrlError):
            ctx.peer_attach(2, 0, local_ptrs=ptrs)
        ctx.comm_set_host_callbacks(1, 0, lambda a: None, lambda m: [m])
        ctx.comm_destroy()
        ctx.peer_attach(2, 0, local_ptrs=ptrs)
        with pytest.raises(srl.SrlError):
            ctx.comm_set_host_callbacks(2, 0, lambda a: None, lambda m: [m, 0])
        ctx.peer_detach()
        with pytest.raises(srl.SrlError):
            ctx.peer_attach(9, 0, local_ptrs=(ptrs * 5)[:9])          # one node: at most 8 ranks
    finally:
        ctx.close()
        other.close()


_IPC_SCRIPT = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import sr_livo_amd as srl
from sr_livo_amd import capi
rank, G, d, max_res, scene, fused = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), sys.argv[6], int(sys.argv[7])
ctx = srl.Context(0)
if scene == "golden":
    g = np.load(os.path.join(sys.argv[1], "tests", "golden", "golden_small.npz"))
    ctx.map_upload(g["map_keys"], g["map_counts"], g["map_xyz"])
else:
    from sr_livo_amd import synth
    n_kp, map_pts, pattern, seed = synth.CONFIGS[scene]
    cands, L = synth.map_candidates(seed, map_pts)
    g = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    ctx.map_insert(cands)
ctx.set_fused_reduce(fused)
handle, _ptr = ctx.peer_export()
open(os.path.join(d, f"h{rank}.tmp"), "wb").write(handle); os.replace(os.path.join(d, f"h{rank}.tmp"), os.path.join(d, f"h{rank}.bin"))
handles = []
for r in range(G):
    p = os.path.join(d, f"h{r}.bin"); t0 = time.time()
    while not os.path.exists(p):
        if time.time() - t0 > 120: raise SystemExit(3)
        time.sleep(0.02)
    handles.append(open(p, "rb").read())
ctx.peer_attach(G, rank, handles=handles)
ctx.sweep_upload(g["raw"])
f = capi.make_frame(g["q_pred"], g["t_pred"], g["t_last"])
for _ in range(5):
    neq, rc = ctx.build_residuals(f, srl.default_opts(max_num_residuals=max_res))
np.savez(os.path.join(d, f"out{rank}.npz"), HtH=np.array(neq.HtH), Hth=np.array(neq.Hth), loss=neq.loss_sum, n=neq.num_residuals, last=neq.last_visited,
         pk=neq.sum_candidates, armed=ctx.arm_stats()["armed"])
# nobody unmaps an inbox a peer may still be storing into
open(os.path.join(d, f"done{rank}"), "w").close()
t0 = time.time()
while not all(os.path.exists(os.path.join(d, f"done{r}")) for r in range(G)) and time.time() - t0 < 60:
    time.sleep(0.02)
ctx.close()
print("OK", rank, flush=True)
"""


def _peer_processes(tmp_path, G, max_res, scene, fused):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", _IPC_SCRIPT, root, str(r), str(G), str(tmp_path), str(max_res), scene, str(int(fused))],
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


@pytest.mark.parametrize("G,max_res,fused", [(2, INT_MAX, 1), (2, 600, 1), (4, INT_MAX, 1), (4, 600, 0), (4, -1, 1), (8, INT_MAX, 0), (8, 37, 1)])
def test_peer_exchange_between_processes_through_hip_ipc(golden, tmp_path, G, max_res, fused):
    """G PROCESSES on the one device: the inboxes travel as HIP IPC handles (what G GPUs of one node do)."""
    res = _peer_processes(tmp_path, G, max_res, "golden", fused)
    ref = _single(golden, golden["raw"], max_res)
    for r in range(G):
        assert int(res[r]["n"]) == ref.num_residuals and int(res[r]["last"]) == ref.last_visited
        assert rel(res[r]["HtH"], np.array(ref.HtH)) < 1e-12 and rel(res[r]["Hth"], np.array(ref.Hth)) < 1e-12
        assert np.array_equal(res[r]["HtH"], res[0]["HtH"]) and float(res[r]["loss"]) == float(res[0]["loss"])


def test_peer_exchange_four_processes_headline_sweep_fused(tmp_path):
    """64k keypoints over 4 processes (16k each): every rank runs the FUSED pass -- the exchange happens inside the association
    kernel's finishing workgroup, one kernel per pass and rank."""
    from sr_livo_amd import synth
    n_kp, map_pts, pattern, seed = synth.CONFIGS["HEADLINE"]
    cands, L = synth.map_candidates(seed, map_pts)
    sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
    ctx = srl.Context(0)
    try:
        ctx.map_insert(cands)
        ctx.sweep_upload(sw["raw"])
        ref, _ = ctx.build_residuals(capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"]), srl.default_opts(max_num_residuals=INT_MAX))
    finally:
        ctx.close()
    res = _peer_processes(tmp_path, 4, INT_MAX, "HEADLINE", 1)
    for r in range(4):
        assert int(res[r]["n"]) == ref.num_residuals and int(res[r]["pk"]) == ref.sum_candidates
        assert rel(res[r]["HtH"], np.array(ref.HtH)) < 1e-12 and rel(res[r]["Hth"], np.array(ref.Hth)) < 1e-12
        assert np.array_equal(res[r]["HtH"], res[0]["HtH"])
        # one device under four rank processes: srl_peer_attach saw the peers' inboxes on its own device -- nobody armed a launch (default
        # policy; on one process per GPU every one of these fused passes arms its successor)
        assert int(res[r]["armed"]) == 0
