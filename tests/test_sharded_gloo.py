"""world_size-2 gloo test (CPU): the sharded path of north_star -- point-range shards, ordered residual
budget across shards (optimize.cpp:107), all-reduce of the 6x6 normal equations, replicated host ESIKF
update -- gives the single-process result.  The per-shard kernel work is stood in for by the oracle (no
GPU here); partition, budget and the host update are the product's own code (srl_shard_range,
srl_shard_budget, lioOptimization::updateIEKF via srl_lio_update_iekf_provided)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, max_res, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sr_livo_amd as srl
    from oracle import pyoracle as po
    from sr_livo_amd import synth

    pts, L = synth.map_candidates(777, 30_000)
    m = po.Map(); m.add_points(pts)
    sw = synth.make_sweep(778, 1500, L)
    raw = sw["raw"]
    b, c = srl.shard_range(len(raw), world, rank)
    shard = raw[b:b + c]
    opts_p = srl.default_opts(max_num_residuals=max_res)
    o_all = po.opts_from_product(opts_p); o_all.max_num_residuals = 2**31 - 1

    def provider(frame, opts, out):
        q, t, tl = np.array(frame.q), np.array(frame.t), np.array(frame.t_last)
        # 1. every shard evaluates all of its keypoints (what the kernel does)
        full = m.build_plane_residuals(o_all, shard, q, t, tl, frame_id=frame.frame_id)
        accepted = int((full["status"] == 2).sum())
        # 2. all-gather of the accepted counts -> this shard's residual budget (product code)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([accepted], dtype=torch.int64))
        budget, mode = srl.shard_budget(opts.max_num_residuals, [int(x) for x in counts], rank)
        # 3. shard partial with the ordered cut applied
        J = full["jacobian"]; d = full["distance"]; w = full["weight"]
        acc_idx = np.where(full["status"] == 2)[0]
        if mode == 2:
            acc_idx = acc_idx[:0]
        elif mode == 1:
            acc_idx = acc_idx[acc_idx == 0]
        else:
            acc_idx = acc_idx[: max(budget, 0)]
        H = J[acc_idx]; h = d[acc_idx] * w[acc_idx]
        buf = torch.zeros(44, dtype=torch.float64)
        buf[:36] = torch.from_numpy((H.T @ H).ravel()); buf[36:42] = torch.from_numpy(H.T @ h)
        buf[42] = float((d[acc_idx] ** 2).sum()); buf[43] = float(len(acc_idx))
        # 4. the one exchange step
        dist.all_reduce(buf)
        out.HtH[:] = buf[:36].tolist(); out.Hth[:] = buf[36:42].tolist()
        out.loss_sum = float(buf[42]); out.num_residuals = int(buf[43]); out.success = int(buf[43] >= opts.min_number_neighbors)
        return 0

    lio = srl.Lio(-1)
    e = po.Eskf(); synth.eskf_prior(e, sw["q_pred"], sw["t_pred"], sw["vel"])
    lio.eskf_set_state(e.get_state()); lio.eskf_set_cov(e.get_cov())
    st = np.concatenate([sw["q_pred"], sw["t_pred"], sw["vel"], np.zeros(6)])
    g = lio.update_iekf_provided(opts_p, provider, len(raw), st, sw["t_last"], log_iters=10)
    # single-process reference on rank 0
    if rank == 0:
        u = po.update_iekf(m, e, po.opts_from_product(opts_p), raw, st, sw["t_last"], log_iters=10)
        ret["ref_state"] = u["state"]; ret["ref_iters"] = u["rc"]; ret["ref_nres"] = u["num_residuals"]; ret["ref_cov"] = e.get_cov()
    ret[f"state{rank}"] = g["state"]; ret[f"iters{rank}"] = g["iters"]; ret[f"nres{rank}"] = g["num_residuals"]
    ret[f"cov{rank}"] = lio.eskf_get_cov(); ret[f"rc{rank}"] = g["rc"]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("max_res", [2**31 - 1, 600, 100])
def test_two_rank_sharded_solve_equals_single_process(max_res):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), max_res, ret), nprocs=world, join=True)
    assert ret["rc0"] == 0 and ret["rc1"] == 0
    assert ret["iters0"] == ret["iters1"] == ret["ref_iters"]
    assert ret["nres0"] == ret["nres1"] == ret["ref_nres"]
    # replicated host update: bitwise identical on both ranks (same all-reduced inputs, same code)
    assert np.array_equal(ret["state0"], ret["state1"]) and np.array_equal(ret["cov0"], ret["cov1"])
    # vs single process: only the summation order of the normal equations differs
    err = np.max(np.abs(ret["state0"] - ret["ref_state"])) / np.max(np.abs(ret["ref_state"]))
    assert err < 1e-11
    assert np.max(np.abs(ret["cov0"] - ret["ref_cov"])) / np.max(np.abs(ret["ref_cov"])) < 1e-9
