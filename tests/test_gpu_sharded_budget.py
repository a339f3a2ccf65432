"""The sharded ordered cut (optimize.cpp:107 across ordered point-range shards): every rank's residual budget is what the ranks
before it left.  With a communicator the per-rank counts are all-gathered ON THE STREAM and srl_reduce_kernel derives budget
and mode itself (`prior += gather[r]` for r < rank; no host synchronisation in the loop).  Those rank > 0 branches cannot be
reached with the 1-rank communicator a single-GPU box allows, so srl_debug_set_gather_counts preloads the gathered counts and
lets the context act as any rank: the result must be bit-identical to the host-side srl_shard_budget path (host callbacks),
for every budget regime -- budget left, budget exactly spent, budget overspent, and max_num_residuals <= 0 (stop at the first
keypoint with a plane, wherever it is)."""
import numpy as np
import pytest

import sr_livo_amd as srl
from sr_livo_amd import capi

pytestmark = pytest.mark.gpu


def _neq_tuple(neq):
    return (neq.num_residuals, neq.last_visited, neq.nan_error, bytes(bytearray(np.array(neq.HtH).tobytes())), bytes(bytearray(np.array(neq.Hth).tobytes())), neq.loss_sum)


@pytest.mark.parametrize("max_res", [600, 37, -1])
def test_reduce_kernel_derives_the_rank_budget_on_the_device(golden, max_res):
    ctx = srl.Context(0)
    try:
        ctx.map_upload(golden["map_keys"], golden["map_counts"], golden["map_xyz"])
        frame = capi.make_frame(golden["q_pred"], golden["t_pred"], golden["t_last"])
        opts = srl.default_opts(max_num_residuals=max_res)
        seen_modes = set()
        for nranks, rank in [(2, 1), (4, 1), (4, 3), (8, 5)]:
            if max_res > 0:
                scenarios = [[0] * rank, [max_res // (2 * rank)] * rank, [max_res // rank + 1] * rank,
                             [max_res] + [0] * (rank - 1), [max_res - 1] + [0] * (rank - 1), [0] * (rank - 1) + [max_res - 5]]
            else:
                scenarios = [[0] * rank, [0] * (rank - 1) + [3], [7] + [0] * (rank - 1)]
            for prior in scenarios:
                counts = list(prior) + [0] * (nranks - rank)                 # entries of this rank and later ones are never read
                # (A) on-device derivation from the preloaded gather
                ctx.set_gather_counts(nranks, rank, counts)
                ctx.sweep_upload(golden["raw"])
                b, n, tot = ctx.sweep_shard()
                assert tot == len(golden["raw"]) and n == len(golden["raw"]) * (rank + 1) // nranks - b
                neq_a, rc_a = ctx.build_residuals(frame, opts)
                ctx.set_gather_counts(counts=None)
                # (B) host side: srl_shard_budget on the gathered counts
                mine = {}
                def gather(m):
                    mine["n"] = m
                    return counts[:rank] + [m] + counts[rank + 1:]
                ctx.comm_set_host_callbacks(nranks, rank, lambda a: None, gather)
                ctx.sweep_upload(golden["raw"])
                neq_b, rc_b = ctx.build_residuals(frame, opts)
                ctx.comm_set_host_callbacks(1, 0, lambda a: None, lambda m: [m])
                assert rc_a == rc_b == 0
                assert _neq_tuple(neq_a) == _neq_tuple(neq_b), (max_res, nranks, rank, prior)
                budget, mode = srl.shard_budget(max_res, counts[:rank] + [mine["n"]] + counts[rank + 1:], rank)
                seen_modes.add(mode)
                if mode == 2:
                    assert neq_a.num_residuals == 0 and neq_a.last_visited == -1
                elif max_res > 0:
                    assert neq_a.num_residuals == min(budget, mine["n"])
                else:
                    assert neq_a.num_residuals <= 1
        assert seen_modes == ({0, 2} if max_res > 0 else {1, 2})
    finally:
        ctx.close()
