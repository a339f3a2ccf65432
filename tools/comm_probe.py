"""What the sharded code path costs per srl_build_residuals call with ONE rank (all a 1-GPU box can run): wall-clock us per call,
headline sweep.  default | 1-rank RCCL communicator forced, fused pass (device mailbox -> ncclAllReduce -> publish kernel) |
the same with the reduce kernel (round 2's form)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_livo_amd as srl
from sr_livo_amd import capi, synth

n_kp, map_pts, pattern, seed = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "HEADLINE"]
cands, L = synth.map_candidates(seed, map_pts)
sw = synth.make_sweep(seed + 1000, n_kp, L, pattern=pattern)
ctx = srl.Context(0)
ctx.pin_thread_to_gpu_numa()
ctx.map_insert(cands)
opts = srl.default_opts(max_num_residuals=2**31 - 1)
f = capi.make_frame(sw["q_pred"], sw["t_pred"], sw["t_last"])


def run(label, reps=300):
    ctx.sweep_upload(sw["raw"])
    for _ in range(20):
        ctx.build_residuals(f, opts)
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            out, _rc = ctx.build_residuals(f, opts)
        best.append((time.perf_counter() - t0) / reps * 1e6)
    print(f"{label:58s} {min(best):7.2f} us per call  (residuals {out.num_residuals})", flush=True)


run("default (single rank, fused, host mailbox)")
os.environ["SRL_FORCE_COLLECTIVES"] = "1"
ctx.comm_init_rank(1, 0, srl.Context.comm_unique_id())
run("1-rank communicator, fused pass + all-reduce + publish")
ctx.set_fused_reduce(0)
run("1-rank communicator, reduce kernel + all-reduce + publish")
ctx.set_fused_reduce(1)
ctx.comm_destroy()
run("default again")
ctx.set_fused_reduce(0)
run("single rank, reduce kernel (no communicator)")
