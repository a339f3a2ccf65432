"""Everything bench.py needs BEFORE its timed region: ranks, device, map, transport, the stream of sweeps -- and the few helpers the
legs behind the region share (barrier, max over ranks, the resident re-solve).  No timing decisions live here."""
import gc
import os
import time

import numpy as np
import torch

import sr_livo_amd as srl
from sr_livo_amd import synth

from .launcher import c_stdout_to_stderr, pin_to_gpu_numa_node
from .stream import Streamer, _EskfAdapter, make_stream


def last_error(lio):
    lib = lio.lib
    return (lib.srl_lio_last_error(lio.h) or b"").decode(errors="replace") or (lib.srl_last_error(lio.ctx.h) or b"").decode(errors="replace")


class Run:
    """one rank of `python bench.py --gpus N`: its context, map, transport and stream"""

    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("SRL_BENCH_ALL_ON_DEVICE0") == "1":
            # test hook (1-GPU boxes): every rank on device 0 -- lets the N > 1 path run end to end (the inboxes travel as HIP IPC handles
            # exactly as between GPUs).  Never a performance figure.
            self.local_rank = 0
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product has no CPU path")
        try:
            os.nice(-10)              # a 20-step region lasts 2 ms: one pre-emption of the polling thread (80 us) is 4 % of it.  Best effort.
        except OSError:
            pass
        torch.cuda.set_device(self.local_rank)
        torch.cuda.synchronize()      # torch's lazy initialisation happens HERE, not inside the barrier in front of the timed region
        self.pin_info = {"pinned": False, "disabled": True} if args.no_numa_pin else pin_to_gpu_numa_node(self.local_rank)
        self.dist = None
        if self.world > 1 or "RANK" in os.environ:
            import datetime
            import torch.distributed as dist_mod
            self.dist = dist_mod
            # control plane only (barrier, id broadcast, one max-reduce): gloo over loopback.  No torch NCCL process group is created --
            # the only communicator on the GPUs is the library's own, on the process's one RCCL instance.
            with c_stdout_to_stderr():          # gloo announces its connections on stdout
                self.dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
        world, rank = self.world, self.rank
        self.n_kp, self.map_pts, self.pattern, seed = synth.CONFIGS[args.workload]
        self.sharded = (world > 1 or (args.force_comm and "RANK" in os.environ)) and args.mode == "sharded"
        own = rank if (world > 1 and not self.sharded) else 0          # replicas: every rank its own scene
        self.sweep_seed, map_seed = seed + 1000 + own, seed + own

        # ---------------- inputs: map built by the product's device-side addPointsToMap, sweeps in page-locked host memory
        t0 = time.time()
        self.cands, self.L = synth.map_candidates(map_seed, self.map_pts)
        # --workload SPREAD: the off-cache aux workload (keypoints area-uniform over the whole scene, random order; profiling only)
        self.gen = (lambda sd: synth.make_spread_sweep(sd, self.n_kp, self.cands, self.L)) if args.workload == "SPREAD" else None
        self.sweep = self.gen(self.sweep_seed) if self.gen else synth.make_sweep(self.sweep_seed, self.n_kp, self.L, pattern=self.pattern)
        self.lio = lio = srl.Lio(self.local_rank)
        if args.no_fused_reduce:
            lio.ctx.set_fused_reduce(0)
        lio.add_points_to_map(self.cands)
        self.n_map = lio.map_size()
        self.comm_info, self.transport = None, None
        if self.sharded:
            self.attach(args.transport)
        elif args.force_comm and self.dist is not None:
            self.attach("rccl")
        sweep = self.sweep
        self.prior_state = synth.eskf_prior(_EskfAdapter(lio), sweep["q_pred"], sweep["t_pred"], sweep["vel"]).copy()
        self.prior_cov = lio.eskf_get_cov().copy()
        self.state0 = np.concatenate([sweep["q_pred"], sweep["t_pred"], sweep["vel"], np.zeros(6)])
        self.opts = srl.default_opts(max_num_residuals=args.max_num_residuals, select_mode=args.select_mode)
        # THE STREAM (one full solve per sweep, "incl. H2D of the sweep"; src/lioOptimization.cpp:1003-1027 never solves a
        # sweep twice): S distinct sweeps of the scene -- own seeds, own poses, hence own priors.  Sweep 0 is the sweep every other leg
        # (CPU baselines, parity, profiles) uses.
        self.stream = make_stream(sweep, self.prior_state, self.sweep_seed, self.n_kp, self.L, self.pattern, max(int(args.stream_sweeps), 1), self.iterations_of, self.gen)
        self.S = len(self.stream)
        lio.resident_sweep(sweep["raw"])
        self.setup_s = time.time() - t0
        # The interpreter's cyclic collector is host noise, not part of the path: with torch imported one full collection costs ~40 ms
        # (measured: exactly one 38-43 ms step per run, gone without torch in the process).  Everything allocated so far moves out of the
        # collector's reach; the collector itself stays on.
        gc.collect()
        gc.freeze()
        # one resident solve = eskf_set_state + eskf_set_cov (reset the prior) + update_iekf, through a closure that converts its arguments once
        self._solve = lio.bound_solver(self.opts, self.prior_state, self.prior_cov, self.state0, sweep["t_last"], args.frame_id, self.n_kp)
        self.streamer = Streamer(lio, self.stream, self.opts, self.prior_cov, args.frame_id, self.n_kp)
        if args.no_armed:
            lio.ctx.set_armed_launch(False)

    # ---- transports of the sharded sum
    def attach(self, transport):
        lio, dist, world, rank = self.lio, self.dist, self.world, self.rank
        if transport == "peer":
            handles = [None] * world
            dist.all_gather_object(handles, lio.ctx.peer_export()[0])
            lio.ctx.peer_attach(world, rank, handles=handles)
            self.comm_info = {"transport": "direct peer exchange (srl_peer_attach): rows stored into the peers' inboxes, summed in rank order inside the "
                                           "association kernel's finishing workgroup; no RCCL call on the data path"}
        else:
            stand_in = os.environ.get("SRL_BENCH_RCCL_LIBRARY")           # test hook (1-GPU boxes): librccl refuses two ranks on one device
            if stand_in and not getattr(Run, "_stand_in_chosen", False):
                srl.comm_set_library(stand_in)                            # (once per process, before its first communicator call)
                Run._stand_in_chosen = True
            uid = [srl.Context.comm_unique_id() if rank == 0 else None]
            if dist is not None:
                dist.broadcast_object_list(uid, src=0)
            with c_stdout_to_stderr():
                lio.ctx.comm_init_rank(world, rank, uid[0])
            origin, ver, pre = srl.comm_backend_info()
            self.comm_info = {"rccl": origin, "version": ver, "instance": "already loaded in the process" if pre else "dlopen'ed by libsrlivo_hip.so"}
        self.transport = transport

    def detach(self):
        self.lio.ctx.disarm()
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()          # nobody unmaps an inbox / destroys a communicator a peer may still be using
        if self.transport == "peer":
            self.lio.ctx.peer_detach()
        elif self.transport is not None:
            self.lio.ctx.comm_destroy()
        self.transport = None

    def iterations_of(self, e):
        self.lio.resident_sweep(e["sweep"]["raw"])
        rc, it, _ = self.lio.bound_solver(self.opts, e["prior_state"], self.prior_cov, e["state0"], e["sweep"]["t_last"], self.args.frame_id, self.n_kp)()
        return it if rc == 0 else -1

    def solve(self):
        rc, it, nr = self._solve()
        if rc:
            raise SystemExit(f"update_iekf failed with status {rc}: {last_error(self.lio)}")
        return {"iters": it, "num_residuals": nr, "state": self._solve.state}

    def barrier(self):
        self.lio.ctx.disarm()     # (the launch the last pass armed would hold a device-wide synchronisation until it leaves by itself)
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        te = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(te, op=self.dist.ReduceOp.MAX)
        return float(te.item())

    def clock_warmup(self):
        """setup, untimed: bring GPU and host core to their steady clocks (a timed region of 20 solves lasts 2 ms: measured right after the map
        build, its steps kept getting faster until the end -- 107 -> 103 us per solve); the W warm-up steps of the contract follow"""
        ms = self.args.clock_warmup_ms
        n = 0
        self.streamer.begin()
        if self.dist is None:
            t = time.perf_counter()
            while time.perf_counter() - t < ms * 1e-3:
                self.streamer.step(); n += 1
        else:
            for _ in range(int(ms * 4)):      # ranks solve in lock-step (the exchange is collective): a count, not a clock
                self.streamer.step(); n += 1
        return n

    def close(self):
        try:
            self.lio.ctx.disarm(); torch.cuda.synchronize()
            if self.transport is not None:
                self.detach()
        except Exception:  # noqa: BLE001
            pass
        self.streamer.close()
        self.lio.close()
        self.lio = None
        self.cands = None
