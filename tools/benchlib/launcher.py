"""Process plumbing of bench.py: `--gpus N` without a launcher around it, NUMA placement, C-level stdout chatter."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SELF_LAUNCH_TIMEOUT_S = 900

def pin_to_gpu_numa_node(device_index):
    """Run this process on the CPUs of the NUMA node the GPU hangs off (2-socket hosts: the mailbox read and the doorbell
    write of every ESIKF iteration otherwise cross the socket interconnect -- measured +3 us per iteration, tools/numa_probe.py)
    through the library's own helper, srl_thread_pin_to_gpu_numa (the main thread is pinned before any other thread exists, so
    the whole process follows).  Standard placement for a latency-bound host loop; INTEGRATION.md says the same for the node.
    Never fatal."""
    info = {"pinned": False}
    try:
        import sr_livo_amd as srl
        ctx = srl.Context(device_index)
        try:
            node = ctx.pin_thread_to_gpu_numa()
        finally:
            ctx.close()
        if node is not None:
            info.update(pinned=True, numa_node=int(node), cpus=len(os.sched_getaffinity(0)))
    except Exception as e:  # noqa: BLE001
        info["error"] = repr(e)
    return info


class c_stdout_to_stderr:
    """RCCL prints a version banner through C stdio on communicator creation; the bench contract is ONE JSON line on stdout.
    Inside this block file descriptor 1 points at stderr, and the C buffers are flushed before it is pointed back."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self.libc = ctypes.CDLL(None)
        self.libc.fflush(None)
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        self.libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False



def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher around it: re-run this command line as N ranks under torch.distributed.run
    (one process per GPU, rendezvous over loopback) and pass rank 0's line through.  The sharded path has never met a real N-GPU node
    (the build boxes have one GPU): should the chosen exchange fail or hang there, the run falls back -- RCCL all-reduce -> direct peer
    exchange -> independent replicas (BASELINE config 5, no collective) -- and says so in the line (`fallback`)."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    base = [a for a in sys.argv[1:]]
    explicit = any(a in ("--transport", "--mode") or a.startswith("--transport=") or a.startswith("--mode=") for a in base)
    # (a fallback attempt does not touch the transport that has just failed or hung again: --no-other-transport)
    attempts = [([], None)] if explicit else [([], None), (["--no-other-transport", "--transport", "peer"], "RCCL form failed or hung: direct peer exchange"),
                                              (["--mode", "replay"], "sharded forms failed or hung: independent replicas, one sweep per GPU")]
    last_rc = 1
    for extra, note in attempts:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + base + extra
        # the launcher and its ranks form a process group of their own: an attempt that hangs is ended as a whole (killing only the launcher
        # would leave its ranks spinning on the GPUs under the next attempt)
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=SELF_LAUNCH_TIMEOUT_S)
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(p.pid, signal.SIGKILL)               # exactly the group started above
            except ProcessLookupError:
                pass
            p.communicate()
            print(f"bench.py: {' '.join(extra) or 'default transport'} timed out after {SELF_LAUNCH_TIMEOUT_S} s", file=sys.stderr)
            continue
        last_rc = p.returncode
        lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{")]
        if p.returncode == 0 and lines:
            line = lines[-1]
            if note:
                try:
                    d = json.loads(line)
                    d["fallback"] = note
                    line = json.dumps(d, allow_nan=False, separators=(",", ":"))
                except ValueError:
                    pass
            print(line, flush=True)
            return 0
        print(f"bench.py: {' '.join(extra) or 'default transport'} failed with exit code {p.returncode}", file=sys.stderr)
    return last_rc or 1

