"""One-node rank launcher for `bench.py --gpus N` (BASELINE.json configs[4]).

`python bench.py --gpus N` must work unaided.  When WORLD_SIZE is not in the environment and N > 1, bench.py calls
`launch_ranks` BEFORE it imports torch, libtinyorb or anything else that could touch the GPU: a process that has
initialised HIP must never fork/exec workers, so the parent stays a plain Python process that only spawns N children
(one rank per GPU, the environment torch.distributed.run would give them), waits, and exits with the worst child
status.  The children inherit stdout/stderr: rank 0's single JSON line is the parent's output.

This module imports nothing but the standard library.
"""
import json
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    """Environment of one rank: what `torch.distributed.run --nnodes=1 --nproc-per-node world` sets."""
    env = dict(os.environ if base is None else base)
    env.update({
        "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
        "GROUP_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
        "TINYORB_LAUNCHED_BY": str(os.getpid()),
    })
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    return env


def _gpu_libraries_mapped():
    """Names of GPU runtime libraries mapped into THIS process (must be empty in the launcher)."""
    found = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                for name in ("libamdhip64", "libhsa-runtime64", "libtinyorb", "librccl", "libtorch"):
                    if name in line:
                        found.add(name)
    except OSError:
        pass
    return sorted(found)


def _kill_all(procs):
    for p in procs:
        if p.poll() is None:
            try:
                p.kill()
            except OSError:
                pass
    for p in procs:
        try:
            p.wait(timeout=5)
        except Exception:
            pass


def launch_ranks(world, argv, worker=None, timeout=None, grace=10.0):
    """Spawns `world` rank processes running `worker + argv` (default worker: this interpreter on bench.py's path in
    argv[0]) and waits for them.  Returns the worst exit status.  If a rank fails (or `timeout` passes) the others get
    SIGTERM by PID and, `grace` seconds later, SIGKILL: a rank blocked in a driver call or a collective does not die on
    SIGTERM, and the launcher must not hang on it.  Nothing is ever restarted."""
    if worker is None:
        worker = [sys.executable]
    port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
    procs = []
    try:
        for r in range(world):
            procs.append(subprocess.Popen(list(worker) + list(argv), env=rank_env(r, world, port)))
    except BaseException:
        _kill_all(procs)  # rank k could not be started: ranks 0..k-1 must not stay behind
        raise
    report = os.environ.get("TINYORB_LAUNCH_REPORT")
    if report:  # test hook: what the launcher did, and proof that it stayed off the GPU
        with open(report, "w") as f:
            json.dump({"launcher_pid": os.getpid(), "children": [p.pid for p in procs], "world": world, "port": port,
                       "torch_imported": "torch" in sys.modules, "gpu_libraries_mapped": _gpu_libraries_mapped()}, f)
    t0 = time.monotonic()
    status = 0
    live = list(procs)
    kill_at = None  # set when the survivors were sent SIGTERM
    try:
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and kill_at is None:
                    status = status or (rc if rc > 0 else 128 - rc)
                    for q in live:  # one rank failed: the others would wait for it in a collective forever
                        q.terminate()
                    kill_at = time.monotonic() + grace
                elif rc != 0:
                    status = status or (rc if rc > 0 else 128 - rc)
            now = time.monotonic()
            if timeout is not None and kill_at is None and now - t0 > timeout and live:
                status = status or 124
                for q in live:
                    q.terminate()
                kill_at = now + grace
            if kill_at is not None and now > kill_at and live:
                for q in live:
                    q.kill()
                kill_at = float("inf")
            if live:
                time.sleep(0.05)
    finally:
        _kill_all(procs)  # make sure nothing survives us
    return status
