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


def launch_ranks(world, argv, worker=None, timeout=None):
    """Spawns `world` rank processes running `worker + argv` (default worker: this interpreter on bench.py's path in
    argv[0]) and waits for them.  Returns the worst exit status.  If a rank fails the others are terminated by PID."""
    if worker is None:
        worker = [sys.executable]
    port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(list(worker) + list(argv), env=rank_env(r, world, port)))
    report = os.environ.get("TINYORB_LAUNCH_REPORT")
    if report:  # test hook: what the launcher did, and proof that it stayed off the GPU
        with open(report, "w") as f:
            json.dump({"launcher_pid": os.getpid(), "children": [p.pid for p in procs], "world": world, "port": port,
                       "torch_imported": "torch" in sys.modules, "gpu_libraries_mapped": _gpu_libraries_mapped()}, f)
    t0 = time.monotonic()
    status = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                status = status or (rc if rc > 0 else 128 - rc)
                for q in live:  # one rank failed: the others would wait for it in a collective forever
                    q.terminate()
        if timeout is not None and time.monotonic() - t0 > timeout and live:
            status = status or 124
            for q in live:
                q.terminate()
            timeout = None
        if live:
            time.sleep(0.05)
    for p in procs:  # make sure nothing survives us
        if p.poll() is None:
            p.send_signal(signal.SIGKILL)
    return status
