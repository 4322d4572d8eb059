"""`python bench.py --gpus N` must start its own ranks (BASELINE.json configs[4]) without the launcher ever touching the
GPU: the parent spawns N children with the torch.distributed.run environment and only waits for them."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
import json, os, sys
out = os.path.join(os.environ["STUB_DIR"], "rank%s.json" % os.environ["RANK"])
json.dump({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR",
                                          "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY", "TINYORB_LAUNCHED_BY")}
          | {"argv": sys.argv[1:], "pid": os.getpid(), "ppid": os.getppid()}, open(out, "w"))
if os.environ["RANK"] == "0":
    print(json.dumps({"metric": "stub", "n_gpus": int(os.environ["WORLD_SIZE"])}), flush=True)
sys.exit(int(os.environ.get("STUB_EXIT_RANK%s" % os.environ["RANK"], "0")))
'''


def _run_launcher(tmp_path, world, extra_env=None):
    stub = tmp_path / "stub_worker.py"
    stub.write_text(STUB)
    report = tmp_path / "report.json"
    driver = (
        "import sys, json; sys.path.insert(0, %r)\n"
        "from tinyslam_amd import launch\n"
        "rc = launch.launch_ranks(%d, [%r, '--gpus', '%d', '--steps', '2'])\n"
        "sys.exit(rc)\n" % (ROOT, world, str(stub), world))
    env = dict(os.environ, STUB_DIR=str(tmp_path), TINYORB_LAUNCH_REPORT=str(report))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", driver], env=env, capture_output=True, text=True, timeout=120)
    return r, json.load(open(report))


def test_launcher_spawns_ranks_with_the_distributed_environment(tmp_path):
    r, rep = _run_launcher(tmp_path, 4)
    assert r.returncode == 0, r.stderr
    # rank 0's single JSON line is the launcher's stdout
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "stub", "n_gpus": 4}
    assert rep["world"] == 4 and len(rep["children"]) == 4
    # the launcher stayed a plain Python process: no torch, no HIP/HSA runtime, no libtinyorb, no RCCL mapped
    assert rep["torch_imported"] is False and rep["gpu_libraries_mapped"] == []
    ranks = [json.load(open(tmp_path / ("rank%d.json" % i))) for i in range(4)]
    assert sorted(x["pid"] for x in ranks) == sorted(rep["children"])
    ports = {x["MASTER_PORT"] for x in ranks}
    assert len(ports) == 1 and int(ports.pop()) == rep["port"]
    for i, x in enumerate(ranks):
        assert x["RANK"] == str(i) and x["LOCAL_RANK"] == str(i)
        assert x["WORLD_SIZE"] == "4" and x["LOCAL_WORLD_SIZE"] == "4"
        assert x["MASTER_ADDR"] == "127.0.0.1"
        assert x["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert x["ppid"] == rep["launcher_pid"] and x["TINYORB_LAUNCHED_BY"] == str(rep["launcher_pid"])
        assert x["argv"] == ["--gpus", "4", "--steps", "2"]


def test_launcher_relays_a_failing_rank(tmp_path):
    r, rep = _run_launcher(tmp_path, 2, {"STUB_EXIT_RANK1": "7"})
    assert r.returncode == 7


def test_launcher_kills_a_rank_that_ignores_sigterm(tmp_path):
    """Rank 1 fails; rank 0 ignores SIGTERM and would sleep for ten minutes (a rank blocked in a collective): the
    launcher must escalate to SIGKILL after its grace period and return rank 1's status."""
    stub = tmp_path / "stubborn.py"
    stub.write_text(
        "import os, signal, sys, time\n"
        "if os.environ['RANK'] == '1':\n"
        "    time.sleep(0.3); sys.exit(5)\n"
        "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
        "open(os.path.join(os.environ['STUB_DIR'], 'rank0.pid'), 'w').write(str(os.getpid()))\n"
        "time.sleep(600)\n")
    driver = (
        "import sys; sys.path.insert(0, %r)\n"
        "from tinyslam_amd import launch\n"
        "sys.exit(launch.launch_ranks(2, [%r], grace=1.0))\n" % (ROOT, str(stub)))
    env = dict(os.environ, STUB_DIR=str(tmp_path))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "-c", driver], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 5, (r.returncode, r.stderr)
    assert time.monotonic() - t0 < 30
    pid = int(open(tmp_path / "rank0.pid").read())
    try:
        os.kill(pid, 0)
        alive = True
    except OSError:
        alive = False
    assert not alive, "the stubborn rank survived the launcher"


def test_bench_hands_over_to_the_launcher_before_importing_torch(tmp_path):
    """bench.py --gpus 3 without WORLD_SIZE: main() must reach launch_ranks with torch not imported.  The launcher is
    replaced by a recorder so nothing is spawned."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import tinyslam_amd.launch as L\n"
        "def rec(world, argv, **kw):\n"
        "    print('LAUNCH', world, 'torch' in sys.modules, L._gpu_libraries_mapped(), argv[1:]); return 0\n"
        "L.launch_ranks = rec\n"
        "sys.argv = ['bench.py', '--gpus', '3', '--steps', '5', '--total-frames', '2048']\n"
        "import runpy; runpy.run_path(%r, run_name='__main__')\n" % (ROOT, os.path.join(ROOT, "bench.py")))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "LAUNCH 3 False [] ['--gpus', '3', '--steps', '5', '--total-frames', '2048']" in r.stdout


def test_bench_rank_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
