#!/usr/bin/env python3
"""bench.py -- ORB extract throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B]

One "step" = one pass of the hot path (RGBA frames in -> keypoints + BRIEF-256 descriptors out)
over one batch of B synthetic 1280x720 frames that are already resident in HBM (generated on the
device).  BASELINE.json configs[3]: B = 256 frames on one GPU.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank processes its own B frames (weak scaling:
configs[4] is 2048 frames over 8 GPUs = 256 per GPU) and each step ends with the collate of all
results to rank 0 over RCCL.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel against HBM bandwidth with
the algorithmic bytes of SURVEY.md 8(d); `cpu_baseline` is the CPU restatement (oracle/, "port")
timed on the host cores on a bounded sample of the same frames -- it is never the thing shipped.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, DEPTH, MAX_FEATURES = 1280, 720, 2, 8192
THRESHOLD = 20.0 / 255.0
SEED0 = 1000
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(n_sample, intended=False):
    """Times oracle/ (the CPU restatement) on the first n_sample frames of the workload."""
    import numpy as np
    from oracle import orb_oracle
    orb_oracle.build()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    cores = min(cores, 16)  # the CPU share of a one-GPU box
    frames = np.stack([orb_oracle.synth_frame(W, H, SEED0 + i) for i in range(n_sample)])
    t0 = time.perf_counter()
    if intended:
        totals, _, _ = orb_oracle.extract_intended_batch(frames, depth=DEPTH, threshold=THRESHOLD, max_features=MAX_FEATURES,
                                                         arc=9, nms=True, n_threads=cores)
    else:
        totals, _, _ = orb_oracle.extract_batch(frames, depth=DEPTH, threshold=THRESHOLD, max_features=MAX_FEATURES,
                                                n_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": n_sample / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d of the bench's 1280x720 frames (seeds %d..), oracle/orb_oracle.c frame-parallel over %d "
                      "threads, %.1f s wall" % (n_sample, SEED0, cores, dt),
            "keypoints_per_frame": float(np.minimum(totals, MAX_FEATURES).mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="frames for the CPU baseline (0 = skip)")
    ap.add_argument("--staged", action="store_true", help="force the one-kernel-per-stage pipeline")
    ap.add_argument("--mode", choices=("literal", "intended"), default="literal",
                    help="literal = the reference's algorithm (the headline, BASELINE.json); intended = the opt-in "
                         "repaired algorithm with FAST-9 + NMS (DESIGN.md section 8; not in the reference)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from tinyslam_amd import node, orb

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < n_dev else local_rank % max(n_dev, 1)  # rehearsal: ranks share a GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        # "nccl" is RCCL on ROCm.  TINYORB_DIST_BACKEND=gloo only exists to rehearse the N > 1 code path
        # on a one-GPU box (several ranks sharing device 0, which RCCL refuses).
        backend = os.environ.get("TINYORB_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    B = args.frames
    cfg = orb.OrbConfig(orb.Extent3d(W, H), max_features=MAX_FEATURES, hierarchy_depth=DEPTH,
                        initial_threshold=THRESHOLD, device=dev_index, max_batch=B,
                        flags=(orb.ORB_FLAG_STAGED if args.staged else 0) | (orb.ORB_FLAG_DOUBLE_OUTPUT if world > 1 else 0)
                        | ((orb.ORB_FLAG_INTENDED | orb.ORB_FLAG_NMS) if args.mode == "intended" else 0),
                        fast_arc=9 if args.mode == "intended" else 0)
    prog = orb.OrbProgram(cfg).init()
    frames_dev = prog.synth_frames_device(B, SEED0 + rank * B)  # rank g owns frames [g*B, (g+1)*B)
    n_sets = 2 if world > 1 else 1
    views = []
    for s_ in range(n_sets):
        prog.batch_select_output(s_)
        d_counts, d_corners, d_desc = prog.batch_device_buffers()
        views.append((node.as_tensor(d_counts, (B,), "<i4", dev), node.as_tensor(d_corners, (B, MAX_FEATURES, 4), "<i4", dev),
                      node.as_tensor(d_desc, (B, MAX_FEATURES, 8), "<i4", dev)))
    prog.batch_select_output(0)
    state = {"k": 0, "pending": None}
    free = [None, None]
    compute_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None

    def collate(pending):
        slot, done = pending
        counts_t, corners_t, desc_t = views[slot]
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(done)  # the kernels that wrote this output set
            node.collate_to_root(counts_t, corners_t, desc_t, MAX_FEATURES)
            free[slot] = torch.cuda.Event()
            free[slot].record(comm_stream)  # the gather has read this output set

    def step():
        """N = 1: extract.  N > 1: launch batch k into output set k%2 on the compute stream, then collate
        batch k-1 on the communication stream (it waits on the event recorded behind batch k-1's kernels):
        the RCCL gather of one batch overlaps the kernels of the next.  Every batch is collated inside the
        timed region (flush() drains the last one)."""
        if world == 1:
            prog.extract_batch_device(frames_dev, B)
            return
        slot = state["k"] & 1
        prog.batch_select_output(slot)
        if free[slot] is not None:
            compute_stream.wait_event(free[slot])  # do not overwrite a set that is still being gathered
        prog.extract_batch_device(frames_dev, B, stream=compute_stream.cuda_stream)
        done = torch.cuda.Event()
        done.record(compute_stream)
        if state["pending"] is not None:
            collate(state["pending"])
        state["pending"] = (slot, done)
        state["k"] += 1

    def flush():
        if world > 1 and state["pending"] is not None:
            collate(state["pending"])
            state["pending"] = None

    def fence():
        flush()
        prog.batch_sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    prog.profile_enable(True)
    prog.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prog.profile_enable(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    counts = prog.batch_counts(B)
    stored = np.minimum(counts, MAX_FEATURES)
    kp_local = torch.tensor([float(stored.sum())], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(kp_local, op=dist.ReduceOp.SUM)
    kp_per_step = float(kp_local.item())

    if rank == 0:
        prof = prog.profile()
        total_frames = B * world * args.steps
        fps = total_frames / elapsed
        # dominant kernel = largest accumulated device time inside the timed region
        dom = max(prof.items(), key=lambda kv: kv[1][0]) if prof else (None, (0.0, 0))
        n_mean = float(stored.mean())
        bytes_per_frame = 4 * W * H + 48 * n_mean + 4  # RGBA read once + records + counter
        roofline = None
        if dom[0]:
            avg_ms = dom[1][0] / dom[1][1]
            frames_per_launch = B * args.steps / dom[1][1]
            achieved = bytes_per_frame * frames_per_launch / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_ms,
                        "algorithmic_bytes_per_launch": bytes_per_frame * frames_per_launch,
                        "all_kernels_ms_per_step": {k: v[0] / args.steps for k, v in prof.items()}}
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("kernel") == dom[0] and tj.get("frames_per_launch") == frames_per_launch:
                    roofline["traffic"] = tj.get("hbm_bytes_per_launch")
                    roofline["traffic_source"] = tj.get("source")
        out = {
            "metric": "ORB extract throughput, 1280x720 (frames/sec; Mkeypoints/sec alongside)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (gradient+blobs+wedges+noise, seeds %d.., generated on device)" % SEED0,
            "config": {"workload": "BASELINE.json configs[3]: batch of %d independent 1280x720 RGBA frames per GPU, "
                                   "device-resident, full ORB (%s)"
                                   % (B, "FAST-12 + orientation + blur + BRIEF-256" if args.mode == "literal" else
                                      "opt-in intended mode, NOT the reference's algorithm: FAST-9 + NMS + full-circle "
                                      "orientation + separable Gaussian + BRIEF-256"),
                       "frames_per_gpu": B, "width": W, "height": H, "hierarchy_depth": DEPTH,
                       "max_features": MAX_FEATURES, "threshold": THRESHOLD, "mode": args.mode,
                       "pipeline": "staged" if args.staged else "default",
                       "collate": "RCCL gather of every batch to rank 0, overlapped with the next batch's kernels" if world > 1 else "none (1 GPU)"},
            "mkeypoints_per_s": kp_per_step * args.steps / elapsed / 1e6,
            "keypoints_per_frame": n_mean,
            "hbm_algorithmic_gbs": bytes_per_frame * fps / 1e9,
            "roofline": roofline,
        }
        if world == 1:
            n_cpu = args.cpu_sample if args.cpu_sample >= 0 else 128
            if n_cpu > 0:
                out["cpu_baseline"] = cpu_baseline(n_cpu, intended=args.mode == "intended")
        print(json.dumps(out), flush=True)
    prog.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
