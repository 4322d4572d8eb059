#!/usr/bin/env python3
"""bench.py -- ORB extract throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B] [--total-frames T] [--repeats R]

One "step" = one pass of the hot path (RGBA frames in -> keypoints + BRIEF-256 descriptors out) over one batch of
synthetic 1280x720 frames that are already resident in HBM (generated on the device).

* default (weak scaling): every GPU processes its own B = 256 frames per step -- BASELINE.json configs[3] at N = 1,
  configs[4] (2048 frames over 8 GPUs = 256 per GPU) at N = 8;
* --total-frames T (strong scaling, e.g. 2048): a step is the whole T-frame job, sharded in contiguous ranges over the
  N GPUs, every rank working through its shard in B-frame batches.
With N > 1 every batch is collated on rank 0 over RCCL inside the timed region (the gather of one batch overlaps the
kernels of the next).

`python bench.py --gpus N` launches its own N rank processes (tinyslam_amd/launch.py) when it was not started by
torch.distributed.run; the launcher never touches the GPU.

The timed region is EXACTLY K steps between barrier + synchronize on both sides, max over ranks; it is repeated R
times and `value` / `ms_per_step` are the median repeat (all repeats in `repeats_ms_per_step`).  Kernel durations for
`roofline` come from a further pass of the same K steps with every launch bracketed by HIP events on its stream, so the
events are outside the timed repeats.  `cpu_baseline` is the CPU restatement (oracle/, "port") timed on the host cores
on a bounded sample of the same frames -- it is never the thing shipped.  Prints ONE JSON line on rank 0.
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
# --content: (synthetic-frame flags: 1 gradient, 2 blobs, 4 wedges, 8 noise; FAST threshold in 1/255)
CONTENT = {"flat": (1, 20), "sparse": (5, 20), "default": (15, 20), "dense": (15, 16), "overflow": (15, 8)}


def content_of(args):
    """(synthetic-frame flags, threshold) of --content."""
    flags, thr255 = CONTENT[args.content]
    return flags, thr255 / 255.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per batch")
    ap.add_argument("--total-frames", type=int, default=0,
                    help="strong scaling: a step is this many frames in all (BASELINE.json configs[4]: 2048), sharded "
                         "over the GPUs; 0 = weak scaling, --frames per GPU per step")
    ap.add_argument("--repeats", type=int, default=5, help="how often the K-step timed region is repeated (median reported)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="frames for the CPU baseline (0 = skip)")
    ap.add_argument("--no-host-out", action="store_true", help="skip the device-in -> host-out measurement (N = 1)")
    ap.add_argument("--collate", choices=("transport", "padded", "none"), default="transport",
                    help="N > 1: what the gather to rank 0 carries -- 40-byte transport records packed back to back, or "
                         "the 48-byte record slabs padded to the fullest frame (round 1's form); none = the results stay "
                         "sharded on their GPUs (a consumer that runs where the frames were extracted): kernels only")
    ap.add_argument("--force-collate", action="store_true",
                    help="N = 1: run the N > 1 collate all the same -- process group of one rank (\"nccl\" = RCCL), counters "
                         "all_gathered, records through the exact-size all_to_all (a send to itself), expansion on rank 0 -- "
                         "so that a one-GPU box executes the RCCL path the multi-GPU runs take")
    ap.add_argument("--content", choices=tuple(CONTENT), default="default",
                    help="what the synthetic frames hold (keypoints per 1280x720 frame in brackets): flat = the gradient alone [0]; "
                         "sparse = gradient + wedges [~650]; default = gradient + blobs + wedges + noise at the reference threshold "
                         "20/255 [~3.8 k, the headline]; dense = the same frames at threshold 16/255 [~7.7 k, just under "
                         "max_features]; overflow = threshold 8/255 [~64 k detected, 8192 stored: the band queues overflow and "
                         "every frame is cut at max_features]")
    ap.add_argument("--staged", action="store_true", help="force the one-kernel-per-stage pipeline")
    ap.add_argument("--input", choices=("rgba", "y8"), default="rgba",
                    help="rgba = the reference's input (the headline); y8 = the opt-in one-byte-per-pixel variant "
                         "(ORB_FLAG_INPUT_Y8, the reference's roadmap item; algorithmic bytes W*H + 48 N + 4)")
    ap.add_argument("--mode", choices=("literal", "intended"), default="literal",
                    help="literal = the reference's algorithm (the headline, BASELINE.json); intended = the opt-in "
                         "repaired algorithm with FAST-9 + NMS (DESIGN.md section 8; not in the reference)")
    ap.add_argument("--host", choices=("ranks", "node"), default="ranks",
                    help="ranks = one process per GPU over torch.distributed (the driver's contract); node = ONE process "
                         "driving all GPUs through the orb_node_* C ABI (what a Rust host binds), pipelined collate, no "
                         "torch.distributed")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="N = 1 only: consecutive batches alternate between this many programs, each on a stream of its own, so "
                         "that the tail of one batch's kernels overlaps the head of the next (default 1: one program, one stream "
                         "-- the line the roofline figures are accounted on)")
    ap.add_argument("--contract", type=int, default=0,
                    help="OrbOptions::fp_contract (CRD-13): the arithmetic of the adapter's shader compiler, a mask -- 1 luminance, 2 blur "
                         "taps, 4 BRIEF rotation as fused multiply-adds, 8 dot() / matrix * vector reduced from the last term; 7 = an "
                         "LLVM-style contracting compiler, 15 = Mesa with an fma, 8 = Mesa without one.  0 (default) = every product and "
                         "sum rounded: the headline")
    ap.add_argument("--angle-bins", type=int, default=0,
                    help="--mode intended only (IM-6b, OrbOptions::angle_bins): descriptors rotated by the centre of their keypoint's angle bin "
                         "instead of its milliradian code; 1024 bins = a 1 MB rotated-pattern table that stays in every XCD's L2")
    ap.add_argument("--no-single-frame", action="store_true", help="skip the single-frame latency figure (profiling runs)")
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed steps run for this long before the W warm-up steps, so that the clocks are up (the "
                         "first of five repeats used to read 10 %% slow)")
    return ap.parse_args(argv)


def cpu_baseline(n_sample, gpu_counts, intended=False, y8=False, syn_flags=15, threshold=THRESHOLD, fp=0, angle_bins=0):
    """Times oracle/ (the CPU restatement) on the first n_sample frames of the workload and checks that its per-frame
    counters equal the GPU's on those frames."""
    import numpy as np
    from oracle import orb_oracle
    orb_oracle.build()
    nproc = os.cpu_count() or 1          # what the box has
    usable = nproc
    try:
        usable = len(os.sched_getaffinity(0))  # what this process may run on
    except AttributeError:
        pass
    cores = min(usable, 16)  # threads used: the CPU share of a one-GPU box
    gen = orb_oracle.synth_frame_y8 if y8 else orb_oracle.synth_frame  # ORB_SYN_Y8: integer luma of the same recipe
    frames = np.stack([gen(W, H, SEED0 + i, syn_flags) for i in range(n_sample)])
    t0 = time.perf_counter()
    if intended:
        totals, _, _ = orb_oracle.extract_intended_batch(frames, depth=DEPTH, threshold=threshold, max_features=MAX_FEATURES,
                                                         arc=9, nms=True, n_threads=cores, angle_bins=angle_bins)
    else:
        totals, _, _ = orb_oracle.extract_batch(frames, depth=DEPTH, threshold=threshold, max_features=MAX_FEATURES,
                                                n_threads=cores, y8=y8, contract=fp & 7, dot_order=(fp >> 3) & 1)
    dt = time.perf_counter() - t0
    m = min(n_sample, len(gpu_counts))
    return {"value": n_sample / dt, "unit": "frames/s", "cores": cores, "host_cpus": nproc, "host_cpus_usable": usable, "kind": "port",
            "sample": "%d of the bench's 1280x720 frames (seeds %d..), oracle/orb_oracle.c frame-parallel over %d "
                      "threads, %.1f s wall" % (n_sample, SEED0, cores, dt),
            "keypoints_per_frame": float(np.minimum(totals, MAX_FEATURES).mean()),
            "counts_equal_gpu": bool(np.array_equal(np.asarray(totals[:m], dtype=np.int64),
                                                    np.asarray(gpu_counts[:m], dtype=np.int64))),
            "counts_compared": int(m)}


def roofline_of(args, prof, launches_frames, bytes_per_frame, profiled_s):
    """`roofline` of the dominant kernel from the profiled pass (HIP events around every launch)."""
    prof_k = {k_: v for k_, v in prof.items() if k_ not in ("k_compact", "k_compact_transport", "k_unpack_transport")}
    dom = max(prof_k.items(), key=lambda kv: kv[1][0]) if prof_k else (None, (0.0, 0))
    if not dom[0]:
        return None
    avg_ms = dom[1][0] / dom[1][1]
    frames_per_launch = launches_frames / dom[1][1]
    achieved = bytes_per_frame * frames_per_launch / (avg_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": bytes_per_frame * frames_per_launch,
                "frames_per_launch": frames_per_launch,
                "measured": "HIP events around every launch on the launch stream, in a pass of the same %d steps "
                            "right after the timed repeats (%.4f ms per step with the events in)"
                            % (args.steps, profiled_s / args.steps * 1e3),
                "all_kernels_ms_per_step": {k_: v[0] / args.steps for k_, v in prof.items()}}
    tpath = os.path.join(ROOT, "profiles", "traffic_%s_%s%s%s.json" % (args.mode, args.input, "_fp%d" % args.contract if args.contract else "",
                                                                        "_bins%d" % args.angle_bins if args.angle_bins else ""))
    if not os.path.exists(tpath) and not args.contract and not args.angle_bins:
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # the headline: literal mode, RGBA input
    # `traffic` is a committed measurement (rocprofv3 counter passes cannot run inside this process): it is only quoted when it
    # was taken on exactly these kernels -- the stamp is a hash of csrc/ -- and on this workload; otherwise null, with the reason
    if not os.path.exists(tpath):
        roofline["traffic_note"] = "no counter pass committed for this mode / input"
    else:
        from tinyslam_amd import build as orb_build
        tj = json.load(open(tpath))
        if args.content != "default":
            roofline["traffic_note"] = "the committed counter pass is of --content default"
        elif not (tj.get("kernel") == dom[0] and tj.get("frames_per_launch") == frames_per_launch
                  and tj.get("input", "rgba") == args.input and tj.get("mode", "literal") == args.mode):
            roofline["traffic_note"] = "the committed counter pass is of another kernel or batch size"
        elif tj.get("csrc_sha256") != orb_build.source_hash():
            roofline["traffic_note"] = ("stale: %s was taken on other kernel sources (csrc hash %s..., tree %s...); re-run "
                                        "tools/collect_profiles.sh" % (os.path.basename(tpath), str(tj.get("csrc_sha256"))[:12],
                                                                       orb_build.source_hash()[:12]))
        else:
            roofline["traffic"] = tj.get("hbm_bytes_per_launch")
            roofline["traffic_source"] = tj.get("source")
    return roofline


def workload_text(args, world, B, strong):
    return (("BASELINE.json configs[4]: one job of %d independent 1280x720 RGBA frames sharded over "
             "%d GPU(s), %d-frame batches" % (args.total_frames, world, B)) if strong else
            ("BASELINE.json configs[3]: batch of %d independent 1280x720 RGBA frames per GPU" % B)
            + ", device-resident, full ORB (%s)"
            % ("FAST-12 + orientation + blur + BRIEF-256" if args.mode == "literal" else
               "opt-in intended mode, NOT the reference's algorithm: FAST-9 + NMS + full-circle "
               "orientation + separable Gaussian + BRIEF-256"))


def single_frame_latency(orb, cfg_kwargs, n=200, n_loop=2000):
    """The reference's only call shape (orb.rs:469-557): one blocking extract_corners per frame, 1280x720.  Returns
    the keys of the bench line: mean microseconds of orb_extract_corners alone on a resident frame; frames/s of the reference's loop
    write_input_image -> extract_corners -> read_corners -> read_descriptors from a host frame; frames/s of the same loop with
    orb_write_input_image_pinned uploading frame k + 1 under the kernels of frame k -- each under both ways the call can wait for the
    device: "poll" (default: the host thread spins on a completion word) and "block" (ORB_FLAG_SINGLE_BLOCKING_WAIT: a bounded spin, then
    the thread sleeps until the completion interrupt, as the reference's device.poll(Wait) does)."""
    import numpy as np

    def measure(flags):
        kw = dict(cfg_kwargs)
        kw["flags"] = kw.get("flags", 0) | flags
        cfg = orb.OrbConfig(orb.Extent3d(W, H), max_batch=1, **kw)
        with orb.OrbProgram(cfg).init() as p1:
            dev = p1.synth_frames_device(1, SEED0)
            for _ in range(20):
                p1.extract_corners()
            t0 = time.perf_counter()
            for _ in range(n):
                p1.extract_corners()
            extract_us = (time.perf_counter() - t0) / n * 1e6
            frame = p1.copy_to_host(dev, W * H * 4)
            corners = np.zeros(MAX_FEATURES, dtype=orb.CORNER_DTYPE)
            desc = np.zeros((MAX_FEATURES, 8), dtype=np.uint32)

            def six_calls(write):
                """frames/s of n_loop iterations of write -> extract -> read -> read: (from the median iteration, from the total time,
                iterations that took more than ten times the median).  On a shared box a host thread that spins now and then loses its CPU
                for a scheduler slice (20 ms): the median is what the path costs, the total what a camera loop gets."""
                its = []
                for k in range(n_loop + 10):
                    a = time.perf_counter()
                    write()
                    p1.extract_corners()
                    p1.read_corners(corners)
                    p1.read_descriptors(desc)
                    if k >= 10:
                        its.append(time.perf_counter() - a)
                med = float(np.median(its))
                return 1.0 / med, len(its) / float(np.sum(its)), int(np.sum(np.asarray(its) > 10.0 * med))
            blocking = six_calls(lambda: p1.write_input_image(frame))
            pins = [orb.PinnedArray((H, W, 4), np.uint8) for _ in range(2)]
            for pn in pins:
                pn.array[:] = frame.reshape(H, W, 4)
            st = {"k": 0}

            def write_ahead():  # frame k + 1 goes up while frame k is extracted (one image ahead)
                p1.write_input_image_pinned(pins[st["k"] & 1].array)
                st["k"] += 1
            write_ahead()
            ahead = six_calls(write_ahead)
            p1.upload_sync()
            for pn in pins:
                pn.close()
        return extract_us, blocking, ahead

    def keys(r):
        return {"blocking_write": {"fps_median_iteration": r[1][0], "fps_total_time": r[1][1], "stalled_iterations": r[1][2]},
                "pinned_write_one_ahead": {"fps_median_iteration": r[2][0], "fps_total_time": r[2][1], "stalled_iterations": r[2][2]},
                "extract_us": r[0]}
    poll = measure(0)
    block = measure(orb.ORB_FLAG_SINGLE_BLOCKING_WAIT)
    return {"single_frame_us": poll[0], "single_frame_loop_fps": poll[1][0], "single_frame_loop_pinned_fps": poll[2][0],
            "single_frame_us_blocking_wait": block[0],
            "single_frame_loop": dict(keys(poll), wait="poll (default): the host thread spins on the completion word", iterations=n_loop,
                                      what="write_input_image -> extract_corners -> read_corners -> read_descriptors per 1280x720 frame from a host "
                                           "frame; pinned: orb_write_input_image_pinned uploads frame k + 1 under the kernels of frame k"),
            "single_frame_loop_blocking_wait": dict(keys(block), wait="block (ORB_FLAG_SINGLE_BLOCKING_WAIT / TINYORB_SINGLE_WAIT=block): 50 us of spinning, "
                                                                      "then asleep until the completion interrupt", iterations=n_loop)}


def run_node(args):
    """--host node: one process, the GPUs of the node through the orb_node_* C ABI (include/tinyorb.h) -- the path a
    Rust host binds.  A step = one job of B frames per GPU (weak) sharded over the devices; the collate of job k (pack,
    exact-size Send/Recv to the first device, expansion) overlaps the kernels of job k+1: extract(k), then
    collate_begin + collate_end of job k-1.  Every job is collated inside the timed region."""
    import numpy as np
    from tinyslam_amd import orb
    world, B = args.gpus, args.frames
    if args.total_frames:
        raise SystemExit("--host node runs the weak-scaling workload (one B-frame shard per GPU per step)")
    loop = os.environ.get("TINYORB_NODE_LOOPBACK", "0") not in ("", "0")
    devices = [0] * world if loop else list(range(world))
    syn_flags, threshold = content_of(args)
    cfg_kwargs = dict(max_features=MAX_FEATURES, hierarchy_depth=DEPTH, initial_threshold=threshold,
                      flags=(orb.ORB_FLAG_STAGED if args.staged else 0)
                      | ((orb.ORB_FLAG_INTENDED | orb.ORB_FLAG_NMS) if args.mode == "intended" else 0)
                      | (orb.ORB_FLAG_INPUT_Y8 if args.input == "y8" else 0),
                      fast_arc=9 if args.mode == "intended" else 0, fp_contract=args.contract, angle_bins=args.angle_bins)
    cfg = orb.OrbConfig(orb.Extent3d(W, H), max_batch=B, **cfg_kwargs)
    frame_bytes = W * H * (1 if args.input == "y8" else 4)
    F = B * world
    with orb.OrbNode(cfg, devices) as node:
        if args.collate == "none":  # results stay packed on the device that computed them (orb_node_set_results)
            node.set_results(True)
        progs = [node.program(r) for r in range(world)]
        ptrs = [progs[r].synth_frames_device(B, SEED0 + r * B, syn_flags) for r in range(world)]
        last = {}

        host = {"extract": 0.0, "begin": 0.0, "end": 0.0, "jobs": 0}  # host seconds inside the three calls of a job's pipeline

        def steps(n):
            for _ in range(n):
                t0 = time.perf_counter()
                node.extract_batch(ptrs, F)
                t1 = time.perf_counter()
                host["extract"] += t1 - t0
                host["jobs"] += 1
                if node.pending() == 2:
                    node.collate_begin()
                    t2 = time.perf_counter()
                    last["r"] = node.collate_end(F)
                    host["begin"] += t2 - t1
                    host["end"] += time.perf_counter() - t2

        def drain():
            while node.pending():
                last["r"] = node.collate_end(F)
            for pr in progs:
                pr.batch_sync()

        def timed(n):
            drain()
            t0 = time.perf_counter()
            steps(n)
            drain()
            return time.perf_counter() - t0

        t_end = time.perf_counter() + args.preheat_ms * 1e-3
        while time.perf_counter() < t_end:
            steps(2)
        steps(args.warmup)
        host.update(extract=0.0, begin=0.0, end=0.0, jobs=0)
        repeats = [timed(args.steps) for _ in range(max(1, args.repeats))]
        host_ms = {k: host[k] / max(1, host["jobs"]) * 1e3 for k in ("extract", "begin", "end")}
        # the same repeats with the results left sharded (orb_node_set_results): kernel scaling without the links into the first device
        sharded = None
        if args.collate != "none":
            drain()
            node.set_results(True)
            rs = [timed(args.steps) for _ in range(max(1, args.repeats))]
            drain()
            node.set_results(False)
            sharded = sorted(rs)[len(rs) // 2]
            steps(2)  # the profiled pass below is of the collated form again
        progs[0].profile_enable(True)
        progs[0].profile_reset()
        profiled = timed(args.steps)
        prof = progs[0].profile()
        progs[0].profile_enable(False)
        elapsed = sorted(repeats)[len(repeats) // 2]
        counts, offsets, _, _ = last["r"]
        kp_per_step = float(offsets[F])
        n_mean = kp_per_step / F
        bytes_per_frame = frame_bytes + 48 * n_mean + 4
        fps = F * args.steps / elapsed
        out = {
            "metric": "ORB extract throughput, 1280x720 (frames/sec; Mkeypoints/sec alongside)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (%s, seeds %d.., generated on device)"
                    % ("+".join(n for b, n in ((1, "gradient"), (2, "blobs"), (4, "wedges"), (8, "noise")) if syn_flags & b), SEED0),
            "config": {"workload": workload_text(args, world, B, False),
                       "frames_per_gpu_per_batch": B, "frames_per_step": F, "width": W, "height": H,
                       "hierarchy_depth": DEPTH, "max_features": MAX_FEATURES, "threshold": threshold, "content": args.content, "mode": args.mode,
                       "input": args.input, "pipeline": "staged" if args.staged else "default", "fp_contract": args.contract, "angle_bins": args.angle_bins,
                       "host": "node: one process, orb_node_* C ABI, no torch.distributed"
                               + (" (TINYORB_NODE_LOOPBACK: %d ranks on device 0, device copies instead of RCCL)" % world if loop else ""),
                       "collate": ("none: every job's records packed on the device that computed them (orb_node_set_results)"
                                   if args.collate == "none" else
                                   "every job packed and collated on the first device (exact-size transport records), "
                                   "overlapped with the next job's kernels")},
            "repeats_ms_per_step": [r / args.steps * 1e3 for r in repeats],
            "min_ms_per_step": min(repeats) / args.steps * 1e3, "max_ms_per_step": max(repeats) / args.steps * 1e3,
            "mkeypoints_per_s": kp_per_step * args.steps / elapsed / 1e6,
            "keypoints_per_frame": n_mean,
            "hbm_algorithmic_gbs": bytes_per_frame * fps / 1e9,
            "roofline": roofline_of(args, prof, B * args.steps, bytes_per_frame, profiled),
            "collate": {"host_extract_ms_per_job": host_ms["extract"], "host_collate_begin_ms_per_job": host_ms["begin"],
                        "host_collate_end_ms_per_job": host_ms["end"],
                        "host_ms_what": "the host thread inside orb_node_extract_batch (enqueue of kernels and packs on every device), "
                                        "orb_node_collate_begin (waits for the OLDEST job's pack events, enqueues its exchange) and "
                                        "orb_node_collate_end (waits for that exchange) per job of the timed repeats; a step's budget is ms_per_step"},
        }
        if sharded is not None:
            out["value_sharded"] = F * args.steps / sharded
            out["ms_per_step_sharded"] = sharded / args.steps * 1e3
            out["sharded_what"] = ("the same K steps x %d repeats with orb_node_set_results(ORB_NODE_RESULTS_SHARDED): every rank packs its own "
                                   "records, nothing crosses a link -- kernel scaling alone; `value` is the collated figure" % max(1, args.repeats))
        if world == 1:
            n_cpu = args.cpu_sample if args.cpu_sample >= 0 else 128
            if n_cpu > 0:
                out["cpu_baseline"] = cpu_baseline(n_cpu, counts[:B], intended=args.mode == "intended", y8=args.input == "y8",
                                                   syn_flags=syn_flags, threshold=threshold, fp=args.contract, angle_bins=args.angle_bins)
    if args.mode == "literal" and args.content == "default" and not args.staged and not args.no_single_frame:
        out.update(single_frame_latency(orb, cfg_kwargs))
    emit(out)


def run_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from tinyslam_amd import node, orb

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < n_dev else local_rank % max(n_dev, 1)  # rehearsal: ranks share a GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = None
    collating = world > 1 or args.force_collate  # every batch goes through the collate to rank 0
    if collating:
        # "nccl" is RCCL on ROCm.  TINYORB_DIST_BACKEND=gloo only exists to rehearse the N > 1 code path
        # on a one-GPU box (several ranks sharing device 0, which RCCL refuses).
        backend = os.environ.get("TINYORB_DIST_BACKEND", "nccl")
        if world == 1 and "MASTER_ADDR" not in os.environ:  # --force-collate without a launcher: a group of one
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    B = args.frames
    strong = args.total_frames > 0
    if strong:
        lo, hi = node.shard_range(args.total_frames, world, rank)
    else:
        lo, hi = rank * B, (rank + 1) * B  # rank g owns frames [g*B, (g+1)*B)
    n_local = hi - lo
    batches = [(b0, min(B, n_local - b0)) for b0 in range(0, n_local, B)]  # (first local frame, frames)
    job_frames = args.total_frames if strong else B * world

    syn_flags, threshold = content_of(args)
    cfg = orb.OrbConfig(orb.Extent3d(W, H), max_features=MAX_FEATURES, hierarchy_depth=DEPTH,
                        initial_threshold=threshold, device=dev_index, max_batch=B,
                        flags=(orb.ORB_FLAG_STAGED if args.staged else 0) | orb.ORB_FLAG_DOUBLE_OUTPUT
                        | ((orb.ORB_FLAG_INTENDED | orb.ORB_FLAG_NMS) if args.mode == "intended" else 0)
                        | (orb.ORB_FLAG_INPUT_Y8 if args.input == "y8" else 0),
                        fast_arc=9 if args.mode == "intended" else 0, fp_contract=args.contract, angle_bins=args.angle_bins)
    prog = orb.OrbProgram(cfg).init()
    frame_bytes = W * H * (1 if args.input == "y8" else 4)
    frames_t = torch.empty(max(n_local, 1) * frame_bytes, dtype=torch.uint8, device=dev)  # this rank's shard, in HBM
    for b0, nb in batches:
        prog.synth_frames_device(nb, SEED0 + lo + b0, syn_flags, frames_dev_ptr=frames_t.data_ptr() + b0 * frame_bytes)
    views = []
    for s_ in range(2):
        prog.batch_select_output(s_)
        d_counts, d_corners, d_desc = prog.batch_device_buffers()
        views.append((node.as_tensor(d_counts, (B,), "<i4", dev), node.as_tensor(d_corners, (B, MAX_FEATURES, 4), "<i4", dev),
                      node.as_tensor(d_desc, (B, MAX_FEATURES, 8), "<i4", dev)))
    prog.batch_select_output(0)
    state = {"k": 0, "pending": None, "gathered_bytes": 0, "expected_bytes": 0}
    # --in-flight n (N = 1): n programs (each with its own planes and lists) on n streams, batch k on program k % n
    fly = max(1, args.in_flight) if not collating else 1
    fly_progs = [prog] + [orb.OrbProgram(cfg).init() for _ in range(fly - 1)]
    fly_streams = [torch.cuda.Stream(device=dev) for _ in range(fly)] if fly > 1 else []
    free = [None, None]
    compute_stream = torch.cuda.Stream(device=dev) if collating else None
    comm_stream = torch.cuda.Stream(device=dev) if collating else None

    # N > 1, transport collate: per output set a buffer for this rank's packed 40-byte records
    transport = collating and args.collate == "transport"
    tbuf = [torch.empty((B * MAX_FEATURES, node.TRANSPORT_WORDS), dtype=torch.int32, device=dev) for _ in range(2)] if transport else None

    # N > 1, transport form: the lagged exact-size collator (no host stall per batch, no byte too many) and, on rank 0,
    # the two record arrays of a whole job per output set, allocated once
    collator = node.TransportCollator(B, MAX_FEATURES, dev) if transport else None
    all_records = world * B * MAX_FEATURES
    corners_all = [torch.empty((all_records, 4), dtype=torch.int32, device=dev) for _ in range(2)] if transport and rank == 0 else None
    desc_all = [torch.empty((all_records, 8), dtype=torch.int32, device=dev) for _ in range(2)] if transport and rank == 0 else None
    state["ticket"] = None
    state["host_s"] = 0.0

    def exchange_ticket():
        """Batch k-1, one batch later: its counters are on the host, so its records move now, exactly sized, and rank 0
        expands them into the reference's two record arrays, frames of the job in order."""
        if state["ticket"] is None:
            return
        before = collator.bytes_exchanged
        info = collator.exchange(state["ticket"])
        state["ticket"] = None
        state["gathered_bytes"] += collator.bytes_exchanged - before
        state["expected_bytes"] += 4 * node.TRANSPORT_WORDS * sum(info["totals"])
        if rank == 0:
            sl, first = info["slot"], np.asarray(info["first"], dtype=np.uint64)
            prog.unpack_transport(info["merged"].data_ptr(), first[:-1], info["totals"], first[:-1],
                                  corners_all[sl].data_ptr(), desc_all[sl].data_ptr(), stream=comm_stream.cuda_stream)
            state["last"] = (sl, torch.from_numpy(info["counts_all"]), first, corners_all[sl], desc_all[sl])

    def collate(pending):
        slot, done, nb = pending
        counts_t, corners_t, desc_t = views[slot]
        t_host = time.perf_counter()
        with torch.cuda.stream(comm_stream):
            if state.get("sharded"):  # the second set of repeats: the results stay on the GPU that computed them (as --collate none)
                comm_stream.wait_event(done)
            elif transport:
                exchange_ticket()  # the batch before: its counters are on the host, its records move now
                comm_stream.wait_event(done)  # the kernels that wrote this output set
                cs = comm_stream.cuda_stream
                prog.batch_pack_transport(slot, B, tbuf[slot].data_ptr(), B * MAX_FEATURES, stream=cs)
                state["ticket"] = collator.submit(slot, counts_t, tbuf[slot])
            elif args.collate == "padded":
                comm_stream.wait_event(done)
                out = node.collate_to_root(counts_t, corners_t, desc_t, MAX_FEATURES)
                if out is not None:
                    state["gathered_bytes"] += out[1].numel() * 4 + out[2].numel() * 4
            else:  # none: nothing leaves the GPU
                comm_stream.wait_event(done)
            free[slot] = torch.cuda.Event()
            free[slot].record(comm_stream)  # the pack (transport) or the gather has read this output set
        state["host_s"] += time.perf_counter() - t_host

    def step():
        """One pass over this rank's frames.  N = 1: extract, batch by batch.  N > 1: batch k goes into output set k%2 on
        the compute stream, then batch k-1 is collated on the communication stream (it waits on the event recorded
        behind batch k-1's kernels): the RCCL gather of one batch overlaps the kernels of the next.  Every batch is
        collated inside the timed region (flush() drains the last one).  In the strong-scaling mode every rank runs the
        same number of collates (an empty shard tail still takes part with zero counts)."""
        for b0, nb in batches:
            ptr = frames_t.data_ptr() + b0 * frame_bytes
            if not collating:
                if fly > 1 and not state.get("serial"):
                    i = state["k"] % fly
                    fly_progs[i].extract_batch_device(ptr, nb, stream=fly_streams[i].cuda_stream)
                    state["k"] += 1
                else:
                    prog.extract_batch_device(ptr, nb)
                continue
            slot = state["k"] & 1
            prog.batch_select_output(slot)
            if free[slot] is not None:
                compute_stream.wait_event(free[slot])  # do not overwrite a set that is still being gathered
            if nb < B:  # ragged last batch: frames past nb must read as empty in the gather
                with torch.cuda.stream(compute_stream):
                    views[slot][0][nb:].zero_()
            prog.extract_batch_device(ptr, nb, stream=compute_stream.cuda_stream)
            done = torch.cuda.Event()
            done.record(compute_stream)
            if state["pending"] is not None:
                collate(state["pending"])
            state["pending"] = (slot, done, nb)
            state["k"] += 1

    def flush():
        if collating and state["pending"] is not None:
            collate(state["pending"])
            state["pending"] = None
        if transport:
            with torch.cuda.stream(comm_stream):
                exchange_ticket()

    def fence():
        flush()
        prog.batch_sync()
        torch.cuda.synchronize()
        if collating:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if collating:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if strong and world > 1:
        n_b = torch.tensor([len(batches)], device=dev)
        mx_b = n_b.clone()
        dist.all_reduce(mx_b, op=dist.ReduceOp.MAX)
        if int(mx_b.item()) != len(batches):
            raise SystemExit("--total-frames %d does not split into the same number of %d-frame batches on every rank"
                             % (args.total_frames, B))

    if world == 1:  # clocks up before the W warm-up steps (N > 1: every rank would have to agree on the count)
        t_end = time.perf_counter() + args.preheat_ms * 1e-3
        while time.perf_counter() < t_end:
            step()
            prog.batch_sync()
    for _ in range(args.warmup):
        step()
    state["host_s"] = 0.0
    if collator is not None:
        collator.wait_s = 0.0
    repeats = [timed(args.steps) for _ in range(max(1, args.repeats))]
    n_collates = max(1, len(repeats) * args.steps * max(1, len(batches)))
    collate_host_ms = state["host_s"] / n_collates * 1e3
    # of which: waiting for the lagged counters' copy (TransportCollator.exchange; normally an event that is long set) -- the rest is
    # Python + enqueue cost (pack launch, all_gather, all_to_all_single, unpack launch)
    collate_wait_ms = (collator.wait_s / n_collates * 1e3) if collator is not None else 0.0
    # The same timed repeats with the results left sharded (nothing packed, gathered or exchanged; the stream ordering stays): what the
    # kernels scale like, beside `value`, which at N > 1 also carries the links into rank 0 -- one line separates the two.
    sharded = None
    if collating and args.collate != "none":
        fence()
        state["sharded"] = True
        rs = [timed(args.steps) for _ in range(max(1, args.repeats))]
        fence()
        state["sharded"] = False
        sharded = sorted(rs)[len(rs) // 2]
    state["gathered_bytes"] = 0
    prog.profile_enable(True)
    prog.profile_reset()
    state["serial"] = True  # --in-flight n: the per-kernel figures come from one program on one stream (no kernel shares the chip)
    profiled = timed(args.steps)  # same K steps with HIP events around every launch
    prof = prog.profile()
    prog.profile_enable(False)
    gathered_per_step = state["gathered_bytes"] / max(1, args.steps)
    elapsed = sorted(repeats)[len(repeats) // 2]

    # keypoints of this rank's shard (last batch's counters stand for all of them only in the weak mode; in the
    # strong mode every batch is visited once more, outside any timed region)
    stored_local, counts_first = 0.0, None
    for b0, nb in batches:
        if len(batches) > 1 or collating:
            prog.batch_select_output(0)
            prog.extract_batch_device(frames_t.data_ptr() + b0 * frame_bytes, nb)
        c = prog.batch_counts(nb)
        if counts_first is None:
            counts_first = c.copy()
        stored_local += float(np.minimum(c, MAX_FEATURES).sum())
    kp = torch.tensor([stored_local], dtype=torch.float64, device=dev)
    if collating:
        dist.all_reduce(kp, op=dist.ReduceOp.SUM)
    kp_per_step = float(kp.item())

    # collate alone (N > 1): the same gather, serialised, to price the links into rank 0
    collate_info = None
    if collating:
        fence()
        prog.batch_select_output(0)
        prog.extract_batch_device(frames_t.data_ptr(), batches[0][1] if batches else 0, stream=compute_stream.cuda_stream)
        done = torch.cuda.Event()
        done.record(compute_stream)
        fence()
        state["gathered_bytes"] = state["expected_bytes"] = 0
        n_rep = 10
        t0 = time.perf_counter()
        for _ in range(n_rep):
            collate((0, done, batches[0][1]))
        if transport:
            with torch.cuda.stream(comm_stream):
                exchange_ticket()
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        per = state["gathered_bytes"] / n_rep
        from_peers = per * (world - 1) / world
        root_check = None
        if transport and rank == 0 and state.get("last"):
            # rank 0's own frames must have come through pack -> gather -> unpack unchanged (device-side compare, untimed)
            slot_l, counts_all, first, corners_all, desc_all = state["last"]
            own = torch.clamp(views[slot_l][0], max=MAX_FEATURES).tolist()
            root_check = int(first[-1]) == int(torch.clamp(counts_all, max=MAX_FEATURES).sum().item())
            at = 0
            for f in range(0, B, max(B // 8, 1)):
                at = int(sum(own[:f]))
                root_check = root_check and torch.equal(corners_all[at:at + own[f]], views[slot_l][1][f, :own[f]]) \
                    and torch.equal(desc_all[at:at + own[f]], views[slot_l][2][f, :own[f]])
        collate_info = {"form": args.collate, "bytes_per_keypoint": 40 if transport else 48, "root_check": root_check,
                        "bytes_gathered_per_batch": per, "bytes_from_peers_per_batch": from_peers,
                        "ms_alone_per_batch": dt / n_rep * 1e3, "gbs_into_root": from_peers / (dt / n_rep) / 1e9,
                        "gbs_per_link": from_peers / (dt / n_rep) / 1e9 / max(world - 1, 1),
                        "bytes_gathered_per_step_timed": gathered_per_step, "backend": backend,
                        "host_ms_per_batch": collate_host_ms,
                        "host_wait_ms_per_batch": collate_wait_ms, "host_enqueue_ms_per_batch": collate_host_ms - collate_wait_ms,
                        "host_ms_what": "rank 0's Python thread inside collate() per batch of the timed repeats: host_wait = blocked on the event behind "
                                        "the lagged counters' copy (node.TransportCollator.exchange), host_enqueue = everything else (pack launch, "
                                        "all_gather, all_to_all_single, unpack launch); a step's budget is ms_per_step",
                        "exact": bool(transport and state["gathered_bytes"] == state["expected_bytes"]),
                        "how": ("exact and lagged: batch k's counters are all_gathered and copied to pinned host memory when it "
                                "is packed; one batch later every rank reads the same totals S_r and one all_to_all_single with "
                                "split sizes moves exactly sum(S_r) x 40 bytes into rank 0; buffers allocated once"
                                if transport else "padded slabs")}
        if world == 1:
            collate_info["forced"] = "N = 1 with --force-collate: the group has one rank, the exchange is a send to itself"

    # device-resident in -> host-resident out (N = 1): every batch is packed on the device (orb_batch_pack) and fetched
    # into pinned host memory by two exact-size DMA copies on a second stream (orb_batch_fetch) while the next batch
    # computes; two output sets, two host buffers
    host_out = None
    if world == 1 and not collating and not args.no_host_out and batches:
        copy_stream = torch.cuda.Stream(device=dev)
        cs = torch.cuda.current_stream(dev)
        hbs = [orb.HostBatch(B, B * MAX_FEATURES) for _ in range(2)]
        ev_free = [None, None]
        st = {"k": 0, "pending": None}

        def fetch(pending):
            pslot, pnb = pending
            prog.batch_fetch(pslot, hbs[pslot], stream=copy_stream.cuda_stream)  # waits for that pack on the host
            ev_free[pslot] = torch.cuda.Event()
            ev_free[pslot].record(copy_stream)

        def host_step():
            for b0, nb in batches:
                slot = st["k"] & 1
                prog.batch_select_output(slot)
                if ev_free[slot] is not None:
                    cs.wait_event(ev_free[slot])  # the copies of the batch before last have left this set's buffers
                prog.extract_batch_device(frames_t.data_ptr() + b0 * frame_bytes, nb, stream=cs.cuda_stream)
                prog.batch_pack(nb, stream=cs.cuda_stream)
                if st["pending"] is not None:
                    fetch(st["pending"])
                st["pending"] = (slot, nb)
                st["k"] += 1

        def host_flush():
            if st["pending"] is not None:
                fetch(st["pending"])
                st["pending"] = None
            torch.cuda.synchronize()
        for _ in range(2):
            host_step()
        host_flush()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            host_step()
        host_flush()
        dt = time.perf_counter() - t0
        last = hbs[(st["k"] - 1) & 1]
        total_rec = int(last.offsets[batches[-1][1]])
        host_counts_ok = bool(np.array_equal(last.counts[:batches[-1][1]], prog.batch_counts(batches[-1][1])))
        host_out = {"frames_per_s": job_frames * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                    "bytes_to_host_per_batch": total_rec * 48 + batches[-1][1] * 12, "counts_match_device": host_counts_ok,
                    "how": "orb_batch_pack packs a batch's records on the device, orb_batch_fetch copies exactly those "
                           "bytes to pinned host memory (two DMA copies on a second stream) while the next batch computes"}
        host_out["pcie_gbs"] = host_out["bytes_to_host_per_batch"] * len(batches) * args.steps / dt / 1e9
        prog.batch_select_output(0)
        for hb in hbs:
            hb.close()

    if rank == 0:
        total_frames = job_frames * args.steps
        fps = total_frames / elapsed
        launches_frames = n_local * args.steps  # frames this rank pushed through each kernel in the profiled pass
        n_mean = kp_per_step / job_frames
        # the frame read once (RGBA: 4 B per pixel, Y8: 1) + records + counter
        bytes_per_frame = frame_bytes + 48 * n_mean + 4
        roofline = roofline_of(args, prof, launches_frames, bytes_per_frame, profiled)
        out = {
            "metric": "ORB extract throughput, 1280x720 (frames/sec; Mkeypoints/sec alongside)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (%s, seeds %d.., generated on device)"
                    % ("+".join(n for b, n in ((1, "gradient"), (2, "blobs"), (4, "wedges"), (8, "noise")) if syn_flags & b), SEED0),
            "config": {"workload": workload_text(args, world, B, strong),
                       "frames_per_gpu_per_batch": B, "frames_per_step": job_frames, "width": W, "height": H,
                       "hierarchy_depth": DEPTH, "max_features": MAX_FEATURES, "threshold": threshold, "content": args.content, "mode": args.mode,
                       "input": args.input,
                       "pipeline": "staged" if args.staged else "default", "fp_contract": args.contract, "angle_bins": args.angle_bins,
                       "batches_in_flight": fly,
                       "collate": ("none: results stay sharded on their GPUs" if args.collate == "none" else
                                   "RCCL exchange of every batch to rank 0 (exact sizes), overlapped with the next batch's kernels")
                                  if collating else "none (1 GPU)"},
            "repeats_ms_per_step": [r / args.steps * 1e3 for r in repeats],
            "min_ms_per_step": min(repeats) / args.steps * 1e3, "max_ms_per_step": max(repeats) / args.steps * 1e3,
            "mkeypoints_per_s": kp_per_step * args.steps / elapsed / 1e6,
            "keypoints_per_frame": n_mean,
            "hbm_algorithmic_gbs": bytes_per_frame * fps / 1e9,
            "roofline": roofline,
        }
        if collate_info:
            out["collate"] = collate_info
        if sharded is not None:
            out["value_sharded"] = total_frames / sharded
            out["ms_per_step_sharded"] = sharded / args.steps * 1e3
            out["sharded_what"] = ("the same K steps x %d repeats with the results left on the GPU that computed them (no pack, no all_gather, no "
                                   "exchange): kernel scaling alone; `value` is the collated figure the metric asks for" % max(1, args.repeats))
        if host_out:
            out["host_out_frames_per_s"] = host_out["frames_per_s"]
            out["host_out"] = host_out
        if world == 1:
            n_cpu = args.cpu_sample if args.cpu_sample >= 0 else 128
            if n_cpu > 0:
                out["cpu_baseline"] = cpu_baseline(n_cpu, counts_first, intended=args.mode == "intended", y8=args.input == "y8",
                                                   syn_flags=syn_flags, threshold=threshold, fp=args.contract, angle_bins=args.angle_bins)
            if args.mode == "literal" and args.input == "rgba" and args.content == "default" and not args.staged and not args.no_single_frame:
                # the reference's only call shape (orb.rs:469-557): one blocking extract per frame, microseconds per call
                out.update(single_frame_latency(orb, dict(max_features=MAX_FEATURES, hierarchy_depth=DEPTH, initial_threshold=THRESHOLD,
                                                          device=dev_index, fp_contract=args.contract)))
        emit(out)
    prog.close()
    if collating:
        dist.barrier()
        dist.destroy_process_group()


_RESULT_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  librccl prints a version banner on stdout when a communicator is
    created (RCCL 2.26/2.27: "RCCL version : ...", five lines, no switch for it), so the process keeps a private
    duplicate of stdout for the result line and points file descriptor 1 at stderr for everything else."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(out):
    line = (json.dumps(out) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def main():
    args = parse_args()
    if args.host == "node":
        claim_stdout()
        return run_node(args)  # one process for all GPUs: nothing to launch
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not started by torch.distributed.run: spawn the ranks ourselves -- before torch, libtinyorb or anything else
        # that could initialise the GPU is imported into this process (a process that holds a HIP context must not fork
        # or exec workers; this one only waits and relays the exit status)
        from tinyslam_amd import launch
        sys.exit(launch.launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    claim_stdout()
    run_rank(args)


if __name__ == "__main__":
    main()
