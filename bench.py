#!/usr/bin/env python3
"""bench.py -- LiDAR frames/s of Slam::AddFrame (end to end) on synthetic VLS-128 sequences.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one AddFrame (keypoint extraction + ego-motion ICP + localization ICP + map update) on one
VLS-128-shaped scan (128 x 2048 firings, ~260k points) that is already resident in HBM (frame
store); every rank replays its own independent sequence (seed 1000 + rank) on its own GPU -- the
path shards by sequence, so scaling is weak and the only exchange is the RCCL
all-gather of the 4x4 pose + stamp of every sequence after each step.
Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline      achieved algorithmic GB/s of the dominant kernel (HIP events on the context's own
                stream, live during the timed region) against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle ("port": restatement of the reference algorithm, OpenMP where the
                reference has it) on a bounded sample of the same sequence, on the host cores
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", type=int, default=128, help="16 VLP-16, 64 HDL-64, 128 VLS-128 (headline)")
    ap.add_argument("--cpu-frames", type=int, default=160, help="frames of the CPU baseline sample (0 disables); 160 VLS-128 frames are about 10 s of CPU work")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores available to this process")
    ap.add_argument("--param", action="append", default=[], help="Slam parameter override NAME=VALUE (reference setter names)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events at all")
    ap.add_argument("--host-frames", action="store_true", help="hand every scan over from a host buffer (PCIe inclusive; never the headline value)")
    ap.add_argument("--sequences-per-gpu", type=int, default=1, help="independent sequences replayed side by side on every GPU (the headline is 1: BASELINE.json shards 1 per GPU)")
    ap.add_argument("--batch-sequences", type=int, default=4, help="N=1 only: extra leg with this many sequences side by side on the GPU, reported as batch_replay (0 disables)")
    ap.add_argument("--no-lookahead", action="store_true", help="do not extract the next stored frame's keypoints beside the current frame's registration")
    ap.add_argument("--no-numa-bind", action="store_true", help="leave the host threads wherever the scheduler puts them (default: on the GPU's NUMA node)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl (= RCCL over xGMI, the real run); gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks then share devices)")
    ap.add_argument("--profile-all", action="store_true", help="time every scope in the timed region too (costs ~10 %% of the frame rate)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    import lidarslam_amd as L
    from lidarslam_amd.replay import PoseExchange, sequence_seed

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()  # rehearsal: more ranks than GPUs
    torch.cuda.set_device(local_rank)
    numa_node = None if args.no_numa_bind else L.bind_host_to_device(local_rank)  # before any worker thread exists
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    # ---- inputs: one independent sequence per rank, generated on the host, made resident in HBM
    total = args.warmup + args.steps
    slam = L.Slam(local_rank, EgoMotion=3)  # MOTION_EXTRAPOLATION_AND_REGISTRATION: the default mode skips the ego-motion ICP
    for kv in args.param:
        name, value = kv.split("=")
        slam.set_param(name, float(value))
    seed = sequence_seed(rank)
    stamps, npts, host_frames = [], 0, []
    for f in range(total):
        pts, stamp = L.synth_frame(args.model, seed, f)
        if args.host_frames:
            host_frames.append(pts)
        else:
            slam.store_frame(f, pts)
        stamps.append(stamp)
        npts += pts.size
    ctx = slam.context()

    # further sequences of this rank (--sequences-per-gpu): a Slam, a context and a host thread each, free-running
    # beside the first one between the same start and end barriers; their latest poses travel with the first one's
    per_gpu = max(args.sequences_per_gpu, 1)
    lookahead = not args.no_lookahead and not args.host_frames
    latest = np.zeros((per_gpu, 17))
    latest_lock = threading.Lock()
    others, gate = [], threading.Barrier(per_gpu)
    for s in range(1, per_gpu):
        o = L.Slam(local_rank, EgoMotion=3)
        for kv in args.param:
            name, value = kv.split("=")
            o.set_param(name, float(value))
        st = []
        for f in range(total):
            pts, stamp = L.synth_frame(args.model, sequence_seed(rank * per_gpu + s), f)
            o.store_frame(f, pts)
            st.append(stamp)
            npts += pts.size
        others.append((o, st))

    def publish(s, o, stamp):
        with latest_lock:
            latest[s, :16] = np.asarray(o.world_transform()).reshape(16)
            latest[s, 16] = stamp * 1e-6

    def follow(s, o, st):
        try:
            for f in range(total):
                if f == args.warmup:
                    gate.wait()
                if lookahead and f + 1 < total:
                    o.hint_next_stored_frame(f + 1)
                o.add_stored_frame(f, st[f], f)
                publish(s, o, st[f])
            o.context().sync()
        except BaseException:
            gate.abort()  # the first sequence then stops with BrokenBarrierError instead of waiting for ever
            raise
        gate.wait()

    followers = [threading.Thread(target=follow, args=(s + 1, o, st), daemon=True) for s, (o, st) in enumerate(others)]

    # RCCL pose broadcast of the north star: every rank ends up with every sequence's pose table
    exchange = PoseExchange(world, device="cuda" if args.backend == "nccl" else "cpu", per_rank=per_gpu)

    def step(f):
        if args.host_frames:
            slam.add_frame(host_frames[f], stamps[f], f)
        else:
            if lookahead and f + 1 < total:
                slam.hint_next_stored_frame(f + 1)
            slam.add_stored_frame(f, stamps[f], f)
        if not distributed:
            return None
        publish(0, slam, stamps[f])
        with latest_lock:
            rows = latest.copy()
        return exchange.post_rows(rows)

    # Warm-up frames carry HIP events around every scope: that gives the per-kernel table and names the dominant
    # kernel.  Events cost a few microseconds each on a launch-bound path, so the timed region only keeps them on
    # that one kernel, one launch in four (its launches are all counted): the roofline figure is measured live
    # over the timed region without slowing what is being timed.
    if not args.no_profile:
        ctx.profile(True)
        ctx.profile_reset()
    for t in followers:
        t.start()
    for f in range(args.warmup):
        h = step(f)
        if h is not None:
            h.wait()
    warm_kernels, dominant = [], None
    if not args.no_profile and args.warmup > 0:
        warm_kernels = ctx.profile_stats()
        if warm_kernels:
            dominant = max(on_critical_path(warm_kernels), key=lambda k: k["total_ms"])["name"]

    stats_acc = np.zeros(16)
    if not args.no_profile:
        ctx.profile_reset()
        if dominant is not None and not args.profile_all:
            ctx.profile_select(dominant, 4)
        else:
            ctx.profile(True)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.sync()
    gate.wait()  # the other sequences of this rank have finished their warm-up frames too
    t0 = time.perf_counter()
    pending = None
    for f in range(args.warmup, total):
        h = step(f)
        if pending is not None:
            pending.wait()
        pending = h
        stats_acc += slam.stats()
    if pending is not None:
        pending.wait()
    ctx.sync()
    gate.wait()  # ... and their timed frames
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    for t in followers:
        t.join()
    for o, _ in others:
        o.close()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernels = [] if args.no_profile else ctx.profile_stats()
    ctx.profile(False)

    if rank == 0:
        out = {
            "metric": "LiDAR frames/sec (AddFrame end-to-end), VLS-128 scan" if args.model == 128 else f"LiDAR frames/sec (AddFrame end-to-end), model {args.model}",
            "value": world * per_gpu * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32+f64",
            "data": "synthetic",
            "config": {
                "workload": {128: "VLS-128", 64: "HDL-64", 16: "VLP-16"}.get(args.model, str(args.model))
                + " synthetic spinning scan, street canyon, 5 m/s, EgoMotion=MOTION_EXTRAPOLATION_AND_REGISTRATION, Undistortion=REFINED, library defaults",
                "points_per_frame": npts // (total * per_gpu),
                "sequences": world * per_gpu,
                "parallelism": f"{per_gpu} sequence{'s' if per_gpu > 1 else ''} per GPU x {world}, {'RCCL' if args.backend == 'nccl' else 'gloo (rehearsal, ranks share devices)'} all-gather of poses",
                "frames_resident_in_hbm": not args.host_frames,
                "lookahead_extraction": lookahead,
                "host_threads_on_numa_node": numa_node if numa_node is not None and numa_node >= 0 else None,
            },
        }
        n = args.steps
        icp_iters = max(stats_acc[9] + stats_acc[10], 1)
        out["ms_per_icp_iter"] = 1e3 * (stats_acc[2] + stats_acc[3] + stats_acc[4] + stats_acc[5]) / icp_iters
        out["stage_ms_per_frame"] = {
            k: 1e3 * stats_acc[i] / n
            for i, k in enumerate(["total", "extract", "ego_icp", "ego_lm", "loc_icp", "loc_lm", "undistort", "submap", "maps"])
        }
        try:
            out["submap_speculation_hits_per_frame"] = slam.get_param("SubMapSpeculationHits") / (args.steps + args.warmup)
        except Exception:
            pass
        out["stage_ms_per_frame"]["maps_wait"] = 1e3 * stats_acc[14] / n
        out["stage_ms_per_frame"]["maps_async"] = 1e3 * stats_acc[15] / n
        table, table_frames = (kernels, n) if (args.profile_all or not warm_kernels) else (warm_kernels, max(args.warmup, 1))
        if kernels:
            dom = max(on_critical_path(kernels), key=lambda k: k["total_ms"])
            ach = dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9 if dom["total_ms"] > 0 else 0.0
            out["roofline"] = {
                "bound": "hbm",
                "kernel": dom["name"],
                "achieved": ach,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": pmc_traffic(dom["name"], args.model),
                "avg_launch_us": 1e3 * dom["total_ms"] / max(dom["launches"], 1),
                "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1),
            }
            out["kernels_from"] = "timed region" if table is kernels else "warm-up frames (every scope timed)"
            out["kernels"] = {
                k["name"]: {
                    "launches_per_frame": k["launches"] / table_frames,
                    "us_per_launch": 1e3 * k["total_ms"] / max(k["launches"], 1),
                    "GBps": (k["bytes"] / (k["total_ms"] * 1e-3) / 1e9) if k["total_ms"] > 0 else 0.0,
                }
                for k in sorted(table, key=lambda k: -k["total_ms"])
            }
        if world == 1 and args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(args, seed)
    slam.close()
    if rank == 0:
        if world == 1 and per_gpu == 1 and args.batch_sequences > 1 and not args.host_frames:
            out["batch_replay"] = batch_replay(args, local_rank)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def on_critical_path(scopes):
    """The scopes that run on the context's own stream.  "*_ahead" scopes are enqueued on the look-ahead stream beside
    the registration: their events also bracket the time they wait for that stream, and they are off the frame's
    critical path by construction -- they are listed in the table but never named the dominant kernel."""
    kept = [k for k in scopes if not k["name"].endswith("_ahead")]
    return kept or scopes


def batch_replay(args, device):
    """Not the headline: the same GPU replaying several independent sequences side by side (seeds 1000, 1001, ...),
    same frames per sequence as the timed region above, aggregate frames/s between a common start and the last end."""
    from lidarslam_amd.replay import ConcurrentReplay, sequence_seed

    n = args.batch_sequences
    rep = ConcurrentReplay(device, args.model, [sequence_seed(s) for s in range(n)], args.warmup + args.steps, lookahead=not args.no_lookahead, EgoMotion=3)
    for kv in args.param:
        name, value = kv.split("=")
        for s in rep.slams:
            s.set_param(name, float(value))
    fps = rep.run(args.warmup)
    rep.close()
    return {"sequences_on_one_gpu": n, "value": fps, "unit": "frames/s", "per_sequence": fps / n, "steps": args.steps, "warmup": args.warmup}


# profiling scope -> device kernel, for looking a scope up in the committed PMC table
SCOPE_KERNEL = {
    "accumulate_jac": ["k_accumulate"], "accumulate_cost": ["k_accumulate"],
    "knn_fine_edge": ["k_knn_first<8, 16, 2>", "k_knn_first<16, 16, 2>"], "knn_fine_plane": ["k_knn_first<5, 8, 2>"],
    "knn_coarse_edge": ["k_knn_second<8>", "k_knn_second<16>"], "knn_coarse_plane": ["k_knn_second<5>"],
    "label_nms": ["k_label"], "curvature": ["k_curvature<4>"],
}


STREAMING_SCOPES = {"accumulate_jac", "accumulate_cost", "curvature", "label_nms"}


def pmc_traffic(scope, model):
    """HBM bytes per launch of the roofline kernel from the newest committed rocprofv3 --pmc table
    (profiles/rNN_vls128_pmc_traffic.json, made by scripts/round_measure.sh + collect_profiles.py from
    separate FETCH_SIZE and WRITE_SIZE passes of this same command).  The counters cannot be read from
    inside the process, so this is the figure of the profiled run, or None when no table matches."""
    import glob

    if model != 128 or scope not in SCOPE_KERNEL:
        return None
    tabs = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_vls128_pmc_traffic.json")))
    if not tabs:
        return None
    tab = json.load(open(tabs[-1]))
    rows = [tab[k] for k in SCOPE_KERNEL[scope] if k in tab]
    calls = sum(r["calls"] for r in rows)
    if not rows or calls == 0:
        return None
    # launch-weighted over the kernel instances behind the scope.  gfx950 FETCH_SIZE reports half the bytes of
    # wide coalesced streaming reads (MI355X_MICROARCH.md, HBM): doubled for the kernels that stream their input
    # (the normal-equation and per-point kernels; 2 x FETCH then lands within 10 % of the algorithmic bytes), raw for
    # the gather kernels (kNN), whose access width the guide calls uncalibrated.
    fetch_scale = 2.0 if scope in STREAMING_SCOPES else 1.0
    return sum((fetch_scale * r["fetch_kib"] + r["write_kib"]) * 1024.0 * r["calls"] for r in rows) / calls


def cpu_baseline(args, seed):
    """The CPU oracle (restatement of the reference algorithm, OpenMP over rings / keypoints like the
    reference) on the first frames of the same sequence, timed on this box's host cores."""
    from oracle import oracle as O

    import lidarslam_amd as L

    # the GPU box gives one GPU job a share of 16 host cores, whatever the affinity mask says
    threads = args.cpu_threads or min(16, len(os.sched_getaffinity(0)))
    s = O.Slam(EgoMotion=3, NbThreads=threads)
    times = []
    skip = 2  # the first frames build the map from nothing and are not representative
    for f in range(args.cpu_frames + skip):
        pts, stamp = L.synth_frame(args.model, seed, f)
        t = time.perf_counter()
        s.add_frame(pts, stamp, f)
        times.append(time.perf_counter() - t)
    t = np.array(times[skip:])
    return {
        "value": float(len(t) / t.sum()),
        "unit": "frames/s",
        "cores": threads,
        "kind": "port",
        "sample": f"frames {skip}..{skip + len(t) - 1} of the same sequence (seed {seed}), median {1e3 * float(np.median(t)):.1f} ms/frame",
    }


if __name__ == "__main__":
    main()
