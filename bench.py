#!/usr/bin/env python3
"""bench.py -- LiDAR frames/s of Slam::AddFrame (end to end, from a HOST cloud) on synthetic VLS-128 sequences.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one AddFrame (upload of the scan over PCIe + keypoint extraction + ego-motion ICP + localization ICP + map
update) on one VLS-128-shaped scan (128 x 2048 firings, ~260k points) that is handed over as a host buffer, the way
Slam::AddFrames receives it (slam_lib/src/Slam.cxx:230-237): the upload is INSIDE the timed region.  It is a replay:
the caller holds the next cloud while the current one is registered and announces it (lsa_slam_hint_next_frame), so
the upload (pinned staging, copy stream) and the extraction of frame f + 1 run beside the registration of frame f.
Two more legs are reported beside the headline (N = 1): the same replay from scans already resident in HBM
("replay_resident") and the strictly causal one -- host clouds, nothing announced ahead ("causal_no_lookahead").
Every rank replays its own independent sequence (seed 1000 + rank) on its own GPU -- the path shards by sequence
(SURVEY.md 8e), scaling is weak, the only exchange is the all-gather of the 4x4 pose + stamp of every sequence
after each step (RCCL when there is more than one rank).  Without a launcher, `--gpus N` with N > 1 starts the N
ranks itself.  Rank 0 prints ONE JSON line with two extra objects:
  roofline      achieved algorithmic GB/s of the dominant kernel FAMILY (summed launch time; HIP events on the
                context's own stream, live during the timed region) against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle ("port": restatement of the reference algorithm, OpenMP where the reference has it)
                on the very frames of the timed region, on the host cores
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# profiling scopes -> kernel families (a family is what one step of the algorithm launches, whatever its kernels are)
FAMILIES = {
    "match": ("match_search", "match_model", "knn_fine_", "knn_coarse_", "model_"),  # KeypointsMatcher::BuildMatchResiduals
    "lm_solve": ("lm_solve", "accumulate_"),                                       # LocalOptimizer::Solve
    "extract": ("ring_bucket", "invalidate", "curvature", "label_nms", "compact"),  # SpinningSensorKeypointExtractor
    "target_grid": ("target_grid_build",),
}
FAMILY_PREFIX = {"match": "match_", "lm_solve": "lm_solve"}  # what lsa_profile_select takes for the timed region


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", type=int, default=128, help="16 VLP-16, 64 HDL-64, 128 VLS-128 (headline)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores available to this process (at most 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--param", action="append", default=[], help="Slam parameter override NAME=VALUE (reference setter names)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events at all")
    ap.add_argument("--resident", action="store_true", help="headline leg from scans already resident in HBM (never the default)")
    ap.add_argument("--causal", action="store_true", help="headline leg without announcing the next cloud (no look-ahead)")
    ap.add_argument("--no-extra-legs", action="store_true", help="N=1: skip the replay_resident / causal_no_lookahead / batch_replay legs")
    ap.add_argument("--sequences-per-gpu", type=int, default=1, help="independent sequences replayed side by side on every GPU (the headline is 1: BASELINE.json shards 1 per GPU)")
    ap.add_argument("--batch-sequences", type=int, default=8, help="N=1 only: extra leg with this many sequences side by side on the GPU, reported as batch_replay (0 disables)")
    ap.add_argument("--no-numa-bind", action="store_true", help="leave the host threads wherever the scheduler puts them (default: on the GPU's NUMA node)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl (= RCCL over xGMI, the real run); gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks then share devices)")
    ap.add_argument("--profile-all", action="store_true", help="time every scope in the timed region too (costs ~10 %% of the frame rate)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh processes (this process has not touched
    the GPU), hand rank 0's JSON line through, fail if any rank fails."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0's stdout is THE line; what the other ranks print goes to stderr, labelled, so that a rank that fails is seen
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE if r else None))
    outs = [None] * len(procs)

    def drain(r):
        outs[r] = procs[r].communicate()

    readers = [threading.Thread(target=drain, args=(r,)) for r in range(len(procs))]
    for t in readers:
        t.start()
    for t in readers:
        t.join()
    rcs = [p.returncode for p in procs]
    for r in range(1, len(procs)):
        for stream in outs[r]:
            text = (stream or b"").decode(errors="replace").strip()
            if text and (rcs[r] != 0 or os.environ.get("LSA_BENCH_VERBOSE")):
                sys.stderr.write("".join(f"[rank {r}] {line}\n" for line in text.splitlines()[-40:]))
    sys.stdout.write(outs[0][0].decode())
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit(f"ranks exited with {rcs} (rank r's output above, prefixed [rank r])")


def family_of(scope):
    for fam, prefixes in FAMILIES.items():
        if scope.startswith(prefixes):
            return fam
    return None


def by_family(scopes):
    """{family: {"total_ms", "bytes", "launches" (of its first scope = steps of the algorithm), "scopes"}} over the
    scopes of the context's own stream ("*_ahead" scopes run on the look-ahead stream, off the critical path)."""
    fams = {}
    for k in scopes:
        fam = family_of(k["name"])
        if fam is None or k["name"].endswith("_ahead"):
            continue
        f = fams.setdefault(fam, {"total_ms": 0.0, "bytes": 0.0, "launches": 0, "scopes": []})
        f["total_ms"] += k["total_ms"]
        f["bytes"] += k["bytes"]
        f["launches"] = max(f["launches"], k["launches"])
        f["scopes"].append(k["name"])
    return fams


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist

    import lidarslam_amd as L
    from lidarslam_amd.replay import PoseExchange, sequence_seed
    from lidarslam_amd._native import ptr

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

    total = args.warmup + args.steps
    per_gpu = max(args.sequences_per_gpu, 1)
    mode = "resident" if args.resident else ("causal" if args.causal else "announced")

    def make_slam():
        s = L.Slam(local_rank, EgoMotion=3)  # MOTION_EXTRAPOLATION_AND_REGISTRATION: the default mode skips the ego-motion ICP
        for kv in args.param:
            name, value = kv.split("=")
            s.set_param(name, float(value))
        return s

    # ---- inputs: one independent sequence per (rank, sequence), generated on the host, kept as host clouds
    sequences = []
    npts = 0
    for s in range(per_gpu):
        seed = sequence_seed(rank * per_gpu + s)
        frames = [L.synth_frame(args.model, seed, f) for f in range(total)]
        npts += sum(p.size for p, _ in frames)
        sequences.append((seed, frames))

    class Replay:
        """One sequence on one Slam handle, in one of the three modes."""

        def __init__(self, frames, mode):
            self.frames, self.mode = frames, mode
            self.slam = make_slam()
            self.clouds = [L.Slam.cloud_pointer(pts) for pts, _ in frames]  # the harness's own per-call work stays out of the timed loop
            if mode == "resident":
                for f, (pts, _) in enumerate(frames):
                    self.slam.store_frame(f, pts)

        def step(self, f, announce=True):
            pts, stamp = self.frames[f]
            nxt = announce and f + 1 < len(self.frames)
            if self.mode == "resident":
                if nxt:
                    self.slam.hint_next_stored_frame(f + 1)
                self.slam.add_stored_frame(f, stamp, f)
            else:
                if self.mode == "announced" and nxt:
                    self.slam.hint_next_frame_at(self.clouds[f + 1])  # the replay holds the next cloud already
                self.slam.add_frame_at(self.clouds[f], stamp, f)

    main_replay = Replay(sequences[0][1], mode)
    slam = main_replay.slam
    ctx = slam.context()

    # further sequences of this rank (--sequences-per-gpu): a Slam, a context and a host thread each, free-running
    # beside the first one between the same start and end barriers; their latest poses travel with the first one's
    latest = np.zeros((per_gpu, 17))
    latest_lock = threading.Lock()
    others, gate = [Replay(fr, mode) for _, fr in sequences[1:]], threading.Barrier(per_gpu)

    def publish(s, o, stamp):
        with latest_lock:
            latest[s, :16] = np.asarray(o.world_transform()).reshape(16)
            latest[s, 16] = stamp * 1e-6

    def follow(s, rp):
        try:
            for f in range(total):
                if f == args.warmup:
                    gate.wait()
                rp.step(f, announce=f != args.warmup - 1)
                publish(s, rp.slam, rp.frames[f][1])
            rp.slam.context().sync()
        except BaseException:
            gate.abort()  # the first sequence then stops with BrokenBarrierError instead of waiting for ever
            raise
        gate.wait()

    followers = [threading.Thread(target=follow, args=(s + 1, rp), daemon=True) for s, rp in enumerate(others)]

    # pose table of the north star: every rank ends up with every sequence's pose (a collective only when world > 1)
    exchange = PoseExchange(world, device="cuda" if args.backend == "nccl" else "cpu", per_rank=per_gpu)

    def step(f):
        # the last warm-up frame announces nothing: the first timed frame's upload and extraction are inside the timed
        # region like every other's (the last timed frame announces the frame behind it when there is one -- K uploads
        # and K extractions are timed either way)
        main_replay.step(f, announce=f != args.warmup - 1)
        if not distributed:
            return None
        publish(0, slam, main_replay.frames[f][1])
        with latest_lock:
            rows = latest.copy()
        return exchange.post_rows(rows)

    # Warm-up frames carry HIP events around every scope: that gives the per-kernel table and names the dominant
    # kernel FAMILY (largest summed launch time).  Events cost a few microseconds each on this launch-bound path, so
    # the timed region only keeps them on that family, one step in four (all its launches are counted): the roofline
    # figure is measured live over the timed region without slowing what is being timed.
    if not args.no_profile:
        ctx.profile(True)
        ctx.profile_reset()
    for t in followers:
        t.start()
    for f in range(args.warmup):
        if f == 2 and args.warmup > 4 and not args.no_profile:
            # the first launch of every kernel loads its code (milliseconds, once): those frames must not name the family
            ctx.sync()
            ctx.profile_reset()
        h = step(f)
        if h is not None:
            h.wait()
    warm_kernels, dominant = [], None
    if not args.no_profile and args.warmup > 0:
        warm_kernels = ctx.profile_stats()
        fams = by_family(warm_kernels)
        if fams:
            dominant = max(fams, key=lambda k: fams[k]["total_ms"])

    stats_acc, stats_now = np.zeros(16), np.zeros(16)
    stats_ptr = ptr(stats_now)
    if not args.no_profile:
        ctx.profile_reset()
        if dominant in FAMILY_PREFIX and not args.profile_all:
            ctx.profile_select(FAMILY_PREFIX[dominant], 4)
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
        slam.stats_into(stats_ptr)
        stats_acc += stats_now
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
    others_alive = others  # closed once their counters have been read
    per_rank = None
    if distributed:
        dev = "cuda" if args.backend == "nccl" else "cpu"
        mine = torch.tensor([elapsed, float(slam.get_param("DeviceSolveFallbacks")), float(slam.get_param("IcpGateTimeouts"))], dtype=torch.float64, device=dev)
        table = torch.zeros(world * 3, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(table, mine)  # every rank's own clock and fall-back counters: rank 0 reports them
        per_rank = table.cpu().numpy().reshape(world, 3)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ctx_event_overhead = ctx.profile_event_overhead_us()
    kernels = [] if args.no_profile else ctx.profile_stats()
    ctx.profile(False)
    extra = {"uploads_taken_over": slam.get_param("UploadsAdopted"), "extractions_taken_over": slam.get_param("LookaheadAdopted"),
             "device_solve_fallbacks": slam.get_param("DeviceSolveFallbacks") + sum(rp.slam.get_param("DeviceSolveFallbacks") for rp in others_alive),
             "icp_gate_timeouts": slam.get_param("IcpGateTimeouts") + sum(rp.slam.get_param("IcpGateTimeouts") for rp in others_alive)}
    for rp in others_alive:
        rp.slam.close()
    if per_rank is not None:
        extra["ranks"] = {"backend": args.backend, "world_size": world,
                          "frames_per_s_per_rank": [per_gpu * args.steps / float(r[0]) for r in per_rank],
                          "device_solve_fallbacks_per_rank": [int(r[1]) for r in per_rank], "icp_gate_timeouts_per_rank": [int(r[2]) for r in per_rank]}

    if rank == 0:
        name = {128: "VLS-128", 64: "HDL-64", 16: "VLP-16"}.get(args.model, str(args.model))
        exchange_txt = "" if world == 1 else (", RCCL all-gather of poses" if args.backend == "nccl" else ", gloo all-gather of poses (rehearsal, ranks share devices)")
        out = {
            "metric": f"LiDAR frames/sec (AddFrame end-to-end), {name} scan",
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
                "workload": name + " synthetic spinning scan, street canyon, 5 m/s, EgoMotion=MOTION_EXTRAPOLATION_AND_REGISTRATION, Undistortion=REFINED, library defaults",
                "points_per_frame": npts // (total * per_gpu),
                "sequences": world * per_gpu,
                "parallelism": f"{per_gpu} sequence{'s' if per_gpu > 1 else ''} per GPU x {world}{exchange_txt}",
                "frames_resident_in_hbm": mode == "resident",
                "frames_from": {"announced": "host clouds, next cloud announced to AddFrame's caller API (offline replay): upload + extraction overlap the previous frame",
                                "causal": "host clouds, nothing announced ahead", "resident": "frame store in HBM, look-ahead extraction"}[mode],
                "upload_in_timed_region": mode != "resident",
                "host_threads_on_numa_node": numa_node if numa_node is not None and numa_node >= 0 else None,
            },
        }
        out["config"].update(extra)
        n = args.steps
        icp_iters = max(stats_acc[9] + stats_acc[10], 1)
        out["ms_per_icp_iter"] = 1e3 * (stats_acc[2] + stats_acc[3] + stats_acc[4] + stats_acc[5]) / icp_iters
        out["stage_ms_per_frame"] = {
            k: 1e3 * stats_acc[i] / n
            for i, k in enumerate(["total", "extract", "ego_icp", "ego_lm", "loc_icp", "loc_lm", "undistort", "submap", "maps"])
        }
        out["stage_ms_per_frame"]["maps_wait"] = 1e3 * stats_acc[14] / n
        out["stage_ms_per_frame"]["maps_async"] = 1e3 * stats_acc[15] / n
        table, table_frames = (kernels, n) if (args.profile_all or not warm_kernels) else (warm_kernels, max(args.warmup - 2 if args.warmup > 4 else args.warmup, 1))
        if kernels:
            fams = by_family(kernels)
            fam = dominant if dominant in fams else max(fams, key=lambda k: fams[k]["total_ms"])
            dom = fams[fam]
            ach = dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9 if dom["total_ms"] > 0 else 0.0
            out["roofline"] = {
                "bound": "hbm",
                "kernel": fam,
                "scopes": dom["scopes"],
                "achieved": ach,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": pmc_traffic(fam, args.model),
                "avg_launch_us": 1e3 * dom["total_ms"] / max(dom["launches"], 1),
                "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1),
                "launches_per_frame": dom["launches"] / n,
                "chosen_by": "largest summed launch time by kernel family over the warm-up frames",
                "event_overhead_us": ctx_event_overhead,  # what a pair of events measures around nothing: taken off every launch
            }
            out["kernels_from"] = "timed region" if table is kernels else "warm-up frames (every scope timed)"
            out["kernels"] = {
                k["name"]: {
                    "family": family_of(k["name"]),
                    "launches_per_frame": k["launches"] / table_frames,
                    "us_per_launch": 1e3 * k["total_ms"] / max(k["launches"], 1),
                    "GBps": (k["bytes"] / (k["total_ms"] * 1e-3) / 1e9) if k["total_ms"] > 0 else 0.0,
                }
                for k in sorted(table, key=lambda k: -k["total_ms"])
            }
    slam.close()
    if rank == 0:
        if world == 1 and per_gpu == 1 and not args.no_extra_legs:
            for key, m in (("replay_resident", "resident"), ("causal_no_lookahead", "causal"), ("replay_announced", "announced")):
                if m != mode:
                    out[key] = extra_leg(Replay(sequences[0][1], m), args)
            # the causal leg again with the clouds in page-locked buffers, as a driver that registers its ring of scan
            # buffers once would hand them over (lsa_pin_host_memory): the upload is then one DMA at the bus rate
            for pts, _ in sequences[0][1]:
                L.Slam.pin_cloud(pts)
            out["causal_pinned_buffers"] = extra_leg(Replay(sequences[0][1], "causal"), args)
            out["causal_pinned_buffers"]["frames_from"] = "host clouds in page-locked buffers (registered once, outside the timed region), nothing announced ahead"
            for pts, _ in sequences[0][1]:
                L.Slam.unpin_cloud(pts)
            if args.batch_sequences > 1:
                out["batch_replay"] = batch_replay(args, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, sequences[0])
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def extra_leg(rp, args):
    """Not the headline: the same frames through another way of handing them over, timed like the headline."""
    for f in range(args.warmup):
        rp.step(f, announce=f != args.warmup - 1)
    rp.slam.context().sync()
    t0 = time.perf_counter()
    for f in range(args.warmup, args.warmup + args.steps):
        rp.step(f)
    rp.slam.context().sync()
    dt = time.perf_counter() - t0
    rp.slam.close()
    return {"value": args.steps / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / args.steps}


def batch_replay(args, device):
    """Not the headline: the same GPU replaying several independent sequences side by side (seeds 1000, 1001, ...),
    same frames per sequence as the timed region above, aggregate frames/s between a common start and the last end."""
    from lidarslam_amd.replay import ConcurrentReplay, sequence_seed

    n = args.batch_sequences
    rep = ConcurrentReplay(device, args.model, [sequence_seed(s) for s in range(n)], args.warmup + args.steps, lookahead=True, EgoMotion=3)
    for kv in args.param:
        name, value = kv.split("=")
        for s in rep.slams:
            s.set_param(name, float(value))
    fps = rep.run(args.warmup)
    maps = "device" if rep.slams and rep.slams[0].get_param("DeviceMapsInUse") else "host"
    fallbacks = [int(s.get_param("DeviceSolveFallbacks")) for s in rep.slams]  # of EVERY sequence: solves side by side compete for the CUs
    timeouts = [int(s.get_param("IcpGateTimeouts")) for s in rep.slams]
    rep.close()
    return {"sequences_on_one_gpu": n, "value": fps, "device_solve_fallbacks": fallbacks, "icp_gate_timeouts": timeouts, "unit": "frames/s", "per_sequence": fps / n, "steps": args.steps, "warmup": args.warmup,
            "frames_from": "frame store in HBM, look-ahead extraction", "rolling_maps": maps}


# kernel family -> device kernels, for looking the family up in the committed PMC table
FAMILY_KERNELS = {
    "match": ["k_search_all", "k_model_all", "k_knn_first", "k_knn_second", "k_model<"],
    "lm_solve": ["k_lm_solve", "k_accumulate"],
}


def pmc_traffic(family, model):
    """HBM bytes per launch of the roofline family from the newest committed rocprofv3 --pmc table
    (profiles/rNN_vls128_pmc_traffic.json, made by scripts/round_measure.sh + collect_profiles.py from separate
    FETCH_SIZE and WRITE_SIZE passes of this same command).  The counters cannot be read from inside the process, so
    this is the figure of the profiled run, or None when no table matches.  FETCH_SIZE is left as the counter reports
    it (gather kernels: the guide calls their access width uncalibrated)."""
    import glob

    if model != 128 or family not in FAMILY_KERNELS:
        return None
    tabs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_vls128_pmc_traffic.json")))
    if not tabs:
        return None
    tab = json.load(open(tabs[-1]))
    # per step of the algorithm = one launch of every KERNEL of the family; a kernel's template variants (k = 8 / 10 ...)
    # are alternatives, not further launches: mean over the variants by calls, then the kernels summed, each weighted by
    # how often it is launched relative to the most frequent one
    groups = {}
    for k, v in tab.items():
        for p in FAMILY_KERNELS[family]:
            if p in k:
                g = groups.setdefault(p, [0, 0.0])
                g[0] += v["calls"]
                g[1] += (v["fetch_kib"] + v["write_kib"]) * 1024.0 * v["calls"]
                break
    if not groups:
        return None
    most = max(g[0] for g in groups.values())
    if most == 0:
        return None
    return sum(g[1] for g in groups.values()) / most


def cpu_baseline(args, sequence):
    """The CPU oracle (restatement of the reference algorithm, OpenMP over rings / keypoints like the reference) on the
    frames of the timed region: it registers the warm-up frames too (the maps have to exist) but only the timed ones
    count, exactly as on the GPU.  SURVEY.md 8d asks for NbThreads = 1, 4 (the reference's recommended setting,
    ros_wrapping/lidar_slam/params/slam_config_outdoor.yaml:93) and all cores: the headline figure is the all-cores run over
    the whole timed region; 1 and 4 threads run a bounded sample of it (the first frames of the timed region, ~10 s each)."""
    import numpy as np

    from oracle import oracle as O

    seed, frames = sequence
    # the GPU box gives one GPU job a share of 16 host cores, whatever the affinity mask says
    cores = min(16, len(os.sched_getaffinity(0)))
    threads = args.cpu_threads or cores

    def run(nthreads, timed):
        s = O.Slam(EgoMotion=3, NbThreads=nthreads)
        times = []
        # the warm-up frames build the maps: always with all the cores (what is timed is the frames behind them)
        s.set_param("NbThreads", threads)
        for f, (pts, stamp) in enumerate(frames[: args.warmup + timed]):
            if f == args.warmup:
                s.set_param("NbThreads", nthreads)
            t = time.perf_counter()
            s.add_frame(pts, stamp, f)
            times.append(time.perf_counter() - t)
        return np.array(times[args.warmup:])

    t = run(threads, args.steps)
    out = {
        "value": float(len(t) / t.sum()),
        "unit": "frames/s",
        "cores": threads,
        "kind": "port",
        "sample": f"frames {args.warmup}..{args.warmup + len(t) - 1} of the same sequence (seed {seed}): the frames of the timed region, after the same {args.warmup} warm-up frames; median {1e3 * float(np.median(t)):.1f} ms/frame",
    }
    per_frame = float(np.median(t))
    by_threads = {str(threads): {"value": out["value"], "cores": threads, "frames": int(len(t))}}
    for n in (1, 4):
        if n >= threads or args.model != 128 and args.steps > 60:
            continue
        # a bounded sample: about 10 s of CPU work at this thread count (perfect scaling assumed for the estimate)
        k = int(max(3, min(args.steps, 10.0 / max(per_frame * threads / n * 0.6, 1e-3))))
        tn = run(n, k)
        by_threads[str(n)] = {"value": float(len(tn) / tn.sum()), "cores": n, "frames": int(len(tn))}
    out["by_threads"] = by_threads
    return out


if __name__ == "__main__":
    main()
