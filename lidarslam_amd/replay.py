"""Batch replay of independent sequences across GPUs.

The path shards by SEQUENCE: frame k of a sequence needs the pose, the maps and the previous
keypoints of frame k-1, so one sequence lives on one GPU; sequences are independent, so rank r
replays sequence r and nothing is exchanged on the critical path.  The only collective is the
pose table: after every step each rank contributes its 4x4 pose + stamp (17 doubles) to an
all-gather (RCCL over xGMI on the GPUs, gloo on CPU for the tests), issued asynchronously on a
side stream, never on the ICP stream.
"""
import numpy as np

POSE_WORDS = 17  # row-major 4x4 + stamp [s]


def sequence_seed(rank, base=1000):
    """Deterministic generator seed of the sequence a rank replays (1000 + sequence id)."""
    return base + rank


class PoseExchange:
    """All-gathers the latest pose of every sequence.  `device` is 'cuda' (RCCL) or 'cpu' (gloo).

    Exchanges may overlap the caller's next steps, and the caller may post step f + 1 before it waits for step f: every
    exchange in flight owns one of `depth` buffer sets (pinned staging row, device row, gathered table), and a set is
    only reused after the exchange that used it last has completed (`post_rows` waits for it itself if the caller did
    not) -- no copy or collective ever reads or writes a buffer another one still uses."""

    def __init__(self, world, device="cuda", per_rank=1, depth=2, always_collective=False):
        import torch

        self.always_collective = always_collective  # world == 1 normally needs no process group: tests force the collective

        self.torch = torch
        self.world = world
        self.device = device
        self.per_rank = per_rank  # sequences replayed side by side on one GPU: their poses travel together
        n = POSE_WORDS * per_rank
        self.depth = max(int(depth), 1)
        self.mine = [torch.zeros(n, dtype=torch.float64, device=device) for _ in range(self.depth)]
        self.tables = [torch.zeros(n * world, dtype=torch.float64, device=device) for _ in range(self.depth)]
        self.hosts = [torch.zeros(n, dtype=torch.float64) for _ in range(self.depth)]
        self.inflight = [None] * self.depth
        self.posted = 0
        self.table = self.tables[0].clone()  # the table of the last completed exchange (its own copy, never a buffer in rotation)
        self.side = None
        if device == "cuda":
            self.hosts = [h.pin_memory() for h in self.hosts]
            self.side = torch.cuda.Stream()

    def post(self, pose4x4, stamp_s):
        """Starts the exchange of this rank's pose; returns a work handle (None when world == 1)."""
        row = np.empty(POSE_WORDS)
        row[:16] = np.ascontiguousarray(pose4x4, np.float64).reshape(16)
        row[16] = stamp_s
        return self.post_rows(row)

    def post_rows(self, rows):
        """Same for the (per_rank, 17) latest pose rows of all the sequences this rank replays."""
        rows = self.torch.from_numpy(np.ascontiguousarray(rows, np.float64).reshape(POSE_WORDS * self.per_rank))
        slot = self.posted % self.depth
        self.posted += 1
        if self.inflight[slot] is not None:
            self.inflight[slot].wait()  # the exchange that used this buffer set last
        host, mine, table = self.hosts[slot], self.mine[slot], self.tables[slot]
        if self.world == 1 and not self.always_collective:
            table.copy_(rows)
            self.table = table.clone()
            return None
        import torch.distributed as dist

        host.copy_(rows)
        done = None
        if self.side is not None:
            with self.torch.cuda.stream(self.side):
                mine.copy_(host, non_blocking=True)
                work = dist.all_gather_into_tensor(table, mine, async_op=True)
                work.wait()  # RCCL: orders the side stream behind the collective, the host does not block
                done = self.torch.cuda.Event()
                done.record(self.side)  # copy and collective of THIS exchange are over when the event is
        else:
            mine.copy_(host)
            work = dist.all_gather_into_tensor(table, mine, async_op=True)
        handle = _Exchange(self, slot, work, done)
        self.inflight[slot] = handle
        return handle

    def poses(self):
        """(sequences, 4, 4) poses and (sequences,) stamps of the last completed exchange, rank-major."""
        n = self.world * self.per_rank
        t = self.table.detach().cpu().numpy().reshape(n, POSE_WORDS)
        return t[:, :16].reshape(n, 4, 4).copy(), t[:, 16].copy()


class _Exchange:
    """Handle of one exchange: wait() completes it (once) and makes its table the current one."""

    def __init__(self, owner, slot, work, done=None):
        self.owner, self.slot, self.work, self.done = owner, slot, work, done

    def wait(self):
        if self.work is None:
            return
        if self.done is not None:
            self.done.synchronize()  # this exchange's copy and collective on the side stream, nothing later
        else:
            self.work.wait()
        self.work = None
        # a copy: the buffer set goes back into rotation and a later post may gather into it while the caller reads
        self.owner.table = self.owner.tables[self.slot].clone()
        if self.owner.inflight[self.slot] is self:
            self.owner.inflight[self.slot] = None


class ConcurrentReplay:
    """Replays several independent sequences side by side on ONE device: a Slam, a context and a host thread
    each.  One sequence is a chain of small dependent kernels and host hand-overs that leaves most of the chip
    idle; independent sequences fill it.  Every thread drives its own handle, the library keeps no
    state outside of the handles, and the results are those of a lone run bit for bit (tests/test_gpu_pipeline.py)."""

    def __init__(self, device, model, seeds, frames, lookahead=True, **params):
        import lidarslam_amd as L

        # Where the rolling maps live ("MapsOnDevice", the caller's choice when given).  One sequence: on the device --
        # nothing of the map crosses the bus and the frame does not wait for host threads (measured on MI355X, VLS-128:
        # 1 124 against 707 frames/s).  Several sequences side by side: on the host -- a process has 4 hardware queues,
        # the maps' chains of small kernels compete for them with the ICP kernels of the other sequences (8 sequences:
        # 1810 against 2210 frames/s), where the host cores are there (2.6 per sequence against 1.4).
        params.setdefault("MapsOnDevice", 1 if len(seeds) == 1 else 0)
        # ICP loops enqueued whole, every solve leaving the next iteration's pose on the device (links: nothing spins on a
        # hardware queue, unlike the gates of ICPAhead = 1, which side by side held up the other sequences' kernels)
        params.setdefault("ICPAhead", 2)
        # worker threads woken ahead of their jobs poll for them: worth a core for one or two sequences, not for eight
        params.setdefault("WorkerPrewake", 1 if len(seeds) <= 2 else 0)
        self.frames = frames
        self.lookahead = lookahead  # extract frame f + 1 beside the registration of frame f (same results)
        self.slams, self.stamps = [], []
        for seed in seeds:
            slam = L.Slam(device, **params)
            st = []
            for f in range(frames):
                pts, stamp = L.synth_frame(model, seed, f)
                slam.store_frame(f, pts)
                st.append(stamp)
            self.slams.append(slam)
            self.stamps.append(st)
        self.poses = np.zeros((len(seeds), frames, 4, 4))

    def run(self, warmup):
        """Untimed frames [0, warmup), then the rest timed from a common start to the last sequence's end.
        Returns the aggregate frames/s."""
        import threading
        import time

        if not 0 <= warmup < self.frames:
            raise ValueError("warmup must leave at least one timed frame")
        n = len(self.slams)
        gate = threading.Barrier(n + 1)
        errors = []

        def worker(s):
            try:
                for f in range(self.frames):
                    if f == warmup:
                        gate.wait()
                    if self.lookahead and f + 1 < self.frames:
                        self.slams[s].hint_next_stored_frame(f + 1)
                    self.slams[s].add_stored_frame(f, self.stamps[s][f], f)
                    self.poses[s, f] = self.slams[s].world_transform()
                self.slams[s].context().sync()
            except Exception as e:  # a failed sequence must not leave the others waiting
                errors.append(e)
                gate.abort()
                return
            gate.wait()

        threads = [threading.Thread(target=worker, args=(s,)) for s in range(n)]
        for t in threads:
            t.start()
        try:
            gate.wait()
            t0 = time.perf_counter()
            gate.wait()
            dt = time.perf_counter() - t0
        except threading.BrokenBarrierError:
            dt = float("nan")
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return n * (self.frames - warmup) / dt

    def close(self):
        for s in self.slams:
            s.close()
        self.slams = []
