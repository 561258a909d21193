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
    """All-gathers the latest pose of every sequence.  `device` is 'cuda' (RCCL) or 'cpu' (gloo)."""

    def __init__(self, world, device="cuda"):
        import torch

        self.torch = torch
        self.world = world
        self.device = device
        self.mine = torch.zeros(POSE_WORDS, dtype=torch.float64, device=device)
        self.table = torch.zeros(POSE_WORDS * world, dtype=torch.float64, device=device)
        self.host = torch.zeros(POSE_WORDS, dtype=torch.float64)
        self.side = None
        if device == "cuda":
            self.host = self.host.pin_memory()
            self.side = torch.cuda.Stream()

    def post(self, pose4x4, stamp_s):
        """Starts the exchange of this rank's pose; returns a work handle (None when world == 1)."""
        if self.world == 1:
            self.table[:16] = self.torch.from_numpy(np.ascontiguousarray(pose4x4, np.float64).reshape(16))
            self.table[16] = stamp_s
            return None
        import torch.distributed as dist

        self.host[:16] = self.torch.from_numpy(np.ascontiguousarray(pose4x4, np.float64).reshape(16))
        self.host[16] = stamp_s
        if self.side is not None:
            with self.torch.cuda.stream(self.side):
                self.mine.copy_(self.host, non_blocking=True)
                return dist.all_gather_into_tensor(self.table, self.mine, async_op=True)
        self.mine.copy_(self.host)
        return dist.all_gather_into_tensor(self.table, self.mine, async_op=True)

    def poses(self):
        """(world, 4, 4) poses and (world,) stamps of the last completed exchange."""
        t = self.table.detach().cpu().numpy().reshape(self.world, POSE_WORDS)
        return t[:, :16].reshape(self.world, 4, 4).copy(), t[:, 16].copy()
