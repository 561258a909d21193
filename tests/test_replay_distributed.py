"""Multi-process path of the batch replay on CPU: world_size 2, gloo (the GPU runs use the same
code with RCCL).  Each rank replays its own sequence with the CPU oracle standing in for the GPU
pipeline (this test is about sharding and the pose all-gather, not about kernels)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from lidarslam_amd import synth_frame
    from lidarslam_amd.replay import PoseExchange, sequence_seed
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    seed = sequence_seed(rank)
    slam = O.Slam(EgoMotion=3)
    ex = PoseExchange(world, device="cpu")
    tables = []
    pending = None
    for f in range(3):
        pts, stamp = synth_frame(8, seed, f)
        slam.add_frame(pts, stamp, f)
        h = ex.post(slam.world_transform(), stamp * 1e-6)
        if pending is not None:
            pending.wait()
        pending = h
    pending.wait()
    poses, stamps = ex.poses()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), poses=poses, stamps=stamps, mine=slam.world_transform())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pose_exchange(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # every rank holds the same table, row r is rank r's own pose, sequences differ
    assert np.array_equal(r0["poses"], r1["poses"]) and np.array_equal(r0["stamps"], r1["stamps"])
    assert np.array_equal(r0["poses"][0], r0["mine"]) and np.array_equal(r1["poses"][1], r1["mine"])
    assert not np.array_equal(r0["mine"], r1["mine"])
    assert np.allclose(r0["stamps"], 0.3)


def _worker_two_per_rank(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from lidarslam_amd.replay import POSE_WORDS, PoseExchange

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = PoseExchange(world, device="cpu", per_rank=2)
    rows = np.zeros((2, POSE_WORDS))
    for s in range(2):
        T = np.eye(4)
        T[0, 3] = 10 * rank + s
        rows[s, :16], rows[s, 16] = T.reshape(16), 0.1 * (2 * rank + s)
    ex.post_rows(rows).wait()
    poses, stamps = ex.poses()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), poses=poses, stamps=stamps)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_two_sequences_each(tmp_path):
    """--sequences-per-gpu 2: the table is rank-major, (world x per_rank) rows"""
    import torch.multiprocessing as mp

    mp.spawn(_worker_two_per_rank, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["poses"], r1["poses"]) and np.array_equal(r0["stamps"], r1["stamps"])
    assert r0["poses"].shape == (4, 4, 4)
    assert r0["poses"][:, 0, 3].tolist() == [0.0, 1.0, 10.0, 11.0]
    assert np.allclose(r0["stamps"], [0.0, 0.1, 0.2, 0.3])


def test_sequence_sharding_is_by_rank():
    sys.path.insert(0, ROOT)
    from lidarslam_amd.replay import sequence_seed

    assert [sequence_seed(r) for r in range(8)] == list(range(1000, 1008))


def test_single_rank_exchange_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from lidarslam_amd.replay import PoseExchange

    ex = PoseExchange(1, device="cpu")
    T = np.eye(4)
    T[0, 3] = 2.5
    assert ex.post(T, 0.7) is None
    poses, stamps = ex.poses()
    assert np.array_equal(poses[0], T) and stamps[0] == 0.7
