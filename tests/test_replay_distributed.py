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
    tables, mine = [], []
    pending = None
    for f in range(4):
        pts, stamp = synth_frame(8, seed, f)
        slam.add_frame(pts, stamp, f)
        mine.append(slam.world_transform())
        h = ex.post(slam.world_transform(), stamp * 1e-6)  # step f is posted before step f - 1 is waited for
        if pending is not None:
            pending.wait()
            tables.append(ex.poses())  # the table of step f - 1, complete, while step f is in flight
        pending = h
    pending.wait()
    tables.append(ex.poses())
    poses, stamps = ex.poses()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), poses=poses, stamps=stamps, mine=slam.world_transform(),
             step_poses=np.array([t[0] for t in tables]), step_stamps=np.array([t[1] for t in tables]), step_mine=np.array(mine))
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
    assert np.allclose(r0["stamps"], 0.4)
    # every step's table, not only the last: row r of step f is rank r's pose of step f on both ranks (an exchange
    # that reused a buffer of the one before it would mix steps)
    assert r0["step_poses"].shape == (4, 2, 4, 4)
    for f in range(4):
        for r, own in ((0, r0), (1, r1)):
            assert np.array_equal(r0["step_poses"][f, r], own["step_mine"][f]) and np.array_equal(r1["step_poses"][f, r], own["step_mine"][f])
        assert np.allclose(r0["step_stamps"][f], 0.1 * (f + 1)) and np.allclose(r1["step_stamps"][f], 0.1 * (f + 1))


def _worker_back_to_back(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from lidarslam_amd.replay import POSE_WORDS, PoseExchange

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = PoseExchange(world, device="cpu")
    seen = []
    handles = []
    for step in range(12):
        row = np.full(POSE_WORDS, 1000.0 * rank + step)
        handles.append(ex.post_rows(row))  # distinct rows, posted back to back, nobody waits in between
        if step >= 3:
            handles[step - 3].wait()  # long overdue: post_rows has completed it itself before reusing its buffers
    for h in handles:
        h.wait()
        seen.append(ex.table.clone().numpy().reshape(world, POSE_WORDS))
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array(seen))
    dist.barrier()
    dist.destroy_process_group()


def test_back_to_back_posts_never_share_a_buffer(tmp_path):
    """posts without waits in between: a buffer set is only reused once the exchange that used it last is over, so
    every completed table holds rows of ONE step, whole"""
    import torch.multiprocessing as mp

    mp.spawn(_worker_back_to_back, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1)
    final = r0[-1]
    assert np.all(final[0] == 11.0) and np.all(final[1] == 1011.0)
    for t in r0:  # each table: both rows from the same step, every word
        step = t[0, 0]
        assert np.all(t[0] == step) and np.all(t[1] == 1000.0 + step)


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


def _worker_rccl(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    from lidarslam_amd.replay import POSE_WORDS, PoseExchange

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    ex = PoseExchange(world, device="cuda", per_rank=2, always_collective=True)
    seen, handles = [], []
    for step in range(10):
        rows = np.stack([np.full(POSE_WORDS, 100.0 * s + step) for s in range(2)])
        handles.append(ex.post_rows(rows))  # pinned row -> device row -> RCCL all-gather, all on the side stream
        if step >= 1:
            handles[step - 1].wait()
            seen.append(ex.table.detach().cpu().numpy().reshape(2, POSE_WORDS).copy())
    handles[-1].wait()
    seen.append(ex.table.detach().cpu().numpy().reshape(2, POSE_WORDS).copy())
    np.save(os.path.join(out_dir, "rccl.npy"), np.array(seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_pose_exchange_runs_over_rccl_on_one_gpu(tmp_path):
    """the RCCL code path itself (backend nccl, device tensors, side stream, events) with a process group of one
    rank on the one GPU: every step's table is that step's rows, whole -- so the path has executed before an
    8-GPU node ever sees it"""
    import torch.multiprocessing as mp

    mp.spawn(_worker_rccl, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    seen = np.load(tmp_path / "rccl.npy")
    assert seen.shape == (10, 2, 17)
    for step, t in enumerate(seen):
        assert np.all(t[0] == step) and np.all(t[1] == 100.0 + step)
