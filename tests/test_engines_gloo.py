"""Engine-per-GPU exchange on CPU: world_size 2, gloo.  The same code runs over RCCL in bench.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from homerhevc_amd.engines import exchange_reference, frames_for_engine


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, steps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [(24, 40), (12, 20), (12, 20)]
    recv = [torch.zeros(s, dtype=torch.int16) for s in shapes]
    for step in range(steps):
        # "reconstruction" of the frame this engine just finished: value encodes (frame index, plane)
        frame = frames_for_engine(rank, world, 100)[step]
        send = [torch.full(s, frame * 4 + i, dtype=torch.int16) for i, s in enumerate(shapes)]
        for r in exchange_reference(send, recv, rank, world):
            r.wait()
        prev_frame = frames_for_engine((rank - 1) % world, world, 100)[step]
        for i, t in enumerate(recv):
            assert torch.all(t == prev_frame * 4 + i), (rank, step, i)
    dist.barrier()
    dist.destroy_process_group()


def test_reference_ring_exchange_world2():
    mp.spawn(_worker, args=(2, _free_port(), 3), nprocs=2, join=True)


def test_frame_dealing_is_round_robin():
    assert frames_for_engine(0, 8, 20) == [0, 8, 16]
    assert frames_for_engine(7, 8, 20) == [7, 15]
    got = sorted(f for r in range(3) for f in frames_for_engine(r, 3, 10))
    assert got == list(range(10))
    assert exchange_reference([], [], 0, 1) == []
