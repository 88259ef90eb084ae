"""Frame-parallel encoder engines across GPUs: one engine (process) per GPU.

The reference deals frames to `num_enc_engines` engines in decode order (`encoder_engine_thread`,
hmr_encoder_lib.c:3154-3211) and the only data engines share is the reconstructed, padded reference picture
(`hvenc->ref_wnds`, hmr_private.h:1407): engine e needs engine e-1's reconstruction of the previous frame.
That is a ring of point-to-point transfers, not a reduction, so it maps to RCCL send/recv over xGMI.
"""
import torch.distributed as dist


def frames_for_engine(rank, world, n_frames):
    """Frame indices (decode order) engine `rank` encodes: round-robin, as the reference hands them out."""
    return list(range(rank, n_frames, world))


def exchange_reference(send_planes, recv_planes, rank, world):
    """Post the ring transfer of one reference picture: send ours to engine rank+1, receive engine rank-1's.

    Returns the request list (wait on them before the next frame reads `recv_planes`); [] when world == 1.
    Works on any backend (nccl = RCCL on ROCm, gloo in the CPU tests).
    """
    if world <= 1:
        return []
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    ops = []
    for t_send, t_recv in zip(send_planes, recv_planes):
        ops.append(dist.P2POp(dist.isend, t_send, nxt))
        ops.append(dist.P2POp(dist.irecv, t_recv, prv))
    return dist.batch_isend_irecv(ops)
