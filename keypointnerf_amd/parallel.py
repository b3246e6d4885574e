"""Frames of a render job sharded over the GPUs of one node (one process per GPU).

Rays and frames never interact (every reduction is per ray over samples or per point over views), so
the path shards with NO data-path collective: frame i -> rank i mod world (the reference's
render_video_zju renders frames strictly sequentially on one GPU, src/model.py:178-235).  The only
exchange is the final gather of finished images to rank 0 over RCCL/xGMI (torch.distributed backend
"nccl" is RCCL on ROCm; "gloo" is used by the CPU tests).
"""
import math

import torch


def frames_of_rank(n_frames, rank, world):
    """Round-robin assignment frame i -> rank i % world."""
    return list(range(rank, n_frames, world))


def orbit_target_camera(cam_tar, i, n_frames=90):
    """Target camera i of a turntable orbit around the world y axis through the origin, derived from
    cam_tar (the reference builds a 90-camera orbit with cv2.Rodrigues, src/utils.py:23-72)."""
    a = 2.0 * math.pi * (i % n_frames) / n_frames
    c, s = math.cos(a), math.sin(a)
    rot = torch.tensor([[c, 0.0, s, 0.0], [0.0, 1.0, 0.0, 0.0], [-s, 0.0, c, 0.0], [0.0, 0.0, 0.0, 1.0]],
                       dtype=cam_tar["RT"].dtype, device=cam_tar["RT"].device)
    out = dict(cam_tar)
    out["RT"] = cam_tar["RT"] @ rot[None]
    out["KRT"] = cam_tar["K"] @ out["RT"]
    return out


def render_job(render_frame, n_frames, rank=0, world=1, group=None, gather=True):
    """Runs render_frame(i) -> (C,H,W) tensor for this rank's frames; returns the (n_frames,C,H,W) stack on
    rank 0 (None elsewhere) if gather.  One all_gather per round of `world` frames (3 MB per 512^2 RGB
    frame; a few tens of microseconds on one xGMI link)."""
    mine = frames_of_rank(n_frames, rank, world)
    rounds = (n_frames + world - 1) // world
    frames = [None] * n_frames
    for r in range(rounds):
        img = render_frame(mine[r]) if r < len(mine) else None
        if world == 1:
            frames[mine[r]] = img
            continue
        import torch.distributed as dist
        if img is None:  # ragged last round: contribute a dummy of the right shape
            img = torch.zeros_like(last)
        last = img
        if not gather:
            continue
        bucket = [torch.empty_like(img) for _ in range(world)]
        dist.all_gather(bucket, img.contiguous(), group=group)
        for k in range(world):
            i = r * world + k
            if i < n_frames:
                frames[i] = bucket[k]
    if not gather:
        return None
    if rank != 0:
        return None
    return torch.stack(frames, 0)
