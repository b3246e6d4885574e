"""Frames of a render job sharded over the GPUs of one node (one process per GPU).

Rays and frames never interact (every reduction is per ray over samples or per point over views), so
the path shards with NO data-path collective: frame i -> rank i mod world (the reference's
render_video_zju renders frames strictly sequentially on one GPU, src/model.py:178-235).  The only
exchange is the final gather of finished images to rank 0 over RCCL/xGMI (torch.distributed backend
"nccl" is RCCL on ROCm; "gloo" is used by the CPU tests).
"""
import math

import numpy as np
import torch


def frames_of_rank(n_frames, rank, world):
    """Round-robin assignment frame i -> rank i % world."""
    return list(range(rank, n_frames, world))


def _rodrigues(rvec):
    """Rotation matrix of a rotation vector (what cv2.Rodrigues returns for a (3,) float32 input: computed
    in double precision from the float32 components, R = cos(t) I + (1-cos(t)) r r^T + sin(t) [r]_x)."""
    r = np.asarray(rvec, np.float32).astype(np.float64)
    t = float(np.sqrt((r * r).sum()))
    if t < 2.220446049250313e-16:
        return np.eye(3)
    r = r / t
    c, s = math.cos(t), math.sin(t)
    rx = np.array([[0.0, -r[2], r[1]], [r[2], 0.0, -r[0]], [-r[1], r[0], 0.0]])
    return c * np.eye(3) + (1.0 - c) * np.outer(r, r) + s * rx


def orbit_cameras(headpose, focal, trans, sc_factor, im_w, im_h, znear, zfar, n_frames=90):
    """The reference's turntable (get_360cameras, src/utils.py:23-72) restated without cv2: camera idx looks
    at the subject from angle theta = idx * 2 pi / n_frames about the head-pose frame's y axis, flipped by pi
    about x, `trans` in front of it.  Returns the reference's list of dicts
    {'w2cs','c2ws','intrinsics','im_w','im_h','znear','zfar'} (tensors on headpose's device)."""
    device = headpose.device
    T_i = torch.eye(4, device=device)
    rot = headpose[:3, :3].t()
    T_i[:3, :3] = rot
    T_i[:3, 3] = -rot @ headpose[:3, 3]
    dR1 = _rodrigues([np.pi, 0.0, 0.0]).astype(np.float32)
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = float(np.float32(focal))
    K[0, 2], K[1, 2] = float(np.float32(im_w / 2)), float(np.float32(im_h / 2))
    cams, theta = [], 0
    for _ in range(n_frames):
        dR2 = _rodrigues([0.0, theta, 0.0]).astype(np.float32)
        ext = torch.eye(4)
        ext[:3, :3] = torch.from_numpy((dR1 @ dR2).astype(np.float32))
        ext[:3, 3] = torch.tensor([0.0, 0.0, float(trans)])
        extrinsic = torch.matmul(ext.to(device), T_i).clone()
        extrinsic[:3, 3] *= sc_factor
        theta = theta + 2.0 * np.pi / n_frames
        cams.append({"w2cs": extrinsic, "c2ws": torch.inverse(extrinsic), "intrinsics": K.to(device)[None].clone(),
                     "im_w": im_w, "im_h": im_h, "znear": znear, "zfar": zfar})
    return cams


def zju_orbit_cameras(headpose, sc_factor=1.0, n_frames=90, im_w=512, im_h=512):
    """The orbit render_video_zju builds (src/model.py:178-187, 213-214): trans 5.0, near/far = trans -+ 3,
    focal = 25 W + 0.9 (0.125 W - 25 W) = 1337.6 px at W = 512."""
    trans = 5.0
    fstart, fend = im_w * 25, im_w * .125
    focal = fstart + 0.9 * (fend - fstart)
    return orbit_cameras(headpose, focal, trans, sc_factor, im_w, im_h, (trans - 3.0) * sc_factor, (trans + 3.0) * sc_factor,
                         n_frames)


def orbit_cam_tar(camera):
    """cam_tar dict render_novel_views derives from an orbit camera (src/model.py:484-491)."""
    rt = camera["w2cs"].unsqueeze(0)
    return {"K": camera["intrinsics"], "RT": rt, "KRT": camera["intrinsics"] @ rt, "width": camera["im_w"],
            "height": camera["im_h"], "nml_scale": 100., "znear": camera["znear"], "zfar": camera["zfar"]}


def orbit_target_camera(cam_tar, i, n_frames=90):
    """Target camera i of a turntable around the world y axis through the origin, derived from an existing
    cam_tar (keeps its intrinsics and distance; used by the synthetic benchmark scene, whose subject sits at
    the origin — the reference-shaped orbit is orbit_cameras above)."""
    a = 2.0 * math.pi * (i % n_frames) / n_frames
    c, s = math.cos(a), math.sin(a)
    rot = torch.tensor([[c, 0.0, s, 0.0], [0.0, 1.0, 0.0, 0.0], [-s, 0.0, c, 0.0], [0.0, 0.0, 0.0, 1.0]],
                       dtype=cam_tar["RT"].dtype, device=cam_tar["RT"].device)
    out = dict(cam_tar)
    out["RT"] = cam_tar["RT"] @ rot[None]
    out["KRT"] = cam_tar["K"] @ out["RT"]
    return out


def gather_frames_to_root(img, world, rank, group=None, into=None):
    """The job's only exchange: every rank contributes one finished frame, rank 0 receives all `world` of them
    (dist.gather = grouped send/recv over RCCL; nobody else receives anything).  `into` (world, *img.shape) is an
    optional preallocated destination on rank 0.  Returns the list of frames on rank 0, None elsewhere."""
    import torch.distributed as dist
    img = img.contiguous()
    if rank == 0:
        if into is None:
            into = torch.empty((world,) + tuple(img.shape), dtype=img.dtype, device=img.device)
        bucket = list(into.unbind(0))
        dist.gather(img, gather_list=bucket, dst=0, group=group)
        return bucket
    dist.gather(img, gather_list=None, dst=0, group=group)
    return None


class FrameGatherer:
    """The same exchange OFF the critical path: `submit(img)` copies the finished frame into one of `depth` staging buffers and
    starts its gather to rank 0 asynchronously (dist.gather(async_op=True): RCCL runs it on its own stream once the copy is done),
    so that frame i travels over xGMI while frame i + 1 renders; a staging buffer is waited for only when it comes round again.
    `finish()` waits for everything in flight.  On rank 0 `frames(k)` is the list of the `world` frames of round k mod depth
    (valid after that round's wait).  With gloo the tensors live on the host and submit() blocks for the device-to-host copy."""

    def __init__(self, world, rank, shape, dtype=torch.float32, device="cuda", group=None, depth=2):
        import torch.distributed as dist
        self.dist, self.world, self.rank, self.group, self.depth = dist, world, rank, group, depth
        self.staging = [torch.empty(tuple(shape), dtype=dtype, device=device) for _ in range(depth)]
        self.dest = [torch.empty((world,) + tuple(shape), dtype=dtype, device=device) for _ in range(depth)] if rank == 0 else None
        self.work = [None] * depth
        self.rounds = 0
        self.asynchronous = True

    def submit(self, img):
        k = self.rounds % self.depth
        if self.work[k] is not None:
            self.work[k].wait()                       # the buffer's previous gather (two rounds ago) has long finished
        # device staging (RCCL): the copy is stream-ordered in front of the gather.  Host staging (gloo): pageable memory and a
        # host-side read by the backend — the copy must have completed when gather() looks at the buffer, so it is synchronous
        self.staging[k].copy_(img, non_blocking=self.staging[k].is_cuda)
        bucket = list(self.dest[k].unbind(0)) if self.rank == 0 else None
        if self.asynchronous:
            try:
                self.work[k] = self.dist.gather(self.staging[k], gather_list=bucket, dst=0, group=self.group, async_op=True)
            except (RuntimeError, NotImplementedError):   # a backend without asynchronous gather: the same exchange, blocking
                self.asynchronous = False
        if not self.asynchronous:
            self.dist.gather(self.staging[k], gather_list=bucket, dst=0, group=self.group)
            self.work[k] = None
        self.rounds += 1
        return k

    def frames(self, k):
        return list(self.dest[k % self.depth].unbind(0)) if self.rank == 0 else None

    def finish(self):
        for k, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[k] = None


def rows_of_rank(height, rank, world):
    """Strong scaling of ONE frame by INTERLEAVED rows: image row y belongs to rank y mod world, i.e. rank r renders the grid
    (y0 = r, step_y = world, ny = ceil((height - r) / world)).  The subject sits in the middle of the frame (a third of the
    field evaluations are valid, concentrated there): contiguous bands leave the outer ranks nearly idle, rows dealt round-robin
    give every rank the same share of the subject (tests/test_parallel_gloo.py checks the valid-row counts of the bench scene).
    Returns (y0, step_y, ny)."""
    return rank, world, (height - rank + world - 1) // world


def deinterleave_rows(bands):
    """bands: the ranks' (C, H/W, Wd) bands of one frame in rank order (a (W, C, H/W, Wd) tensor or a list) -> (C, H, Wd): row y of
    the frame is row y // W of band y mod W.  Needs height % world == 0 (equal bands gather into one buffer)."""
    b = torch.stack(list(bands), 0) if not isinstance(bands, torch.Tensor) else bands
    W, C, n, Wd = b.shape
    return b.permute(1, 2, 0, 3).reshape(C, n * W, Wd)


def band_of_rank(height, rank, world):
    """Strong scaling of ONE frame: contiguous bands of image rows, rank r owns rows [y0, y0 + n)."""
    base, extra = divmod(height, world)
    y0 = rank * base + min(rank, extra)
    return y0, base + (1 if rank < extra else 0)


def render_job(render_frame, n_frames, rank=0, world=1, group=None, gather=True, frame_like=None):
    """Runs render_frame(i) -> (C,H,W) tensor for this rank's frames; returns the (n_frames,C,H,W) stack on
    rank 0 (None elsewhere) if gather.  One gather to rank 0 per round of `world` frames (3 MB per 512^2 RGB
    frame; a few tens of microseconds on one xGMI link).  Ranks without a frame in a ragged round contribute a
    dummy: its shape comes from `frame_like` (a tensor or a (shape, dtype, device) tuple), from this rank's
    previous frame, or — when this rank never renders (n_frames < world) — from rank 0's first frame, whose
    metadata is broadcast once."""
    mine = frames_of_rank(n_frames, rank, world)
    rounds = (n_frames + world - 1) // world
    if world == 1:
        frames = [render_frame(i) for i in mine]
        return torch.stack(frames, 0) if (gather and frames) else None
    import torch.distributed as dist
    frames = [None] * n_frames
    like = None
    if frame_like is not None:
        like = (tuple(frame_like.shape), frame_like.dtype, frame_like.device) if isinstance(frame_like, torch.Tensor) else frame_like
    for r in range(rounds):
        img = render_frame(mine[r]) if r < len(mine) else None
        if not gather:
            continue
        if like is None:                                   # agree on the frame metadata once (rank 0 always renders round 0)
            meta = [(tuple(img.shape), str(img.dtype).replace("torch.", ""))] if rank == 0 else [None]
            dist.broadcast_object_list(meta, src=0, group=group)
            dev = img.device if img is not None else (torch.device("cuda", torch.cuda.current_device())
                                                      if dist.get_backend(group) == "nccl" else torch.device("cpu"))
            like = (tuple(meta[0][0]), getattr(torch, meta[0][1]), dev)
        if img is None:                                    # ragged round: contribute a dummy of the agreed shape
            img = torch.zeros(like[0], dtype=like[1], device=like[2])
        bucket = gather_frames_to_root(img, world, rank, group=group)
        if rank == 0:
            for k in range(world):
                i = r * world + k
                if i < n_frames:
                    frames[i] = bucket[k].clone()
    if not gather or rank != 0:
        return None
    return torch.stack(frames, 0)
