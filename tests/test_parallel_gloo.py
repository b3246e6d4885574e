"""World-size-2 CPU test (gloo) of the frame sharder + final gather (keypointnerf_amd/parallel.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keypointnerf_amd.parallel import (FrameGatherer, band_of_rank, deinterleave_rows, frames_of_rank, orbit_cam_tar, orbit_target_camera,
                                       render_job, rows_of_rank, zju_orbit_cameras)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _by_value(t):
    """A tensor put on a multiprocessing queue travels as a file descriptor the RECEIVER fetches from the sending process — which may
    have exited by then (EOFError in the parent: seen once on the GPU box).  A numpy array is pickled into the pipe itself."""
    return None if t is None else t.detach().cpu().numpy().copy()


def _from_queue(x):
    import numpy as np
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else x


def _fake_frame(i):
    return torch.full((3, 4, 5), float(i)) + torch.arange(5.0)


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rendered = []

    def render(i):
        rendered.append(i)
        return _fake_frame(i)

    out = render_job(render, n_frames, rank, world)
    q.put((rank, rendered, _by_value(out)))
    dist.barrier()
    dist.destroy_process_group()


def _gatherer_worker(rank, world, port, rounds, q):
    """FrameGatherer: every rank submits one frame per round (asynchronously, two staging buffers); rank 0 reads each round's
    frames after the buffer's next wait / finish()."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = FrameGatherer(world, rank, (3, 4, 5), device="cpu")
    got = []
    for r in range(rounds):
        k = g.submit(_fake_frame(r * world + rank))
        if r >= 1 and rank == 0:                     # round r - 1 used the other buffer: wait for it, then read it
            g.work[(k + 1) % 2].wait()
            got.append(torch.stack(g.frames(k + 1)).clone())
    g.finish()
    if rank == 0:
        got.append(torch.stack(g.frames(rounds - 1)).clone())
    q.put((rank, None, _by_value(torch.cat(got)) if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_async_frame_gatherer_delivers_every_round():
    world, rounds = 2, 5
    res = _run_world(_gatherer_worker, world, rounds)
    expect = torch.stack([_fake_frame(i) for i in range(world * rounds)])
    assert torch.equal(res[0][1], expect) and res[1][1] is None


def test_bands_partition_the_rows():
    for h, w in ((512, 8), (512, 3), (7, 4), (4096, 8)):
        rows = []
        for r in range(w):
            y0, n = band_of_rank(h, r, w)
            rows += list(range(y0, y0 + n))
        assert rows == list(range(h))


def test_assignment_is_a_partition():
    for n, w in ((7, 2), (200, 8), (3, 4)):
        got = sorted(i for r in range(w) for i in frames_of_rank(n, r, w))
        assert got == list(range(n))


def test_two_rank_render_job_gathers_all_frames():
    world, n_frames = 2, 5  # ragged: rank 1 has one frame fewer
    res = _run_world(_worker, world, n_frames)
    assert res[0][0] == [0, 2, 4] and res[1][0] == [1, 3]
    assert res[1][1] is None
    expect = torch.stack([_fake_frame(i) for i in range(n_frames)])
    assert torch.equal(res[0][1], expect)


def _run_world(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=300)
        res[item[0]] = tuple(_from_queue(x) for x in item[1:])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_fewer_frames_than_ranks():
    """n_frames < world: rank 1 never renders; it must still take part in the gather (it used to raise
    UnboundLocalError while rank 0 blocked in the collective)."""
    res = _run_world(_worker, 2, 1)
    assert res[0][0] == [0] and res[1][0] == []
    assert res[1][1] is None
    assert torch.equal(res[0][1], torch.stack([_fake_frame(0)]))


def _real_worker(rank, world, port, n_frames, q):
    """Every rank renders REAL frames: the kernel sources on the wave64 host emulator (tests/simt), cameras of
    the reference-shaped orbit."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = render_job(lambda i: _render_real_frame(i), n_frames, rank, world)
    q.put((rank, _by_value(out)))
    dist.barrier()
    dist.destroy_process_group()


_REAL = {}


def _render_real_frame(i):
    from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict
    from tests import simt_harness as sh
    if not _REAL:
        lib = sh.simt_lib()
        scene = make_scene(n_views=3, src_hw=(32, 32), tar_hw=(8, 8), mask="ellipsoid", seed=5)
        head = torch.eye(4)
        cams = zju_orbit_cameras(head, n_frames=6, im_w=8, im_h=8)
        for c in cams:                       # an 8x8 image plane: shrink the 1337.6-px focal length accordingly
            c["intrinsics"][0, 0, 0] = c["intrinsics"][0, 1, 1] = 12.0
        _REAL.update(lib=lib, scene=scene, hs=sh.HostScene(lib, scene), cams=cams,
                     packed=sh.pack_weights(lib, random_hotpath_state_dict(seed=3)))
    r = _REAL
    cam_tar = orbit_cam_tar(r["cams"][i % len(r["cams"])])
    o = sh.render(r["lib"], r["hs"], r["packed"], cam_tar, r["scene"]["bounds"], (0, 0, 1, 8, 8), 8, 8)
    return torch.from_numpy(o["tex_fg_fine"].copy())


def test_two_rank_job_renders_real_frames():
    n_frames = 3
    res = _run_world(_real_worker, 2, n_frames)
    assert res[1][0] is None
    got = res[0][0]
    assert got.shape == (n_frames, 3, 8, 8)
    expect = torch.stack([_render_real_frame(i) for i in range(n_frames)])   # single process, same frames
    assert torch.equal(got, expect)
    assert torch.isfinite(got).all() and float(got.abs().max()) > 0
    assert not torch.equal(got[0], got[1])                                   # different cameras, different images


def _render_real_rows(y0, step_y, ny):
    """rows y0, y0 + step_y, ... of frame 0 of _render_real_frame's job (the same scene, camera and weights)"""
    from tests import simt_harness as sh
    _render_real_frame(0)
    r = _REAL
    cam_tar = orbit_cam_tar(r["cams"][0])
    o = sh.render(r["lib"], r["hs"], r["packed"], cam_tar, r["scene"]["bounds"], (0, y0, 1, 8, ny, step_y), 8, 8)
    return torch.from_numpy(o["tex_fg_fine"].copy())


def _strong_worker(rank, world, port, q):
    """--scaling strong: ONE frame, row y -> rank y mod world; every rank renders its rows (real kernels on the emulator), the
    bands are gathered to rank 0 (FrameGatherer) and de-interleaved there."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    y0, step_y, ny = rows_of_rank(8, rank, world)
    band = _render_real_rows(y0, step_y, ny)
    g = FrameGatherer(world, rank, tuple(band.shape), device="cpu")
    g.submit(band)
    g.finish()
    q.put((rank, _by_value(deinterleave_rows(g.frames(0))) if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_interleaved_rows_assemble_the_one_rank_frame():
    res = _run_world(_strong_worker, 2)
    assert res[1][0] is None
    assert torch.equal(res[0][0], _render_real_frame(0))            # bit-identical to the frame rendered by one rank


def test_interleaved_rows_partition_and_balance_the_bench_frame():
    """rows_of_rank is a partition of the rows, and on the BENCH scene (512 x 512, subject framed like the reference's orbit,
    a third of the field evaluations valid and concentrated in the middle of the frame) the ranks' shares of the valid
    (point, view) rows differ by less than 10 % for 2, 4 and 8 ranks — contiguous bands (band_of_rank, round 3) leave the outer
    ranks nearly idle, which the same count shows.  Validity restated in numpy (projection, frustum, fg mask; src/model.py:713-739)
    on every 4th column of the frame's coarse samples."""
    import numpy as np
    from oracle import oracle
    from keypointnerf_amd.synthetic import make_scene
    for h, w in ((512, 8), (512, 3), (7, 4)):
        rows = sorted(y0 + k * st for r in range(w) for (y0, st, n) in [rows_of_rank(h, r, w)] for k in range(n))
        assert rows == list(range(h))
    res, S = 512, 64
    scene = make_scene(n_views=3, src_hw=(res, res), tar_hw=(res, res), mask="ellipsoid", seed=1, tar_focal_at_512=800.0)
    ys, xs = np.meshgrid(np.arange(res), np.arange(0, res, 4), indexing="ij")
    pix = np.stack([xs.reshape(-1), ys.reshape(-1)], -1).astype(np.int32)
    dirs, cam_pos, near, far = oracle.make_rays(scene["cam_tar"], scene["bounds"], pix)
    z = near[:, None] + (far - near)[:, None] * np.linspace(0.0, 1.0, S, dtype=np.float32)[None]
    pts = cam_pos[None, None] + dirs[:, None] * z[..., None]                     # (R, S, 3)
    KRT = scene["cam"]["KRT"].numpy()
    fg = scene["src_foreground_mask"].numpy().reshape(3, res, res).astype(np.float32)
    valid = np.ones(pts.shape[:2], bool)
    for v in range(3):
        vh = pts @ KRT[v, :3, :3].T + KRT[v, :3, 3]
        x, y = vh[..., 0] / vh[..., 2], vh[..., 1] / vh[..., 2]
        xn, yn = 2 * x / (res - 1) - 1, 2 * y / (res - 1) - 1
        zn = 2 * (vh[..., 2] - 2.0) / 3.0 - 1
        inside = (np.abs(xn) <= 1.01) & (np.abs(yn) <= 1.01) & (zn >= -1)
        ix, iy = np.clip((xn + 1) / 2 * (res - 1), 0, res - 1), np.clip((yn + 1) / 2 * (res - 1), 0, res - 1)
        x0, y0 = np.floor(ix).astype(int), np.floor(iy).astype(int)
        x1, y1 = np.minimum(x0 + 1, res - 1), np.minimum(y0 + 1, res - 1)
        fx, fy = ix - x0, iy - y0
        m = fg[v][y0, x0] * (1 - fx) * (1 - fy) + fg[v][y0, x1] * fx * (1 - fy) + fg[v][y1, x0] * (1 - fx) * fy + fg[v][y1, x1] * fx * fy
        valid &= inside & (m > 0.1)
    per_row = valid.reshape(res, -1).sum(1).astype(np.float64)                 # valid coarse samples per image row
    assert 0.25 < per_row.sum() / valid.size < 0.45                            # the bench scene's third
    for world in (2, 4, 8):
        inter = np.array([per_row[r::world].sum() for r in range(world)])
        assert inter.max() / inter.min() < 1.10, (world, inter)
        bands = np.array([per_row[y0:y0 + n].sum() for y0, n in (band_of_rank(res, r, world) for r in range(world))])
        if world >= 4:
            assert bands.max() / max(bands.min(), 1.0) > 2.0, (world, bands)    # what round 3's contiguous bands did


def test_reference_orbit_cameras():
    """orbit_cameras restates get_360cameras (reference src/utils.py:23-72): rigid world->camera transforms at
    distance `trans` from the head-pose origin, one full turn in n_frames steps, K as the reference builds it."""
    head = torch.eye(4)
    head[:3, 3] = torch.tensor([0.3, -0.2, 0.1])
    cams = zju_orbit_cameras(head, sc_factor=1.0, n_frames=90)
    assert len(cams) == 90
    K = cams[0]["intrinsics"][0]
    assert abs(float(K[0, 0]) - 1337.6) < 1e-3 and float(K[0, 2]) == 256.0 and float(K[1, 2]) == 256.0
    assert cams[0]["znear"] == 2.0 and cams[0]["zfar"] == 8.0
    centres = []
    for c in cams:
        R, t = c["w2cs"][:3, :3], c["w2cs"][:3, 3]
        assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-5)
        assert torch.allclose(c["c2ws"] @ c["w2cs"], torch.eye(4), atol=1e-5)
        centres.append(-(R.T @ t))
    centres = torch.stack(centres)
    # every camera is `trans` = 5 from the head-pose origin (T_i maps it to 0) and looks at it
    assert torch.allclose((centres - head[:3, 3]).norm(dim=1), torch.full((90,), 5.0), atol=1e-4)
    p = torch.cat([head[:3, 3], torch.ones(1)])
    for c in cams[::15]:
        pc = c["w2cs"] @ p
        assert abs(float(pc[0])) < 1e-4 and abs(float(pc[1])) < 1e-4 and abs(float(pc[2]) - 5.0) < 1e-4
    # camera 0: Rodrigues(pi about x) = diag(1,-1,-1)
    assert torch.allclose(cams[0]["w2cs"][:3, :3], torch.diag(torch.tensor([1.0, -1.0, -1.0])), atol=1e-6)
    # quarter turn after n_frames / 4... 90 is not divisible by 4: check the half turn instead
    assert torch.allclose(cams[45]["w2cs"][:3, :3], torch.diag(torch.tensor([-1.0, -1.0, 1.0])), atol=1e-5)
    ct = orbit_cam_tar(cams[3])
    assert ct["K"].shape == (1, 4, 4) and ct["RT"].shape == (1, 4, 4) and torch.allclose(ct["KRT"], ct["K"] @ ct["RT"])


def test_orbit_camera_is_rigid():
    K = torch.eye(4)[None]
    RT = torch.eye(4)[None]
    RT[0, :3, 3] = torch.tensor([0.0, 0.0, 3.0])
    cam = {"K": K, "RT": RT, "KRT": K @ RT}
    c = orbit_target_camera(cam, 13)
    R = c["RT"][0, :3, :3]
    assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-6)
    # the orbit keeps the distance to the origin
    pos = -(R.T @ c["RT"][0, :3, 3])
    assert abs(float(pos.norm()) - 3.0) < 1e-5
