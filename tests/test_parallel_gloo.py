"""World-size-2 CPU test (gloo) of the frame sharder + final gather (keypointnerf_amd/parallel.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keypointnerf_amd.parallel import (FrameGatherer, band_of_rank, frames_of_rank, orbit_cam_tar, orbit_target_camera, render_job,
                                       zju_orbit_cameras)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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
    q.put((rank, rendered, None if out is None else out.clone()))
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
    q.put((rank, None, torch.cat(got) if rank == 0 else None))
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
        res[item[0]] = item[1:]
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
    q.put((rank, None if out is None else out.clone()))
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
