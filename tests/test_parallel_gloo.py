"""World-size-2 CPU test (gloo) of the frame sharder + final gather (keypointnerf_amd/parallel.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keypointnerf_amd.parallel import frames_of_rank, orbit_target_camera, render_job


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


def test_assignment_is_a_partition():
    for n, w in ((7, 2), (200, 8), (3, 4)):
        got = sorted(i for r in range(w) for i in frames_of_rank(n, r, w))
        assert got == list(range(n))


def test_two_rank_render_job_gathers_all_frames():
    world, n_frames = 2, 5  # ragged: rank 1 has one frame fewer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, rendered, out = q.get(timeout=120)
        res[rank] = (rendered, out)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == [0, 2, 4] and res[1][0] == [1, 3]
    assert res[1][1] is None
    expect = torch.stack([_fake_frame(i) for i in range(n_frames)])
    assert torch.equal(res[0][1], expect)


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
