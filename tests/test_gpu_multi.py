"""RCCL path of bench.py on a node with at least two GPUs (-m gpu; skipped on a 1-GPU box): the N > 1 flow the driver runs
(`python -m torch.distributed.run ... bench.py --gpus N`) with backend nccl = RCCL over xGMI, weak (frames) and strong (bands)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_two_ranks_over_rccl():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rccl_smoke.py"), "2"], capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert p.stdout.count('"rccl_ranks": 2') == 2, p.stdout


def test_rccl_branch_with_one_rank_on_one_gpu():
    """Round 6: what one GPU allows of the RCCL branch — init_process_group("nccl", world_size=1) on cuda:0 and FrameGatherer's
    DEVICE path (device staging, stream-ordered copy, dist.gather(async_op=True), buffer reuse after wait) over five rounds of frames
    rendered by the HIP kernels: every gathered frame bit-equal to the rendered one (scripts/rccl_one_rank.py).  With two GPUs the
    test above runs the real exchange."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rccl_one_rank.py"), "5"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["rccl_ranks"] == 1 and d["dist_backend"] == "nccl" and d["rounds"] == 5 and d["asynchronous"] is True
    assert d["dist_library"].startswith("RCCL ")


def test_bench_distributed_path_over_rccl_with_one_rank():
    """`bench.py --gpus 1 --force-dist`: the code path the driver's N > 1 runs take — init_process_group("nccl", device_id=...), the
    FrameGatherer on device buffers, barriers, the MAX all-reduce of the time on a device tensor, the synchronous gather round — with a
    process group of ONE rank on cuda:0: one JSON line, rccl_ranks 1, the RCCL version in it."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--res", "128", "--samples", "32",
                        "--steps", "3", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["rccl_ranks"] == 1 and d["config"]["dist_backend"] == "nccl" and d["value"] > 0
    assert str(d["config"]["dist_library"]).startswith("RCCL ") and d["config"]["gather"].startswith("asynchronous")


# ---- the N > 1 flow on ONE GPU (the box the driver's GPU tier runs on): two gloo ranks, both on cuda:0, the DEVICE kernels ----
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _device_rows(y0, step_y, ny, res=64):
    from keypointnerf_amd import ops
    from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
    dev = torch.device("cuda", 0)
    scene = to_device(make_scene(n_views=3, src_hw=(128, 128), tar_hw=(res, res), mask="ellipsoid", seed=5), dev)
    w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
    ps = ops.PreparedScene(scene["img"], scene["cam"], scene["feat_geo"], scene["feat_tex"], scene["sp_data"], scene["src_foreground_mask"])
    out = ops.render_rays(ps, w, scene["cam_tar"], scene["bounds"], grid=(0, y0, 1, res, ny, step_y), n_coarse=32, n_fine=32)
    torch.cuda.synchronize()
    return {k: out[k][0].clone() for k in ("tex_fg_fine", "alpha_fine")}


def _strong_rank(rank, world, port, q):
    import torch.distributed as dist
    from keypointnerf_amd.parallel import FrameGatherer, deinterleave_rows, rows_of_rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    y0, step_y, ny = rows_of_rank(64, rank, world)
    band = _device_rows(y0, step_y, ny)["tex_fg_fine"]
    g = FrameGatherer(world, rank, tuple(band.shape), device="cpu")      # gloo: host staging (RCCL: device, tests above)
    g.submit(band)
    g.finish()
    # by value (numpy): a tensor would travel as a file descriptor fetched from THIS process, which may have exited by then
    q.put((rank, deinterleave_rows(g.frames(0)).cpu().numpy().copy() if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_on_the_device_kernels_assembles_the_one_rank_frame():
    """bench.py --scaling strong's flow with two ranks sharing cuda:0 (gloo): row y -> rank y mod 2, every rank renders its rows with
    the HIP kernels (kpn_render_args.step_y), the bands are gathered to rank 0 and de-interleaved — bit-identical to the frame one
    rank renders.  (On the emulator: tests/test_parallel_gloo.py; over RCCL with one GPU per rank: the test above.)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        item = q.get(timeout=600)
        res[item[0]] = None if item[1] is None else torch.from_numpy(item[1])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    one = _device_rows(0, 0, 64)["tex_fg_fine"].cpu()
    assert res[1] is None and res[0].shape == one.shape == (3, 64, 64)
    assert float(one.abs().max()) > 0 and torch.isfinite(one).all()
    assert torch.equal(res[0], one)


def test_bench_two_gloo_ranks_on_one_gpu_prints_one_line():
    """the launcher path of `bench.py --gpus 2` (torch.distributed.run on 127.0.0.1, RANK / WORLD_SIZE from the environment) with
    both ranks on cuda:0: one JSON line from rank 0, whole-job rays/s, strong scaling by interleaved rows"""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--scaling", "strong", "--res", "128",
                        "--samples", "32", "--steps", "2", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["dist_world_size"] == 2 and d["value"] > 0
