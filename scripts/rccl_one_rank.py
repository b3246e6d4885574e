#!/usr/bin/env python
"""What ONE GPU allows of the RCCL branch (round 6; reference render_dynamic.py:35 / train.py:64,71 run one process per GPU over
NCCL): init_process_group("nccl", world_size=1, device_id=cuda:0) and parallel.FrameGatherer's DEVICE path — device staging
buffers, the stream-ordered non-blocking copy in front of the collective, dist.gather(async_op=True) on RCCL's stream, buffer reuse
after work.wait() — for several rounds of frames rendered by the HIP kernels; every gathered frame must equal the rendered one.
Prints one JSON line {"rccl_ranks": 1, "dist_library": "RCCL x.y.z", "rounds": n, "asynchronous": true, ...}.
MEASUREMENT / TEST INFRASTRUCTURE (tests/test_gpu_multi.py runs it)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    res = 96
    from keypointnerf_amd import ops
    from keypointnerf_amd.parallel import FrameGatherer, orbit_target_camera
    from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    scene = to_device(make_scene(n_views=3, src_hw=(128, 128), tar_hw=(res, res), mask="ellipsoid", seed=5, tar_focal_at_512=800.0), dev)
    w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
    ps = ops.PreparedScene(scene["img"], scene["cam"], scene["feat_geo"], scene["feat_tex"], scene["sp_data"], scene["src_foreground_mask"])
    plan = ops.RenderPlan(ps, (0, 0, 1, res, res), 32, 32, fine=True)
    g = FrameGatherer(1, 0, (3, res, res), device=dev, depth=2)
    assert g.staging[0].is_cuda and g.dest[0].is_cuda
    kept, t0 = [], time.perf_counter()
    for i in range(rounds):                              # frame i travels while frame i + 1 renders; buffers come round again from i = 2
        cam = orbit_target_camera(scene["cam_tar"], i, n_frames=12)
        img = ops.render_rays(ps, w, cam, scene["bounds"], plan=plan)["tex_fg_fine"][0]
        kept.append(img.clone())
        k = g.submit(img)
        if i >= 1:                                       # the PREVIOUS round's frame, once its gather has been waited for
            kp = (i - 1) % 2
            if g.work[kp] is not None:
                g.work[kp].wait()
            torch.cuda.current_stream().synchronize()
            got = g.frames(kp)[0]
            assert torch.equal(got, kept[i - 1]), f"round {i - 1}: gathered frame differs from the rendered one"
    g.finish()
    torch.cuda.synchronize()
    assert torch.equal(g.frames((rounds - 1) % 2)[0], kept[-1])
    assert len({float(k.sum()) for k in kept}) == rounds   # the frames really differ from round to round
    dt = time.perf_counter() - t0
    try:
        lib = "RCCL " + ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:  # noqa: BLE001
        lib = f"unknown ({e})"
    print(json.dumps({"rccl_ranks": dist.get_world_size(), "dist_backend": dist.get_backend(), "dist_library": lib, "rounds": rounds,
                      "asynchronous": bool(g.asynchronous), "frame_bytes": 3 * res * res * 4, "seconds": dt,
                      "what": "FrameGatherer device path over the nccl (= RCCL) process group with one rank on cuda:0: device staging, "
                              "async gather, buffer reuse; every gathered frame bit-equal to the rendered one"}))
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
