#!/usr/bin/env python
"""BASELINE configs[2]: a full 360-degree orbit of novel views (render_dynamic.py / render_video_zju in the
reference, src/model.py:178-235: 90-camera turntable, frame i uses camera i % 90) on a synthetic subject, frames
sharded round-robin over the GPUs of one node, finished frames gathered to rank 0 over RCCL.

    python scripts/render_orbit.py --frames 200 --res 512                 # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/render_orbit.py --frames 200

Prints frames/s and writes orbit_rgb8.npy (N,H,W,3 uint8, quantised on the device) on rank 0 with --save.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--save", default="")
    ap.add_argument("--with-encoders", action="store_true",
                    help="every frame is a new source set, as in render_video_zju: run both image encoders (cost-equivalent stand-ins of "
                         "the reference's 28 M parameters, scripts/encoder_standin.py) on the 3 x 512^2 sources and prepare the scene from "
                         "their maps before rendering — once per frame, which is what the drop-in does (the reference runs them twice)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from keypointnerf_amd import ops
    from keypointnerf_amd.parallel import orbit_cam_tar, render_job, zju_orbit_cameras
    from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict

    scene = make_scene(args.views, (args.res, args.res), (args.res, args.res), "ellipsoid", seed=1, device="cuda")
    w = ops.PackedWeights(random_hotpath_state_dict(seed=3))
    ps = ops.PreparedScene(scene["img"], scene["cam"], scene["feat_geo"], scene["feat_tex"], scene["sp_data"], scene["src_foreground_mask"])
    plan = ops.RenderPlan(ps, (0, 0, 1, args.res, args.res), args.samples, args.samples, fine=True)

    # the reference's own turntable (get_360cameras as render_video_zju calls it, src/model.py:178-214, src/utils.py:23-72:
    # 90 cameras at 5 m, focal 1337.6 px at 512, near/far 2/8).  The head pose is the subject's frame: the synthetic
    # subject stands along world -y (OpenCV y-down sources), i.e. rotated by pi about x against the orbit's convention.
    headpose = torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0])).cuda()
    cams = zju_orbit_cameras(headpose, sc_factor=1.0, n_frames=90, im_w=args.res, im_h=args.res)
    cam_tars = [orbit_cam_tar(c) for c in cams]

    enc = None
    if args.with_encoders:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import encoder_standin
        enc = (encoder_standin.GeoEncoder().cuda().eval(), encoder_standin.TexEncoder().cuda().eval())

    def render_frame(i):
        cam = cam_tars[i % 90]                                                      # camera = orbit[frame_index % 90], :214
        scene_i = ps
        if enc is not None:
            with torch.no_grad():
                feat_geo, feat_tex = encoder_standin.encode(enc[0], enc[1], scene["img"])
            scene_i = ops.PreparedScene(scene["img"], scene["cam"], feat_geo, feat_tex, scene["sp_data"], scene["src_foreground_mask"])
        out = ops.render_rays(scene_i, w, cam, scene["bounds"], plan=plan)
        return ops.frame_to_rgb8(out["tex_fg_fine"]).permute(2, 0, 1).contiguous()  # (3,H,W) uint8, what the gather moves

    render_frame(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = render_job(render_frame, args.frames, rank, world)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(f"orbit{' with both encoders per frame' if enc is not None else ''}: {args.frames} frames {args.res}x{args.res} on {world} GPU(s): {dt:.2f} s = {args.frames / dt:.2f} frames/s; "
              f"gathered {tuple(frames.shape)} {frames.dtype}")
        if args.save:
            import numpy as np
            np.save(args.save, frames.permute(0, 2, 3, 1).cpu().numpy())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
