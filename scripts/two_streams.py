import sys, time, torch, ctypes
sys.path.insert(0, ".")
from keypointnerf_amd import ops
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict
sd = random_hotpath_state_dict(seed=3)
scene = make_scene(3, (512, 512), (512, 512), "ellipsoid", seed=1, device="cuda")
w = ops.PackedWeights(sd)
ps = ops.PreparedScene(scene["img"], scene["cam"], scene["feat_geo"], scene["feat_tex"], scene["sp_data"], scene["src_foreground_mask"])
ns = int(sys.argv[1])
streams = [torch.cuda.Stream() for _ in range(ns)]
plans = [ops.RenderPlan(ps, (0, 0, 1, 512, 512), 64, 64, fine=True, chunk_rays=int(sys.argv[2]) if len(sys.argv) > 2 else 0) for _ in range(ns)]
def frame(i):
    with torch.cuda.stream(streams[i % ns]):
        ops.render_rays(ps, w, scene["cam_tar"], scene["bounds"], plan=plans[i % ns])
for i in range(2 * ns): frame(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 8
for i in range(N): frame(i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"streams {ns}: {dt / N * 1e3:.2f} ms/frame")
