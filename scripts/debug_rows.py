"""Which part of a (point, view) row is wrong when the asm-free split-bf16 build fails at 2 waves per SIMD: the 64-vector
(MFMA chain) or also the gather record (VALU + loads only)?  Rows of mode 0 (fp32 kernel) vs mode 1, matched per point."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keypointnerf_amd import lib as kl, ops
from keypointnerf_amd.synthetic import make_scene, random_hotpath_state_dict, to_device
if os.environ.get('KPN_EXPERIMENT_LIB'):
    kl._default = kl.KpnLibrary(os.environ['KPN_EXPERIMENT_LIB'])
L = kl.get_library()
dev = torch.device("cuda", 0)
sc = to_device(make_scene(n_views=3, src_hw=(512, 512), tar_hw=(64, 64), mask="dense", seed=1), dev)
w = ops.PackedWeights(random_hotpath_state_dict(seed=3), device=dev)
ps = ops.PreparedScene(sc["img"], sc["cam"], sc["feat_geo"], sc["feat_tex"], sc["sp_data"], sc["src_foreground_mask"])
N, V = 200_000, 3
lo, hi = sc["bounds"].reshape(2, 3)[0], sc["bounds"].reshape(2, 3)[1]
torch.manual_seed(0)
P = (lo + (hi - lo) * torch.rand(N, 3, device=dev)).contiguous()
D = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1).contiguous()

def rows(mode):
    ops.set_geo_rows_mode(mode)
    out = torch.empty(N, 5, device=dev); valid = torch.empty(N, dtype=torch.uint8, device=dev)
    nb = L.kpn_query_workspace_bytes(N, V)
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    L.check(L.kpn_query(ctypes.byref(ps.desc), ctypes.c_void_p(ps.ws.data_ptr()), ctypes.c_void_p(w.tensor.data_ptr()), N, ctypes.c_void_p(P.data_ptr()),
                        ctypes.c_void_p(D.data_ptr()), 1, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(valid.data_ptr()), ctypes.c_void_p(ws.data_ptr()), nb,
                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    h = ws.cpu().numpy()
    count = int(h[:4].view(np.int32)[0])
    al = lambda b: (b + 255) // 256 * 256
    list_off = 1024
    xs_off = 1024 + al(N * 4)
    lst = h[list_off:list_off + count * 4].view(np.int32)
    ntiles = (count + 31) // 32
    x = h[xs_off:xs_off + ntiles * V * 10 * 64 * 16].view(np.float32).reshape(ntiles, V, 10, 64, 4)
    # per point: slabs 5..9 of both halves -> (N, V, 5, 2, 4)
    per = np.full((N, V, 5, 2, 4), np.nan, np.float32)
    pad = ntiles * 32 - count
    ids = np.concatenate([lst, np.full(pad, -1, np.int32)]).reshape(ntiles, 32)
    sel = x[:, :, 5:10].reshape(ntiles, V, 5, 2, 32, 4)          # lane = h*32 + p
    sel = sel.transpose(0, 4, 1, 2, 3, 5)                          # (tile, p, V, slab, h, 4)
    m = ids >= 0
    per[ids[m]] = sel[m]
    return per, out.cpu().numpy(), ids

ref, out0, _ = rows(int(os.environ.get("REF_MODE", "0")))
for it in range(6):
    got, out1, ids = rows(1)
    d = np.abs(got - ref)
    vec = d[:, :, 0:3].max(axis=(2, 3, 4))      # slabs 5,6,7: the 64-vector's block 1 registers 4..15
    rec = d[:, :, 3:5].max(axis=(2, 3, 4))      # slabs 8,9: gather record
    bad_vec, bad_rec = vec > 1e-2, rec > 1e-4
    print(f"run {it}: rows with wrong vector {int(bad_vec.sum())}, wrong record {int(bad_rec.sum())}, both {int((bad_vec & bad_rec).sum())}; "
          f"points with wrong output {int((np.abs(out1 - out0).max(1) > 1e-2).sum())}")
    if bad_vec.any():
        pts, vs = np.nonzero(bad_vec)
        tiles = {}
        pos = {int(n): (t, p) for t in range(ids.shape[0]) for p, n in enumerate(ids[t]) if n >= 0} if it == 0 else pos
        by = {}
        for n_, v_ in zip(pts[:4000], vs[:4000]):
            by.setdefault((pos[int(n_)][0], int(v_)), []).append(pos[int(n_)][1])
        items = sorted(by.items())[:6]
        print("   (tile, view) -> lanes p affected:", [(k, len(v), min(v), max(v)) for k, v in items])
        n_, v_ = pts[0], vs[0]
        print("   sample row: vec slabs got", got[n_, v_, 0, :, :2].ravel()[:6], "ref", ref[n_, v_, 0, :, :2].ravel()[:6])
ops.set_geo_rows_mode(0)
