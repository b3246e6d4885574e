#!/usr/bin/env python
"""Cost-equivalent stand-ins of the reference's two image encoders, for TIMING only (random weights; MEASUREMENT INFRASTRUCTURE,
not part of the product: the encoders stay the caller's PyTorch modules on MIOpen).

The reference's modules cannot travel to the GPU box, so their layer shapes are restated here from configs/zju.json:46-52,74-81
and src/utils.py:216-245 (ResBlkEncoder: ngf 64, 3 stride-2 convolutions, 4 residual blocks at 512 channels, 2 transposed
convolutions, instance norm) and :262-414 (HGFilterV2: one hourglass of depth 4 at 256 channels behind a 7x7 stride-2 stem, group
norm, bicubic up-sampling; outputs 64 channels at 1/8 and the 8-channel "hd" map at 1/2 of the network input), both fed the source
images down-sampled once (ds_geo = ds_tex = 1, src/model.py:653-680).  Parameter count ~28 M like the reference's.

    python scripts/encoder_standin.py            # ms per encoder pass for 3 source views of 512^2 on cuda:0
"""
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


class Res3(nn.Module):
    """Pre-activation block: three 3x3 convolutions (c -> o/2 -> o/4 -> o/4), their outputs concatenated, plus the input."""

    def __init__(self, c, o):
        super().__init__()
        widths = [(c, o // 2), (o // 2, o // 4), (o // 4, o // 4)]
        self.norms = nn.ModuleList(nn.GroupNorm(min(32, a), a) for a, _ in widths)
        self.convs = nn.ModuleList(nn.Conv2d(a, b, 3, padding=1, bias=False) for a, b in widths)
        self.skip = None if c == o else nn.Sequential(nn.GroupNorm(min(32, c), c), nn.ReLU(True), nn.Conv2d(c, o, 1, bias=False))

    def forward(self, x):
        parts, y = [], x
        for n, c in zip(self.norms, self.convs):
            y = c(F.relu(n(y)))
            parts.append(y)
        return torch.cat(parts, 1) + (x if self.skip is None else self.skip(x))


class Hourglass(nn.Module):
    def __init__(self, depth, c):
        super().__init__()
        self.up, self.low_in, self.low_out = Res3(c, c), Res3(c, c), Res3(c, c)
        self.inner = Hourglass(depth - 1, c) if depth > 1 else Res3(c, c)

    def forward(self, x):
        low = self.low_out(self.inner(self.low_in(F.avg_pool2d(x, 2))))
        return self.up(x) + F.interpolate(low, scale_factor=2, mode="bicubic", align_corners=True)


class GeoEncoder(nn.Module):
    def __init__(self, out_ch=64, hd_ch=8):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, stride=2, padding=3), nn.GroupNorm(32, 64), nn.ReLU(True), Res3(64, 128))
        self.hd = nn.Sequential(nn.ConvTranspose2d(128, 32, 3, stride=2, padding=1, output_padding=1, bias=False), nn.GroupNorm(32, 32),
                                nn.ReLU(True), nn.Conv2d(32, hd_ch, 5, padding=2))
        self.body = nn.Sequential(Res3(128, 128), Res3(128, 256), Hourglass(4, 256), Res3(256, 256), nn.Conv2d(256, 256, 1),
                                  nn.GroupNorm(32, 256), nn.ReLU(True), nn.Conv2d(256, out_ch, 1))

    def forward(self, x):
        x = self.stem(x)
        return [self.body(F.avg_pool2d(x, 2)), self.hd(x)]


class TexEncoder(nn.Module):
    def __init__(self, ngf=64, out_ch=8):
        super().__init__()
        n = lambda c: nn.InstanceNorm2d(c)
        layers = [nn.ReplicationPad2d(3), nn.Conv2d(3, ngf, 7), n(ngf), nn.ReLU(True)]
        c = ngf
        for _ in range(3):
            layers += [nn.Conv2d(c, 2 * c, 3, stride=2, padding=1), n(2 * c), nn.ReLU(True)]
            c *= 2
        self.head = nn.Sequential(*layers)
        self.blocks = nn.ModuleList(nn.Sequential(nn.ReplicationPad2d(1), nn.Conv2d(c, c, 3), n(c), nn.ReLU(True), nn.ReplicationPad2d(1),
                                                  nn.Conv2d(c, c, 3), n(c)) for _ in range(4))
        tail = []
        for _ in range(2):
            tail += [nn.ConvTranspose2d(c, c // 2, 3, stride=2, padding=1, output_padding=1), n(c // 2), nn.ReLU(True)]
            c //= 2
        self.tail = nn.Sequential(*tail, nn.ReplicationPad2d(3), nn.Conv2d(c, out_ch, 7))

    def forward(self, x):
        x = self.head(x)
        for b in self.blocks:
            x = x + b(x)
        return self.tail(x)


def encode(geo, tex, img):
    """What attach_geo_feat / attach_tex_feat do with ds_geo = ds_tex = 1 (src/model.py:653-680)."""
    x = 2.0 * F.avg_pool2d(img, 2) - 1.0
    return geo(x), tex(x)


def time_encoders(n_views=3, res=512, steps=10, device="cuda", channels_last=False):
    geo, tex = GeoEncoder().to(device).eval(), TexEncoder().to(device).eval()
    if channels_last:
        geo, tex = geo.to(memory_format=torch.channels_last), tex.to(memory_format=torch.channels_last)
    img = torch.rand(n_views, 3, res, res, device=device)
    if channels_last:
        img = img.contiguous(memory_format=torch.channels_last)
    out = {"parameters": sum(p.numel() for m in (geo, tex) for p in m.parameters())}
    with torch.no_grad():
        for name, fn in (("geo", lambda: geo(2.0 * F.avg_pool2d(img, 2) - 1.0)), ("tex", lambda: tex(2.0 * F.avg_pool2d(img, 2) - 1.0))):
            for _ in range(3):
                r = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                r = fn()
            torch.cuda.synchronize()
            out[name + "_ms"] = (time.perf_counter() - t0) / steps * 1e3
            out[name + "_out"] = [tuple(t.shape) for t in (r if isinstance(r, list) else [r])]
    return out, geo, tex


if __name__ == "__main__":
    import json
    for cl in (False, True):
        r, _, _ = time_encoders(channels_last=cl)
        r["channels_last"] = cl
        print(json.dumps(r))
