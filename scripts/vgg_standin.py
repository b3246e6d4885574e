#!/usr/bin/env python
"""What the reference's VGG perceptual term costs next to the native training step: VGGLoss (reference src/utils.py:750-805) is
vgg19.features[0:21] applied to the rendered patch and to the ground-truth patch, L1 between the four relu slices, 0.5 x the
sum (configs/zju.json:109-118).  The pretrained weights cannot be had offline, so this is a COST-EQUIVALENT stand-in: the same
nine 3x3 convolutions / three max-pools with random weights, fp32, MIOpen through PyTorch-ROCm; forward of both patches + backward
to the rendered one, as training_step does.  python scripts/vgg_standin.py"""
import time

import torch


def vgg19_prefix():
    cfg = [(3, 64), (64, 64), "M", (64, 128), (128, 128), "M", (128, 256), (256, 256), (256, 256), (256, 256), "M", (256, 512)]
    layers, cuts, n = [], [], 0
    for c in cfg:
        if c == "M":
            layers.append(torch.nn.MaxPool2d(2, 2)); n += 1
        else:
            layers += [torch.nn.Conv2d(c[0], c[1], 3, padding=1), torch.nn.ReLU(inplace=False)]; n += 2
    # slices end behind features[1], [6], [11], [20]
    return torch.nn.Sequential(*layers), (2, 7, 12, 21)


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net, cuts = vgg19_prefix()
    net = net.to(dev).eval()
    for p in net.parameters():
        p.requires_grad_(False)
    weights = [1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def feats(x):
        out, h = [], x
        for i, m in enumerate(net):
            h = m(h)
            if i + 1 in cuts:
                out.append(h)
        return out

    for patch in (32, 64):
        x = torch.rand(1, 3, patch, patch, device=dev, requires_grad=True)
        y = torch.rand(1, 3, patch, patch, device=dev)

        def step():
            fx, fy = feats(x), feats(y)
            loss = sum(w * torch.nn.functional.l1_loss(a, b.detach()) for w, a, b in zip(weights, fx, fy))
            (g,) = torch.autograd.grad(0.5 * loss, x)
            return g

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 50
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        print(f"VGG19[0:21] perceptual term, stand-in weights, {patch}x{patch} patch ({patch * patch} rays): "
              f"{(time.perf_counter() - t0) / K * 1e3:.2f} ms per training step (two forwards + backward to the rendered patch)")


if __name__ == "__main__":
    main()
