#!/usr/bin/env python3
"""Cost of the deterministic mode (_lib.deterministic): U-Net forward + backward alone and the whole step, default vs
deterministic, on BASELINE configs[1] and configs[3] sizes.  GPU box:  python tools/det_cost.py > profiles/r6_det_cost.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from space_time_pde_amd import _lib, implicit_net, lig_jet, nonlinearities, physics, unet3d  # noqa: E402
from space_time_pde_amd.train_step import sharded_step  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    out = {}
    for name, igres, prec in (("configs[1]", (32, 128, 128), "fp32"), ("configs[3]", (64, 256, 256), "bf16")):
        torch.manual_seed(1)
        net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=32,
                                 activation=nonlinearities.NONLINEARITIES["softplus"]).to(dev)
        unet = unet3d.UNet3d(in_features=4, out_features=32, igres=igres, nf=16, mf=256).to(dev).train()
        layer = physics.get_rb2_pde_layer(**bench.RB2)
        crop, pts, tgt = bench.make_inputs(1 << 20, dev, igres=igres)
        cot = torch.randn(1, 32, *igres, device=dev)
        lig_jet.set_mlp_precision(prec)
        rec = {}
        for det in (False, True):
            _lib.deterministic = det

            def unet_only():
                for p in unet.parameters():
                    p.grad = None
                unet(crop).backward(cot)

            def step():
                for p in list(unet.parameters()) + list(net.parameters()):
                    p.grad = None
                sharded_step(unet, net, layer, crop, pts, tgt, 1 << 20, 1.0, 0.0125, "l1")

            for fn, key, n in ((unet_only, "unet_fwd_bwd_ms", 10), (step, "step_ms", 5)):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                rec[("det_" if det else "") + key] = round(e0.elapsed_time(e1) / n, 3)
        _lib.deterministic = False
        nw = sum(p.numel() for p in unet.parameters())
        rec["unet_parameters"] = nw
        rec["accumulator_scratch_MB"] = round(48 * nw / 2 ** 20, 1)
        out[name] = rec
        del unet, net, crop, pts, tgt, cot
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
