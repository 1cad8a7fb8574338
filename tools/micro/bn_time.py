"""Timing of the fused BatchNorm(+ReLU) kernels on channels-last data: python tools/micro/bn_time.py [N C]
(wall numbers include the host side of the autograd function; use rocprofv3 --kernel-trace for the kernels alone)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from space_time_pde_amd import unet3d
dev = torch.device('cuda:0')
shapes = ((524288, 16), (524288, 32), (131072, 64), (4096, 128), (4194304, 16), (4194304, 32))
if len(sys.argv) > 2:
    shapes = ((int(sys.argv[1]), int(sys.argv[2])),)
for (n, c) in shapes:
    bn = torch.nn.BatchNorm3d(c).to(dev).train()
    x = torch.randn(1, n // 1024, 32, 32, c, device=dev, requires_grad=True)
    g = torch.randn_like(x)
    for it in range(3):
        y = unet3d._bn_act(x, bn, True); y.backward(g)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for it in range(10):
        e[0].record(); y = unet3d._bn_act(x, bn, True); e[1].record(); y.backward(g); e[2].record()
        torch.cuda.synchronize(); tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    byt = n * c * 4
    print("N=%8d C=%3d  fwd %7.1f us (%.2f TB/s on 3 passes)  bwd %7.1f us (%.2f TB/s on 6 passes)" % (n, c, tf * 100, 3 * byt / (tf / 10 * 1e-3) / 1e12, tb * 100, 6 * byt / (tb / 10 * 1e-3) / 1e12))
