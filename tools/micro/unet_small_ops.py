"""Which torch ops launch the small copy / fill kernels of one U-Net forward + backward (GPU box)?

    python tools/micro/unet_small_ops.py [T Z X]
"""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from space_time_pde_amd import unet3d  # noqa: E402


def main():
    igres = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 128, 128)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = unet3d.UNet3d(in_features=4, out_features=32, igres=igres, nf=16, mf=256).to(dev).train()
    net.deferred_weight_grads = True
    x = torch.randn(1, 4, *igres, device=dev)
    g = torch.randn(1, *igres, 32, device=dev).permute(0, 4, 1, 2, 3)

    def step():
        for p in net.parameters():
            p.grad = None
        y = net(x)
        y.backward(g)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))
    print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=40, max_name_column_width=50,
                                                     max_src_column_width=120))


if __name__ == "__main__":
    main()
