"""World-size-2 gloo test of the point-sharded training step (host logic of the N>1 path, CPU tensors ->
generic strategy): losses and every gradient must equal the single-process step."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(seed=0):
    sys.path.insert(0, ROOT)
    from space_time_pde_amd import implicit_net, physics, unet3d
    torch.manual_seed(seed)
    unet = unet3d.UNet3d(in_features=4, out_features=32, igres=(4, 4, 4), nf=4, mf=8)
    unet.eval()   # running-stat BatchNorm: the 1-voxel levels of this toy net are ill-conditioned with batch statistics
    imnet = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=4, activation=torch.nn.Softplus)
    layer = physics.get_rb2_pde_layer(mean=(0.01, 0, 0.02, -0.01), std=(0.05, 0.3, 0.15, 0.12), t_crop=2., z_crop=1.,
                                      x_crop=1., use_continuity=True)
    g = torch.Generator().manual_seed(1)
    crop = torch.randn(1, 4, 4, 4, 4, generator=g)
    pts = 0.05 + 0.9 * torch.rand(1, 64, 3, generator=g)
    tgt = torch.randn(1, 64, 4, generator=g)
    return unet, imnet, layer, crop, pts, tgt


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from space_time_pde_amd.train_step import sharded_step
    unet, imnet, layer, crop, pts, tgt = _build()
    n = pts.shape[1] // world
    sl = slice(rank * n, (rank + 1) * n)
    loss, reg, pde = sharded_step(unet, imnet, layer, crop, pts[:, sl], tgt[:, sl], pts.shape[1], 1.0, 0.0125,
                                  sync_unet_grads=True)
    if rank == 0:
        torch.save(dict(loss=loss, reg=reg, pde=pde, g_im=[p.grad for p in imnet.parameters()],
                        g_un=[p.grad for p in unet.parameters()]), out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    from space_time_pde_amd.train_step import sharded_step
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    # same intra-op thread count as the workers: the deepest UNet levels normalise over 1-2 voxels, where BatchNorm
    # amplifies summation-order rounding by 1/sqrt(eps)
    torch.set_num_threads(1)
    unet, imnet, layer, crop, pts, tgt = _build()
    loss, reg, pde = sharded_step(unet, imnet, layer, crop, pts, tgt, pts.shape[1], 1.0, 0.0125, distributed=False)
    assert abs(got["loss"].item() - loss.item()) < 1e-6 * abs(loss.item())
    assert abs(got["pde"].item() - pde.item()) < 1e-5 * abs(pde.item())
    for a, p in zip(got["g_im"], imnet.parameters()):
        assert (a - p.grad).abs().max() <= 2e-5 * p.grad.abs().max() + 1e-9
    for a, p in zip(got["g_un"], unet.parameters()):
        assert (a - p.grad).abs().max() <= 1e-4 * p.grad.abs().max() + 1e-8


def _worker_dp(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from space_time_pde_amd.train_step import data_parallel_step
    unet, imnet, layer, crop, pts, tgt = _build()
    g = torch.Generator().manual_seed(100 + rank)
    crop = crop + 0.1 * torch.randn(crop.shape, generator=g)          # a different crop per rank
    loss, reg, pde = data_parallel_step(unet, imnet, layer, crop, pts, tgt, 1.0, 0.0125)
    if rank == 0:
        torch.save(dict(loss=loss, g_im=[p.grad for p in imnet.parameters()], g_un=[p.grad for p in unet.parameters()]), out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_averages_gradients(tmp_path):
    """The reference's batch-of-crops split (train_ddp.py:401-406): gradients = mean over ranks of the local ones."""
    sys.path.insert(0, ROOT)
    from space_time_pde_amd.train_step import data_parallel_step
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "rank0_dp.pt")
    mp.spawn(_worker_dp, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.set_num_threads(1)
    acc_im, acc_un, losses = None, None, []
    for rank in range(2):
        unet, imnet, layer, crop, pts, tgt = _build()
        g = torch.Generator().manual_seed(100 + rank)
        crop = crop + 0.1 * torch.randn(crop.shape, generator=g)
        loss, _, _ = data_parallel_step(unet, imnet, layer, crop, pts, tgt, 1.0, 0.0125)
        losses.append(loss.item())
        gi, gu = [p.grad.clone() for p in imnet.parameters()], [p.grad.clone() for p in unet.parameters()]
        acc_im = gi if acc_im is None else [a + b for a, b in zip(acc_im, gi)]
        acc_un = gu if acc_un is None else [a + b for a, b in zip(acc_un, gu)]
    assert abs(got["loss"].item() - sum(losses) / 2) < 1e-6 * abs(sum(losses) / 2)
    for a, b in zip(got["g_im"], acc_im):
        assert (a - b / 2).abs().max() <= 2e-5 * (b / 2).abs().max() + 1e-9
    for a, b in zip(got["g_un"], acc_un):
        assert (a - b / 2).abs().max() <= 1e-4 * (b / 2).abs().max() + 1e-8
