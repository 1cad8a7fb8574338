"""On-device space-time crop pipeline: what RB2DataLoader.__getitem__ does on the host with
numpy + scipy (experiments/rb2d/dataloader_spacetime.py:118-171), done on the GPU with the HIP multilinear
interpolation kernel (``stpde_interp_fwd`` via regular_nd_grid_interpolation): random crop, linear down-sampling to
the low-resolution input grid, random query points with trilinearly interpolated targets, channel normalisation.

``RB2DeviceLoader`` keeps the reference's constructor arguments that matter on this path (nx, nz, nt,
n_samp_pts_per_crop, downsamp_xz, downsamp_t, normalize_output) and its [c, t, z, x] data convention; the dataset
is either the reference's ``.npz`` (arrays p, b, u, w of shape [t, x, z], experiments/rb2d/README.md:18-43) or any
[4, T, Z, X] tensor.  Only lres_filter='none' / lres_interp='linear' (the reference defaults) are implemented.
"""
import numpy as np
import torch

from .regular_nd_grid_interpolation import regular_nd_grid_interpolation


class RB2DeviceLoader:
    def __init__(self, data, nx=128, nz=128, nt=16, n_samp_pts_per_crop=1024, downsamp_xz=4, downsamp_t=4,
                 normalize_output=False, device=None):
        if isinstance(data, str):
            npz = np.load(data)
            arr = np.stack([npz['p'], npz['b'], npz['u'], npz['w']], axis=0).astype(np.float32)
            data = torch.from_numpy(arr.transpose(0, 1, 3, 2).copy())        # [c, t, z, x] (reference :64-67)
        data = torch.as_tensor(data, dtype=torch.float32)
        if device is not None:
            data = data.to(device)
        if data.dim() != 4 or data.shape[0] != 4:
            raise ValueError("data must be [4, T, Z, X] (p, b, u, w)")
        _, nt_d, nz_d, nx_d = data.shape
        if nx > nx_d or nz > nz_d or nt > nt_d:
            raise ValueError('Resolution in each spatial temporal dimension x ({}), z({}), t({})'
                             'must not exceed dataset limits x ({}) z ({}) t ({})'.format(nx, nz, nt, nx_d, nz_d, nt_d))
        if (nt % downsamp_t != 0) or (nx % downsamp_xz != 0) or (nz % downsamp_xz != 0):
            raise ValueError('nx, nz and nt must be divisible by downsamp factor.')
        self.data = data
        self.data_cl = data.permute(1, 2, 3, 0).contiguous()                  # [T, Z, X, c] for the gather kernels
        self.nx_hres, self.nz_hres, self.nt_hres = nx, nz, nt
        self.nx_lres, self.nz_lres, self.nt_lres = nx // downsamp_xz, nz // downsamp_xz, nt // downsamp_t
        self.n_samp_pts_per_crop = n_samp_pts_per_crop
        self.normalize_output = normalize_output
        self.scale_hres = np.array([nt, nz, nx], dtype=np.int32)
        self.scale_lres = np.array([self.nt_lres, self.nz_lres, self.nx_lres], dtype=np.int32)
        self._ranges = (nt_d - nt + 1, nz_d - nz + 1, nx_d - nx + 1)
        self._mean = data.mean(dim=(1, 2, 3))
        self._std = data.std(dim=(1, 2, 3), unbiased=False)
        dev = data.device
        # low-res lattice in high-res index units: linspace(0, n_hres-1, n_lres) per axis (reference :146-150)
        axes = [torch.linspace(0, n - 1, m, device=dev) for n, m in
                zip((nt, nz, nx), (self.nt_lres, self.nz_lres, self.nx_lres))]
        self._lres_coord = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(1, -1, 3)
        self._xmax = tuple(float(n - 1) for n in (nt, nz, nx))

    def __len__(self):
        return self._ranges[0] * self._ranges[1] * self._ranges[2]

    @property
    def channel_mean(self):
        return self._mean

    @property
    def channel_std(self):
        return self._std

    def _crops(self, idx):
        """idx [B] flat crop ids -> high-res crops [B, nt, nz, nx, 4] (channels-last)."""
        idx = torch.as_tensor(idx, device=self.data.device).long().reshape(-1)
        nzr, nxr = self._ranges[1], self._ranges[2]
        t0, z0, x0 = idx // (nzr * nxr), (idx // nxr) % nzr, idx % nxr        # C-order meshgrid (reference :82-86)
        crops = [self.data_cl[t:t + self.nt_hres, z:z + self.nz_hres, x:x + self.nx_hres]
                 for t, z, x in zip(t0.tolist(), z0.tolist(), x0.tolist())]
        return torch.stack(crops, 0).contiguous()

    def get(self, idx, generator=None, point_coord=None):
        """Batch of crops.  Returns (lres [B,4,nt_l,nz_l,nx_l], point_coord [B,N,3] in (0,1), point_value [B,N,4])."""
        hres = self._crops(idx)
        B = hres.shape[0]
        dev = hres.device
        zeros = (0., 0., 0.)
        lres = regular_nd_grid_interpolation(hres, self._lres_coord.expand(B, -1, 3).contiguous(), zeros, self._xmax)
        lres = lres.reshape(B, self.nt_lres, self.nz_lres, self.nx_lres, 4).permute(0, 4, 1, 2, 3).contiguous()
        if point_coord is None:
            point_coord = torch.rand(B, self.n_samp_pts_per_crop, 3, generator=generator,
                                     device=dev if generator is None or generator.device.type != "cpu" else "cpu").to(dev)
        scale = torch.tensor(self._xmax, device=dev)
        point_value = regular_nd_grid_interpolation(hres, (point_coord * scale).contiguous(), zeros, self._xmax)
        if self.normalize_output:
            lres = (lres - self._mean.view(1, 4, 1, 1, 1)) / self._std.view(1, 4, 1, 1, 1)
            point_value = (point_value - self._mean) / self._std
        return lres, point_coord, point_value

    def __getitem__(self, idx):
        lres, pc, pv = self.get([idx])
        return lres[0], pc[0], pv[0]

    def normalize_grid(self, grid):
        shape = (4,) + (1,) * (grid.dim() - 1)
        return (grid - self._mean.view(shape).to(grid.device)) / self._std.view(shape).to(grid.device)

    def denormalize_grid(self, grid):
        shape = (4,) + (1,) * (grid.dim() - 1)
        return grid * self._std.view(shape).to(grid.device) + self._mean.view(shape).to(grid.device)

    def normalize_points(self, points):
        return (points - self._mean.to(points.device)) / self._std.to(points.device)

    def denormalize_points(self, points):
        return points * self._std.to(points.device) + self._mean.to(points.device)
