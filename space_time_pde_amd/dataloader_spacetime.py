"""On-device space-time crop pipeline: what RB2DataLoader.__getitem__ does on the host with
numpy + scipy (experiments/rb2d/dataloader_spacetime.py:118-171), done on the GPU with the HIP multilinear
interpolation kernel (``stpde_interp_fwd`` via regular_nd_grid_interpolation): random crop, linear down-sampling to
the low-resolution input grid, random query points with trilinearly interpolated targets, channel normalisation.

``RB2DeviceLoader`` is the batched device pipeline (any [4, T, Z, X] tensor or the reference's ``.npz`` with arrays
p, b, u, w of shape [t, x, z], experiments/rb2d/README.md:18-43); ``RB2DataLoader`` is the drop-in ``Dataset`` with the
reference's constructor signature (:18-21) and ``__getitem__`` tuple (:118-171), including ``lres_filter``
(none / gaussian / uniform / median / maximum with scipy.ndimage's 'reflect' boundary, :96-116), ``lres_interp``
(linear / nearest), ``normalize_output`` / ``normalize_hres`` / ``return_hres``.  Both are checked against vectors
produced by the imported reference loader (tests/golden/n3_dataloader.npz).
"""
import os

import numpy as np
import torch

from .regular_nd_grid_interpolation import regular_nd_grid_interpolation


def _reflect_index(n, r, device):
    """Source indices of an axis of length n padded by r on both sides with scipy.ndimage's mode='reflect'
    (d c b a | a b c d | d c b a: the edge sample is repeated), valid for any r (period 2n)."""
    i = torch.arange(-r, n + r, device=device)
    i = torch.remainder(i, 2 * n)
    return torch.where(i >= n, 2 * n - 1 - i, i)


def _correlate_axis(x, w, dim):
    """1-D correlation of x along ``dim`` with the odd-length weight vector w, 'reflect' boundary."""
    r = (w.numel() - 1) // 2
    n = x.shape[dim]
    xp = x.index_select(dim, _reflect_index(n, r, x.device))
    out = torch.zeros_like(x)
    for k in range(w.numel()):
        out = out + w[k] * xp.narrow(dim, k, n)
    return out


def _window_view(x, sizes):
    """x [..., T, Z, X] -> [..., T, Z, X, prod(sizes)] of the reflect-padded neighbourhoods (odd sizes)."""
    for d, sz in zip((-3, -2, -1), sizes):
        r = (sz - 1) // 2
        if r:
            x = x.index_select(x.dim() + d, _reflect_index(x.shape[d], r, x.device))
    t, z, xx = sizes
    v = x.unfold(-3, t, 1).unfold(-3, z, 1).unfold(-3, xx, 1)      # [..., T, Z, X, t, z, x]
    return v.reshape(v.shape[:-3] + (t * z * xx,))


def lres_filter(signal, kind, downsamp_t, downsamp_xz):
    """The reference's pre-filter of the high-res crop (dataloader_spacetime.py:96-116) on a [..., T, Z, X] tensor:
    scipy.ndimage gaussian (sigma = int(downsamp/2) per axis, truncate 4), uniform / median / maximum over a
    (2 downsamp - 1) window, all with the 'reflect' boundary."""
    if kind == 'none' or not kind:
        return signal
    sizes = (downsamp_t * 2 - 1, downsamp_xz * 2 - 1, downsamp_xz * 2 - 1)
    if kind == 'gaussian':
        out = signal
        for dim, sigma in zip((-3, -2, -1), (int(downsamp_t / 2), int(downsamp_xz / 2), int(downsamp_xz / 2))):
            if sigma <= 0:
                continue
            r = int(4.0 * sigma + 0.5)
            k = torch.arange(-r, r + 1, device=signal.device, dtype=torch.float64)
            w = torch.exp(-0.5 * k * k / (sigma * sigma))
            out = _correlate_axis(out, (w / w.sum()).to(signal.dtype), signal.dim() + dim)
        return out
    if kind == 'uniform':
        out = signal
        for dim, sz in zip((-3, -2, -1), sizes):
            if sz > 1:
                out = _correlate_axis(out, torch.full((sz,), 1.0 / sz, device=signal.device, dtype=signal.dtype),
                                      signal.dim() + dim)
        return out
    if kind == 'maximum':
        return _window_view(signal, sizes).amax(dim=-1)
    if kind == 'median':
        return _window_view(signal, sizes).median(dim=-1).values     # window sizes are odd: the exact median
    raise NotImplementedError("lres_filter must be one of none/gaussian/uniform/median/maximum")


class RB2DeviceLoader:
    def __init__(self, data, nx=128, nz=128, nt=16, n_samp_pts_per_crop=1024, downsamp_xz=4, downsamp_t=4,
                 normalize_output=False, device=None, lres_filter='none', lres_interp='linear'):
        if lres_interp not in ('linear', 'nearest'):
            raise ValueError("lres_interp must be 'linear' or 'nearest'")
        self.lres_filter, self.lres_interp = lres_filter, lres_interp
        self.downsamp_xz, self.downsamp_t = downsamp_xz, downsamp_t
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
        self._lres_taps = []
        for a, n in zip(axes, (nt, nz, nx)):
            i0 = torch.clamp(torch.floor(a.double()).long(), 0, max(n - 2, 0))
            a64 = torch.linspace(0, n - 1, a.numel(), device=dev, dtype=torch.float64)
            self._lres_taps.append((i0, (a64 - i0).float()))
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
        if self.lres_filter and self.lres_filter != 'none':
            # the reference interpolates BOTH the low-res grid and the point targets from the filtered crop (:136-155)
            hres = lres_filter(hres.permute(0, 4, 1, 2, 3), self.lres_filter, self.downsamp_t,
                               self.downsamp_xz).permute(0, 2, 3, 4, 1).contiguous()
        if point_coord is None:
            point_coord = torch.rand(B, self.n_samp_pts_per_crop, 3, generator=generator,
                                     device=dev if generator is None or generator.device.type != "cpu" else "cpu").to(dev)
        scale = torch.tensor(self._xmax, device=dev)
        lcoord = self._lres_coord.expand(B, -1, 3).contiguous()
        pcoord = (point_coord * scale).contiguous()
        if self.lres_interp == 'nearest':
            lres, point_value = self._nearest(hres, lcoord), self._nearest(hres, pcoord)
        else:
            # the low-res lattice is structured (linspace per axis, end points ON the crop faces): separable two-tap
            # resampling per axis, exact at the faces (the general-point kernel clips coordinates by 1e-6 of the box like
            # the reference's own interpolation routine does, which scipy's interpolator -- used here by the reference --
            # does not); the random sample points go through the HIP multilinear-interpolation kernel
            lres = hres
            for dim, (i0, w) in zip((1, 2, 3), self._lres_taps):
                lo, hi = lres.index_select(dim, i0), lres.index_select(dim, i0 + 1)
                shape = [1] * lres.dim()
                shape[dim] = -1
                lres = lo + (hi - lo) * w.view(shape)
            point_value = regular_nd_grid_interpolation(hres, pcoord, zeros, self._xmax)
        lres = lres.reshape(B, self.nt_lres, self.nz_lres, self.nx_lres, 4).permute(0, 4, 1, 2, 3).contiguous()
        if self.normalize_output:
            lres = (lres - self._mean.view(1, 4, 1, 1, 1)) / self._std.view(1, 4, 1, 1, 1)
            point_value = (point_value - self._mean) / self._std
        return lres, point_coord, point_value

    @staticmethod
    def _nearest(hres, coord):
        """scipy RegularGridInterpolator(method='nearest') on the unit-spaced crop lattice: node i + 1 when the
        fractional position inside cell i exceeds 0.5, else node i (ties go down)."""
        n = torch.tensor(hres.shape[1:4], device=hres.device)
        i = torch.minimum(torch.clamp(torch.floor(coord), min=0).long(), n - 2)
        idx = torch.where(coord - i <= 0.5, i, i + 1)
        b = torch.arange(hres.shape[0], device=hres.device).view(-1, 1).expand(idx.shape[:2])
        return hres[b, idx[..., 0], idx[..., 1], idx[..., 2]]

    def hres_crop(self, idx):
        """[B, 4, nt, nz, nx] unfiltered high-resolution crops (the reference's return_hres output)."""
        return self._crops(idx).permute(0, 4, 1, 2, 3).contiguous()

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


class RB2DataLoader(torch.utils.data.Dataset):
    """Drop-in for the reference ``RB2DataLoader`` (experiments/rb2d/dataloader_spacetime.py:12-257): same constructor
    arguments, ``__len__``, ``__getitem__`` tuple ([hres,] lres, point_coord, point_value), ``channel_mean`` /
    ``channel_std`` and the (de)normalisation helpers -- but every crop is cut, filtered and interpolated on ``device``
    (the HIP interpolation kernel) and returned as float32 tensors there, so the feeder never touches scipy.

    ``numpy_rng=True`` draws the sample points with ``np.random.rand`` exactly like the reference (:153), so a seeded
    numpy stream reproduces the reference's samples; the default draws them on the device.

    ``device=None`` (default) keeps the Dataset on the HOST like the reference's: it then works unchanged under the
    reference's ``DataLoader(..., num_workers=1, pin_memory=True)`` (experiments/rb2d/train.py:318-321; forked workers
    must not touch the GPU and pin_memory rejects device tensors).  ``device="cuda"`` opts into the on-device pipeline:
    use it with ``num_workers=0, pin_memory=False`` (``__getitem__`` raises inside a worker process), or call
    ``RB2DeviceLoader.get`` directly for whole batches."""

    def __init__(self, data_dir="./", data_filename="./data/rb2d_ra1e6_s42.npz", nx=128, nz=128, nt=16,
                 n_samp_pts_per_crop=1024, downsamp_xz=4, downsamp_t=4, normalize_output=False, normalize_hres=False,
                 return_hres=False, lres_filter='none', lres_interp='linear', device=None, numpy_rng=False):
        self.data_dir, self.data_filename = data_dir, data_filename
        self.normalize_hres, self.return_hres, self.numpy_rng = normalize_hres, return_hres, numpy_rng
        if device is None:
            device = "cpu"
        self._on_device = torch.device(device).type != "cpu"
        self._impl = RB2DeviceLoader(os.path.join(data_dir, data_filename), nx=nx, nz=nz, nt=nt,
                                     n_samp_pts_per_crop=n_samp_pts_per_crop, downsamp_xz=downsamp_xz,
                                     downsamp_t=downsamp_t, normalize_output=normalize_output, device=device,
                                     lres_filter=lres_filter, lres_interp=lres_interp)
        for k in ("nx_hres", "nz_hres", "nt_hres", "nx_lres", "nz_lres", "nt_lres", "n_samp_pts_per_crop",
                  "normalize_output", "scale_hres", "scale_lres", "lres_filter", "lres_interp", "downsamp_xz",
                  "downsamp_t", "data"):
            setattr(self, k, getattr(self._impl, k))

    def __len__(self):
        return len(self._impl)

    def __getitem__(self, idx):
        if self._on_device and torch.utils.data.get_worker_info() is not None:
            raise RuntimeError("RB2DataLoader(device=%r) cannot be used from DataLoader worker processes: pass "
                               "num_workers=0, pin_memory=False, or keep the dataset on the host (device=None)"
                               % (str(self._impl.data.device),))
        pc = None
        if self.numpy_rng:
            pc = torch.from_numpy(np.random.rand(self.n_samp_pts_per_crop, 3).astype(np.float32))[None]
            pc = pc.to(self._impl.data.device)
        lres, pc, pv = self._impl.get([idx], point_coord=pc)
        out = [lres[0], pc[0], pv[0]]
        if self.return_hres:
            hres = self._impl.hres_crop([idx])[0]
            out = [self._impl.normalize_grid(hres) if self.normalize_hres else hres] + out
        return tuple(out)

    @property
    def channel_mean(self):
        return self._impl.channel_mean

    @property
    def channel_std(self):
        return self._impl.channel_std

    def normalize_grid(self, grid):
        return self._impl.normalize_grid(grid)

    def normalize_points(self, points):
        return self._impl.normalize_points(points)

    def denormalize_grid(self, grid):
        return self._impl.denormalize_grid(grid)

    def denormalize_points(self, points):
        return self._impl.denormalize_points(points)
