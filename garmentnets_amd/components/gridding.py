"""VirtualGrid / ArraySlicer: host-side index helpers with the reference's API
(/root/reference/components/gridding.py:101-298).  The heavy per-point versions of this maths run inside
gn_grid_features / gn_trilinear_sample; these classes exist so that a predict script written against the
reference keeps working (grid points, chunk slices, idx <-> point conversions) and are plain torch on any device.
"""
import numpy as np
import torch


class VirtualGrid:
    def __init__(self, lower_corner=(0, 0, 0), upper_corner=(1, 1, 1), grid_shape=(32, 32, 32), batch_size=8,
                 device=torch.device("cpu"), int_dtype=torch.int64, float_dtype=torch.float32):
        self.lower_corner = tuple(lower_corner)
        self.upper_corner = tuple(upper_corner)
        self.grid_shape = tuple(grid_shape)
        self.batch_size = int(batch_size)
        self.device = device
        self.int_dtype = int_dtype
        self.float_dtype = float_dtype

    # -- small helpers -------------------------------------------------------------------------------------
    def _corners(self, device):
        kw = dict(dtype=self.float_dtype, device=device)
        lc = torch.tensor(self.lower_corner, **kw)
        uc = torch.tensor(self.upper_corner, **kw)
        span = torch.tensor(self.grid_shape, **kw) - 1
        return lc, uc, span

    def _strides(self, with_batch):
        dims = ((self.batch_size,) if with_batch else ()) + self.grid_shape
        st = [1] * len(dims)
        for i in range(len(dims) - 2, -1, -1):
            st[i] = st[i + 1] * dims[i + 1]
        return dims, st

    @property
    def num_grids(self):
        return int(np.prod((self.batch_size,) + self.grid_shape))

    # -- lattice -------------------------------------------------------------------------------------------
    def get_grid_idxs(self, include_batch=True):
        dims, _ = self._strides(include_batch)
        axes = [torch.arange(n, device=self.device, dtype=self.int_dtype) for n in dims]
        return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1)

    def get_grid_points(self, include_batch=True):
        idxs = self.get_grid_idxs(include_batch=include_batch)
        if include_batch:
            idxs = idxs[..., 1:]
        lc, uc, span = self._corners(self.device)
        return idxs.to(self.float_dtype) * ((uc - lc) / span) + (-lc)

    # -- points <-> cells ----------------------------------------------------------------------------------
    def get_points_grid_idxs(self, points, batch_idx=None):
        lc, uc, span = self._corners(self.device)
        cell = ((points + (-lc)) * (span / (uc - lc))).to(dtype=self.int_dtype)   # truncation toward zero
        hi = torch.tensor(self.grid_shape, dtype=self.int_dtype, device=cell.device) - 1
        cell = torch.minimum(torch.clamp(cell, min=0), hi)
        if batch_idx is not None:
            cell = torch.cat([batch_idx.view(*points.shape[:-1], 1).to(dtype=cell.dtype), cell], dim=-1)
        return cell

    def flatten_idxs(self, idxs, keepdim=False):
        n = idxs.shape[-1]
        if n not in (3, 4):
            raise RuntimeError("Invalid shape {}".format(str(idxs.shape)))
        _, st = self._strides(n == 4)
        w = torch.tensor(st, dtype=idxs.dtype, device=idxs.device)
        return (idxs * w).sum(dim=-1, keepdim=keepdim, dtype=idxs.dtype)

    def unflatten_idxs(self, flat_idxs, include_batch=True):
        _, st = self._strides(include_batch)
        if flat_idxs.shape[-1:] == (1,):
            flat_idxs = flat_idxs[..., 0]
        out, rem = [], flat_idxs
        for s in st:
            out.append(torch.div(rem, s, rounding_mode="floor"))
            rem = rem % s
        return torch.stack(out, dim=-1)

    def idxs_to_points(self, idxs):
        if idxs.shape[-1] == 4:
            idxs = idxs[..., 1:]
        elif idxs.shape[-1] != 3:
            raise RuntimeError("Invalid shape {}".format(tuple(idxs.shape)))
        kw = dict(dtype=self.float_dtype, device=idxs.device)
        lc = torch.tensor(self.lower_corner, **kw)
        uc = torch.tensor(self.upper_corner, **kw)
        span = torch.tensor(self.grid_shape, **kw) - 1
        return idxs * ((uc - lc) / span) + lc


def ceil_div(a, b):
    return -(-a // b)


class ArraySlicer:
    """Iterates C-ordered chunk slices over the leading ``len(chunks)`` axes of ``shape``."""

    def __init__(self, shape, chunks):
        assert len(chunks) <= len(shape)
        self.relevent_shape = tuple(shape[:len(chunks)])
        self.chunks = tuple(chunks)
        self.chunk_size = tuple(ceil_div(s, c) for s, c in zip(self.relevent_shape, self.chunks))

    def __len__(self):
        return int(np.prod(self.chunk_size))

    def __getitem__(self, idx):
        if not 0 <= idx < len(self):
            raise IndexError(idx)
        coords = np.unravel_index(idx, self.chunk_size)
        return [slice(int(c * k), int(min(n, (k + 1) * c))) for k, c, n in zip(coords, self.chunks, self.relevent_shape)]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]
