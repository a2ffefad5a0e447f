"""PointNet++ set-abstraction / feature-propagation modules on HIP kernels.

API twin of /root/reference/components/pointnet2.py:11-76 (SAModule / GlobalSAModule / FPModule, attribute names
``conv.local_nn`` / ``nn`` kept for checkpoint compatibility).  The third-party operators the reference calls
(torch_cluster.fps / radius / knn, PyG PointConv / global_max_pool / knn_interpolate) are replaced by
gn_fps, gn_ball_query, gn_sa_gather + gn_linear + gn_segment_max, gn_global_max_pool, gn_knn_interpolate.

``batch`` arguments may be the reference's int64 batch vector or a ``Segments`` object (same information plus
host-side sizes, so that no device synchronisation is needed on the hot path).
"""
import os
import threading
import weakref

import numpy as np
import torch

from .. import ops

FUSED_SA = os.environ.get("GARMENTNETS_FUSED_SA", "1") != "0"      # False: the unfused gather -> gn_linear x3 -> segment-max chain

class _DeviceTableCache:
    """key -> small immutable device int32 table, built once from host data.  Batch shapes repeat from step to step: no per-step
    host-to-device copy, and nothing that a HIP-graph capture of the forward pass could not record (garmentnets_amd/graphs.py).
    Shared by the host threads of a process behind a lock (entries are never mutated after insertion)."""

    def __init__(self, limit=256):
        self._lock, self._tables, self._limit = threading.Lock(), {}, limit

    def get(self, key, build):
        with self._lock:
            t = self._tables.get(key)
            if t is None:
                if len(self._tables) >= self._limit:
                    self._tables.clear()
                t = self._tables[key] = build()
            return t


_DEBUG = threading.local()                # per host thread: SAModule -> its last graph (SAModule.last_graph)
_PTR_CACHE = _DeviceTableCache()          # (sizes, device) -> CSR ptr
_SELF_SRC_CACHE = _DeviceTableCache()     # (point sizes, centre sizes, device) -> per-example self-loop sources


def _csr_ptr(sizes, device):
    return _PTR_CACHE.get((tuple(sizes), str(device)),
                          lambda: torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).to(device))


def _example_self_src(sizes, centre_sizes, device):
    """int32 [M]: for centre c = centre_ptr[b] + i of example b, the point ptr[b] + i -- "point i" of the centre's OWN cloud
    (include/garmentnets_hip.h: gn_sa_fused_scoped)"""
    def build():
        starts = np.concatenate([[0], np.cumsum(sizes)])[:-1]
        return torch.tensor(np.concatenate([s + np.arange(m) for s, m in zip(starts, centre_sizes)] or [np.zeros(0)]), dtype=torch.int32).to(device)
    return _SELF_SRC_CACHE.get((tuple(sizes), tuple(centre_sizes), str(device)), build)


class Segments:
    """Sorted batch vector in CSR form: host sizes, device int32 ptr, lazily materialised int64 batch vector."""

    def __init__(self, sizes, device, batch=None):
        self.sizes = [int(s) for s in sizes]
        self.device = device
        self.ptr = _csr_ptr(self.sizes, device)
        self._batch = batch

    @staticmethod
    def of(batch, sizes=None):
        if isinstance(batch, Segments):
            return batch
        if sizes is None:
            n = int(batch.max().item()) + 1 if batch.numel() else 0
            sizes = torch.bincount(batch, minlength=n).cpu().tolist()
        return Segments(sizes, batch.device, batch)

    @property
    def num(self):
        return len(self.sizes)

    @property
    def total(self):
        return int(sum(self.sizes))

    @property
    def batch(self):
        if self._batch is None:
            self._batch = torch.repeat_interleave(torch.arange(self.num, device=self.device),
                                                  torch.tensor(self.sizes, device=self.device))
        return self._batch


class PointConv(torch.nn.Module):
    """Parameter holder with PyG's attribute name (``local_nn``); evaluated by SAModule."""

    def __init__(self, local_nn=None, global_nn=None, add_self_loops=True):
        super().__init__()
        self.local_nn = local_nn
        self.global_nn = global_nn
        self.add_self_loops = add_self_loops
        # "batch": PyG's literal rule on the batched bipartite graph (centre i <-> point i of the CONCATENATED cloud: a garment's result
        # depends on its slot); "example": point i of the centre's own cloud, i.e. what a batch of one gives every garment
        self.self_loop_scope = "batch"


class SAModule(torch.nn.Module):
    """fps -> ball query (<=64, first in index order) -> PointConv(local_nn, max) -- components/pointnet2.py:22-33."""

    def __init__(self, ratio, r, nn, random_start=False):
        super().__init__()
        self.ratio = ratio
        self.r = r
        self.conv = PointConv(nn)
        # torch_cluster.fps defaults to random_start=True (the reference is therefore non-deterministic); the build pins
        # False (first point of each example) and exposes the upstream behaviour as a flag.
        self.random_start = random_start

    def forward(self, x, pos, batch):
        seg = Segments.of(batch)
        out_sizes = [ops.fps_count(n, self.ratio) for n in seg.sizes]
        cseg = Segments(out_sizes, pos.device)
        start = None
        if self.random_start:
            start = torch.tensor([int(torch.randint(0, max(n, 1), (1,))) for n in seg.sizes], dtype=torch.int32).to(pos.device)
        # a cascade's second level samples the points the first one selected, in selection order: farthest-point order is nested, so its sample
        # is a prefix (include/garmentnets_hip.h gn_fps_nested; decided per example on the device from the first level's smallest running maximum)
        gap = nested = None
        if start is None and seg.num > 0:
            gap = torch.zeros(seg.num, dtype=torch.float32, device=pos.device)      # (0 = "not nested": an example the kernel returns from early leaves it)
            src = getattr(pos, "_fps_cascade", None)
            if src is not None and src[1] == seg.sizes and src[2] == pos._version and src[0].device == pos.device:
                nested = src[0]
        idx = ops.fps(pos, seg.ptr, cseg.ptr, max(seg.sizes) if seg.sizes else 0, cseg.total, start, gap_out=gap, nested_gap=nested)
        nbr, cnt = ops.ball_query(pos, seg.ptr, idx, cseg.ptr, self.r, 64)
        pack = self._fused_pack() if FUSED_SA else None
        self_src = None
        if self.conv.add_self_loops and self.conv.self_loop_scope == "example" and seg.num > 1:
            self_src = _example_self_src(seg.sizes, out_sizes, pos.device)
        if pack is not None and ((x is None and pack.cin == 0) or (x is not None and x.shape[1] == pack.cin)):
            # one kernel: gather -> edge MLP on the matrix cores -> BatchNorm -> max; no edge tensor in HBM (csrc/sa_fused.hip)
            out = ops.sa_fused(x, pos, idx, nbr, cnt, pack, self_loops=self.conv.add_self_loops, self_src=self_src)
        else:
            edges, slot_src, S = ops.sa_gather(x, pos, idx, nbr, self_loops=self.conv.add_self_loops, self_src=self_src)
            h = self.conv.local_nn(edges)
            out = ops.segment_max(h, slot_src, cseg.total, S)
        _DEBUG.__dict__.setdefault("graphs", weakref.WeakKeyDictionary())[self] = (idx, nbr)
        pos_out = pos[idx.long()]
        if gap is not None:
            pos_out._fps_cascade = (gap, out_sizes, pos_out._version)     # lives on THIS tensor object only: any op on it yields a tensor without it
        return out, pos_out, cseg

    @property
    def last_graph(self):
        """(fps indices, ball-query table) of THIS THREAD's last forward through this module -- a debugging / test aid, not read on the call path"""
        return _DEBUG.__dict__.get("graphs", {}).get(self)

    def _fused_pack(self):
        """SaFusedPack of local_nn when it is one of the edge MLPs gn_sa_fused is instantiated for, else None (cached per parameter version)"""
        nn_ = self.conv.local_nn
        blocks = list(nn_) if nn_ is not None else []
        if len(blocks) != 3:
            return None
        dims = [b[0].out_features for b in blocks]
        cin = blocks[0][0].in_features - 3
        if not ops.sa_fused_supported(cin, dims):
            return None
        key = tuple((p._version, p.device) for p in nn_.parameters()) + tuple(b._version for b in nn_.buffers())
        cached = self.__dict__.get("_sa_pack")
        if cached is None or cached[0] != key:
            from .mlp import fold_batchnorm
            layers = []
            for b in blocks:
                sc, sh = fold_batchnorm(b[2]) if len(b) > 2 else (None, None)
                layers.append((b[0].weight, b[0].bias, sc, sh))
            cached = (key, ops.pack_sa_fused(layers).to(blocks[0][0].weight.device))
            self.__dict__["_sa_pack"] = cached
        return cached[1]


class GlobalSAModule(torch.nn.Module):
    """MLP(cat[x, pos]) -> per-example max -- components/pointnet2.py:44-52."""

    def __init__(self, nn):
        super().__init__()
        self.nn = nn

    def forward(self, x, pos, batch):
        seg = Segments.of(batch)
        c = x.shape[1]
        buf = ops.new_rows(x.shape[0], c + 3, x.device)
        buf[:, :c] = x
        buf[:, c:] = pos
        h = self.nn(buf)
        out = ops.global_max_pool(h, seg.ptr, seg.num)
        return out, pos.new_zeros((seg.num, 3)), Segments([1] * seg.num, pos.device)


class FPModule(torch.nn.Module):
    """knn_interpolate -> cat skip -> MLP -- components/pointnet2.py:70-76."""

    def __init__(self, k, nn):
        super().__init__()
        self.k = k
        self.nn = nn

    def forward(self, x, pos, batch, x_skip, pos_skip, batch_skip):
        seg, seg_skip = Segments.of(batch), Segments.of(batch_skip)
        c = x.shape[1]
        cs = 0 if x_skip is None else x_skip.shape[1]
        buf = ops.new_rows(pos_skip.shape[0], c + cs, x.device)
        ops.knn_interpolate(x, pos.contiguous(), seg.ptr, pos_skip.contiguous(), seg_skip.ptr, self.k, out=buf[:, :c])
        if x_skip is not None:
            buf[:, c:] = x_skip
        return self.nn(buf), pos_skip, seg_skip
