"""Isosurface extraction on the GPU: Gaussian gradient magnitude + Lewiner marching cubes (MC33) + vertex look-ups.

Twin of the inlined steps of /root/reference/predict.py:160-181 (== common/marching_cubes_util.py:5-19), with the
error contract of skimage.measure.marching_cubes(method='lewiner'): ValueError when the level is outside
[min, max] (caught by predict.py:188), RuntimeError when no surface is found.
"""
import os

import numpy as np
import torch

from .. import ops


def marching_cubes(volume, level=None, spacing=(1.0, 1.0, 1.0), gradient_direction="ascent", capacity=None):
    """volume: (n0,n1,n2) float32 CUDA tensor.  Returns device tensors
    (verts float64 (V,3) = float32 voxel verts * spacing, faces int32 (F,3), normals (V,3), values (V), verts_vox float32)."""
    if volume.dim() != 3:
        raise ValueError("Input volume should be a 3D array.")
    if min(volume.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    volume = volume.float().contiguous()
    mm = ops.minmax(volume).cpu()            # one small D2H: the level-range check is part of the contract
    vmin, vmax = float(mm[0]), float(mm[1])
    if level is None:
        level = 0.5 * (vmin + vmax)
    level = float(level)
    if level < vmin or level > vmax:
        raise ValueError("Surface level must be within volume data range.")
    if len(spacing) != 3:
        raise ValueError("`spacing` must consist of three floats.")
    if gradient_direction not in ("ascent", "descent"):
        raise ValueError("Incorrect input %s in `gradient_direction`, see docstring." % gradient_direction)
    cap_v = int(capacity) if capacity else max(4096, int(6 * max(volume.shape) ** 2))
    while True:
        cap_f = 2 * cap_v + 64
        verts, faces, normals, values, counts = ops.mc33(volume, level, cap_v, cap_f)
        nv, nf = [int(c) for c in counts.cpu()]
        if nv <= cap_v and nf <= cap_f:
            break
        cap_v = max(nv, (nf + 1) // 2) + 64
    if nv == 0:
        raise RuntimeError("No surface found at the given iso value.")
    verts_vox, faces, normals, values = verts[:nv], faces[:nf], normals[:nv], values[:nv]
    if gradient_direction == "descent":
        faces = torch.flip(faces, dims=[1])
    sp = torch.tensor(spacing, dtype=torch.float64, device=volume.device)
    verts64 = verts_vox.double() * sp if not np.array_equal(spacing, (1, 1, 1)) else verts_vox
    return verts64, faces, normals, values, verts_vox


def wnf_to_mesh_gpu(wnf_volume, iso_surface_level=0.5, sigma=0.5, gradient_direction="ascent"):
    """predict.py:160-181 for one (Q,Q,Q) volume on the GPU -> dict of device tensors:
    verts (V,3) float64 in [0,1], verts_f32 (the float32 query points of predict.py:184), faces, normals,
    volume_value, volume_gradient_magnitude, ggm (Q,Q,Q)."""
    Q = wnf_volume.shape[-1]
    spacing = 1 / (Q - 1)
    ggm = ops.ggm3d(wnf_volume.float().contiguous(), sigma)
    verts, faces, normals, values, verts_vox = marching_cubes(wnf_volume, iso_surface_level, (spacing,) * 3, gradient_direction)
    return dict(verts=verts, verts_f32=ops.scale_verts(verts_vox, spacing), faces=faces, normals=normals, volume_value=values,
                volume_gradient_magnitude=ops.gather_nn(ggm, verts_vox, spacing), ggm=ggm)


class _IsoGraph:
    """GGM + min/max + MC33 of ONE (Q,Q,Q) volume captured into a HIP graph (about 25 kernel launches and memsets -> one graph
    launch).  One instance per garment slot, each with its own static buffers, so a batch's replays need no copies between them."""

    def __init__(self, Q, level, sigma, cap_v, cap_f, device):
        self.vol = torch.zeros((Q, Q, Q), dtype=torch.float32, device=device)
        self.args = (level, sigma, cap_v, cap_f)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            self._run()                                            # warm-up outside the capture (LUT uploads, allocator)
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.ggm, self.mc, self.rec = self._run()

    def _run(self):
        level, sigma, cap_v, cap_f = self.args
        ggm = ops.ggm3d(self.vol, sigma)
        mm = ops.minmax(self.vol)
        mc = ops.mc33(self.vol, level, cap_v, cap_f)
        return ggm, mc, torch.cat((mm.double(), mc[4].double()))

    def __call__(self, vol):
        self.vol.copy_(vol, non_blocking=True)
        self.graph.replay()
        return self.ggm, self.mc, self.rec


_ISO_GRAPHS = {}


def clear_iso_graphs():
    """drop the cached per-slot graphs and their static buffers (about 5 volumes of Q^3 floats + the MC33 workspace each)"""
    _ISO_GRAPHS.clear()


_ISO_STREAMS = {}


def _iso_streams(device, n):
    pool = _ISO_STREAMS.setdefault(str(device), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


ISO_STREAMS = int(os.environ.get("GARMENTNETS_ISO_STREAMS", "4"))     # concurrent slot-graph replays of wnf_batch_to_meshes_gpu (1 = in stream)
USE_ISO_GRAPHS = True      # wnf_batch_to_meshes_gpu replays a captured graph per garment slot (False: plain launches)
# the default: GGM / min-max / MC33 of the whole batch in ONE set of launches (gn_*_batch: a volume per blockIdx.y, ~25 launches per
# batch instead of ~25 per garment) on the caller's stream -- no slot graphs, no side streams, no copies out of slot buffers.
# GARMENTNETS_ISO_BATCHED=0: the per-garment slot graphs on side streams
ISO_BATCHED = os.environ.get("GARMENTNETS_ISO_BATCHED", "1") == "1"


class IsoBatchJob:
    """wnf_to_mesh_gpu for a batch, in two phases so that a caller can overlap it with the work that PRODUCES the volumes:
    ``enqueue(wnf_part)`` queues the per-garment kernels (GGM, min/max, MC33 with a generous vertex capacity; one HIP-graph replay per
    garment slot, round-robin on a few side streams forked from the caller's stream at that point) for the next garments of the batch
    and returns at once; ``finish()`` joins the streams, fetches the B (min, max, #verts, #faces) records in ONE device-to-host copy and
    queues the per-garment tails (slicing, vertex look-ups).  predict_batch enqueues the first half of the batch, decodes the second
    half's lattice meanwhile, then enqueues that."""

    def __init__(self, Q, iso_surface_level=0.5, sigma=0.5, gradient_direction="ascent", bank=0):
        if gradient_direction not in ("ascent", "descent"):
            raise ValueError("Incorrect input %s in `gradient_direction`, see docstring." % gradient_direction)
        self.Q, self.level, self.sigma, self.direction = int(Q), float(iso_surface_level), float(sigma), gradient_direction
        self.cap_v = max(4096, int(6 * self.Q ** 2))
        self.cap_f = 2 * self.cap_v + 64
        self.vols, self.ggms, self.mcs, self.recs, self.lanes_used = [], [], [], [], []
        self.padded, self.max_nv = [], 0             # batched path: the (Bp, cap_v, 3) float32 query buffers; largest vertex count of the batch
        self.own_buffers = False                     # True: the outputs are this job's own allocations (batched path), not slot buffers
        self.bank = int(bank)                        # a second set of slot buffers for a caller that keeps two batches in flight (predict.PredictJob)

    def enqueue(self, wnf_part):
        B0, Bp = len(self.vols), wnf_part.shape[0]
        dev = wnf_part.device
        if ISO_BATCHED and Bp > 0 and (self.Q ** 3) % 4 == 0:
            vols = wnf_part.float().contiguous()
            ggm = ops.ggm3d_batch(vols, self.sigma)
            mc = ops.mc33_batch(vols, self.level, self.cap_v, self.cap_f)           # verts, faces, normals, values, counts (device)
            rec = torch.cat((ops.minmax_batch(vols).double(), mc[4].double()), dim=1)
            # the vertex look-ups on the padded (Bp, cap_v) buffers, before the counts are known (rows past a garment's count are zeros)
            spacing = 1 / (self.Q - 1)
            vf32 = ops.scale_verts(mc[0].view(-1, 3), spacing).view(Bp, self.cap_v, 3)
            v64 = mc[0].double() * spacing
            vgm = ops.gather_nn_batch(ggm, mc[0], spacing)
            for i in range(Bp):
                self.vols.append(vols[i])
                self.ggms.append(ggm[i])
                self.mcs.append((mc[0][i], mc[1][i], mc[2][i], mc[3][i], mc[4][i], vf32[i], v64[i], vgm[i]))
                self.recs.append(rec[i])
            self.own_buffers = True
            self.padded.append(vf32)
            return
        main = torch.cuda.current_stream(dev)
        # the garments are independent and one 128^3 volume does not fill 256 CUs (GGM / classify / scan / emit are small grids with
        # dependent launches in between): the slot graphs are replayed round-robin on a few side streams
        lanes = _iso_streams(dev, min(ISO_STREAMS, max(Bp, 1))) if (USE_ISO_GRAPHS and ISO_STREAMS > 1 and Bp > 1
                                                                   and not torch.cuda.is_current_stream_capturing()) else []
        ready = None
        if lanes:
            ready = torch.cuda.Event()
            ready.record(main)
        for i in range(Bp):
            b = B0 + i
            vol = wnf_part[i].float().contiguous()
            self.vols.append(vol)
            if USE_ISO_GRAPHS:                                         # results live in slot b's static buffers until its next replay
                key = (b, self.Q, self.level, self.sigma, self.cap_v, str(vol.device), self.bank)
                if key not in _ISO_GRAPHS:
                    if len(_ISO_GRAPHS) >= 256:
                        _ISO_GRAPHS.clear()
                    _ISO_GRAPHS[key] = _IsoGraph(self.Q, self.level, self.sigma, self.cap_v, self.cap_f, vol.device)
                if lanes:
                    st = lanes[b % len(lanes)]
                    st.wait_event(ready)
                    with torch.cuda.stream(st):
                        ggm, mc, rec = _ISO_GRAPHS[key](vol)
                    if st not in self.lanes_used:
                        self.lanes_used.append(st)
                else:
                    ggm, mc, rec = _ISO_GRAPHS[key](vol)
            else:
                ggm = ops.ggm3d(vol, self.sigma)
                mc = ops.mc33(vol, self.level, self.cap_v, self.cap_f)  # verts, faces, normals, values, counts (device)
                rec = torch.cat((ops.minmax(vol).double(), mc[4].double()))
            self.ggms.append(ggm)
            self.mcs.append(mc)
            self.recs.append(rec)

    def finish(self):
        """-> list of B entries, each a mesh dict or the exception (ValueError / RuntimeError) scikit-image would have raised"""
        if not self.vols:
            return []
        main = torch.cuda.current_stream(self.vols[0].device)
        for st in self.lanes_used:
            main.wait_stream(st)
        host = torch.stack(self.recs).cpu().numpy()                    # the one synchronisation
        spacing, level = 1 / (self.Q - 1), self.level
        out = []
        for b in range(len(self.vols)):
            vmin, vmax, nv, nf = float(host[b, 0]), float(host[b, 1]), int(host[b, 2]), int(host[b, 3])
            if level < vmin or level > vmax:
                out.append(ValueError("Surface level must be within volume data range."))
                continue
            if nv > self.cap_v or nf > self.cap_f:                     # rare: redo this garment with room for what it needs
                try:
                    out.append(wnf_to_mesh_gpu(self.vols[b], level, self.sigma, self.direction))
                except (ValueError, RuntimeError) as e:
                    out.append(e)
                continue
            if nv == 0:
                out.append(RuntimeError("No surface found at the given iso value."))
                continue
            mc = self.mcs[b]
            verts_vox, faces, normals, values = mc[0][:nv], mc[1][:nf], mc[2][:nv], mc[3][:nv]
            if self.own_buffers:                       # batched path: everything is a slice of the batch's buffers, no launch per garment
                self.max_nv = max(self.max_nv, nv)
                if self.direction == "descent":
                    faces = torch.flip(faces, dims=[1])
                out.append(dict(verts=mc[6][:nv], verts_f32=mc[5][:nv], faces=faces, normals=normals, volume_value=values,
                                volume_gradient_magnitude=mc[7][:nv], ggm=self.ggms[b]))
                continue
            if USE_ISO_GRAPHS and not self.own_buffers:   # slot b's static buffers are overwritten by the next replay: hand out copies (a few MB per garment)
                faces, normals, values = faces.clone(), normals.clone(), values.clone()
            if self.direction == "descent":
                faces = torch.flip(faces, dims=[1])
            out.append(dict(verts=verts_vox.double() * spacing, verts_f32=ops.scale_verts(verts_vox, spacing), faces=faces, normals=normals,
                            volume_value=values, volume_gradient_magnitude=ops.gather_nn(self.ggms[b], verts_vox, spacing), ggm=self.ggms[b]))
        return out


    def padded_queries(self):
        """after finish(), batched path with one enqueue: -> (B, max_nv, 3) float32, garment b's vertices in rows [0, nv_b) and zeros
        behind them -- one decoder launch for the whole batch's surface queries instead of one per garment; None when not applicable"""
        if not self.own_buffers or len(self.padded) != 1 or self.max_nv == 0:
            return None
        return self.padded[0][:, :self.max_nv].contiguous()


def wnf_batch_to_meshes_gpu(wnf_all, iso_surface_level=0.5, sigma=0.5, gradient_direction="ascent"):
    """wnf_to_mesh_gpu for a whole (B,Q,Q,Q) batch with ONE host synchronisation (IsoBatchJob: enqueue everything, finish).  Every
    returned tensor is the caller's own (copied out of the slot buffers) except 'ggm', the (Q,Q,Q) gradient-magnitude volume, which
    stays a view of slot b's buffer until the next call with the same (slot, Q, level, sigma) -- clone it to keep it.  Same results and
    the same error contract as the one-garment function: -> list of B entries, each a mesh dict or the exception (ValueError /
    RuntimeError) scikit-image would have raised."""
    job = IsoBatchJob(wnf_all.shape[-1], iso_surface_level, sigma, gradient_direction)
    job.enqueue(wnf_all)
    return job.finish()


def delete_invalid_verts(mc_verts, mc_faces, is_vert_on_surface):
    """common/marching_cubes_util.py:38-52.  Device tensors: the scan-based compaction kernel (csrc/iso.hip gn_mesh_compact; faces come
    back in the dtype they came in).  Host tensors (eval-side use on arrays read back from prediction.zarr): the same thing in torch."""
    if mc_verts.is_cuda:
        v, f = ops.mesh_compact(mc_verts, mc_faces, is_vert_on_surface)
        return v, f.to(mc_faces.dtype)
    keep_face = is_vert_on_surface[mc_faces.long()].all(dim=1)
    faces = mc_faces[keep_face].long()
    used = torch.unique(faces.flatten())
    remap = torch.zeros(mc_verts.shape[0], dtype=torch.long, device=mc_verts.device)
    remap[used] = torch.arange(used.numel(), device=mc_verts.device)
    return mc_verts[used], remap[faces]
