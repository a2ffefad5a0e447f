"""Isosurface extraction on the GPU: Gaussian gradient magnitude + Lewiner marching cubes (MC33) + vertex look-ups.

Twin of the inlined steps of /root/reference/predict.py:160-181 (== common/marching_cubes_util.py:5-19), with the
error contract of skimage.measure.marching_cubes(method='lewiner'): ValueError when the level is outside
[min, max] (caught by predict.py:188), RuntimeError when no surface is found.
"""
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


class IsoBatchJob:
    """wnf_to_mesh_gpu for a batch in two phases, so that a caller can queue it behind the work that PRODUCES the volumes without a host
    synchronisation: ``enqueue(wnf)`` queues GGM / min-max / MC33 of the whole (B,Q,Q,Q) batch in ONE set of launches on the caller's
    stream (gn_*_batch: a volume per blockIdx.y, ~30 launches per batch, a generous vertex capacity) plus the vertex look-ups on the
    padded buffers, and returns at once; ``finish()`` fetches the B (min, max, #verts, #faces) records in ONE device-to-host copy and
    slices the batch's buffers.  All outputs are this job's own allocations (two jobs in flight never share a buffer)."""

    def __init__(self, Q, iso_surface_level=0.5, sigma=0.5, gradient_direction="ascent", cap_v=None, ggm_fp32=False):
        """cap_v: vertex capacity per volume (None: 6 Q^2, enough for a garment-like closed surface); a volume that needs more is redone
        on its own in finish() -- a caller that sees such volumes regularly passes what the last batch needed (predict._iso_capacity)"""
        if gradient_direction not in ("ascent", "descent"):
            raise ValueError("Incorrect input %s in `gradient_direction`, see docstring." % gradient_direction)
        self.Q, self.level, self.sigma, self.direction = int(Q), float(iso_surface_level), float(sigma), gradient_direction
        self.cap_v = max(4096, int(6 * self.Q ** 2), int(cap_v or 0))
        self.cap_f = 2 * self.cap_v + 64
        self.ggm_bits = 32 if ggm_fp32 else 64       # Arith.ggm_fp32: taps accumulated in fp32 (opt-in; default = scipy's arithmetic bit for bit)
        self.need_v = 0                              # after finish(): the largest vertex / half face count any volume of the batch asked for
        self.vols, self.ggms, self.mcs, self.recs = [], [], [], []
        self.ranges = []                             # per enqueue: the (Bp, 2) device (min, max) records (NaN-propagating: any_nan())
        self.padded, self.max_nv = [], 0             # the (Bp, cap_v, 3) float32 query buffers; largest vertex count of the batch

    def enqueue(self, wnf_part):
        Bp = wnf_part.shape[0]
        if Bp == 0:
            return
        vols = wnf_part.float().contiguous()
        ggm, rng = ops.ggm3d_batch_range(vols, self.sigma, self.ggm_bits)      # the volumes' (min, max) ride on the GGM's staging pass
        mc = ops.mc33_batch(vols, self.level, self.cap_v, self.cap_f)           # verts, faces, normals, values, counts (device)
        rec = torch.cat((rng.double(), mc[4].double()), dim=1)
        self.ranges.append(rng)
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
        self.padded.append(vf32)

    def any_nan(self):
        """device bool: some enqueued volume holds a NaN -- read off the NaN-propagating (min, max) records, no pass over the volumes"""
        return torch.isnan(torch.cat(self.ranges)).any()

    def finish(self):
        """-> list of B entries, each a mesh dict or the exception (ValueError / RuntimeError) scikit-image would have raised"""
        if not self.vols:
            return []
        host = torch.stack(self.recs).cpu().numpy()                    # the one synchronisation
        level = self.level
        out = []
        for b in range(len(self.vols)):
            vmin, vmax, nv, nf = float(host[b, 0]), float(host[b, 1]), int(host[b, 2]), int(host[b, 3])
            self.need_v = max(self.need_v, nv, (nf + 1) // 2)
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
            faces = mc[1][:nf]
            self.max_nv = max(self.max_nv, nv)
            if self.direction == "descent":
                faces = torch.flip(faces, dims=[1])
            out.append(dict(verts=mc[6][:nv], verts_f32=mc[5][:nv], faces=faces, normals=mc[2][:nv], volume_value=mc[3][:nv],
                            volume_gradient_magnitude=mc[7][:nv], ggm=self.ggms[b]))
        return out

    def padded_queries(self):
        """after finish(), one enqueue: -> (B, max_nv, 3) float32, garment b's vertices in rows [0, nv_b) and zeros behind them -- one
        decoder launch for the whole batch's surface queries instead of one per garment; None when not applicable"""
        if len(self.padded) != 1 or self.max_nv == 0 or self.padded[0].shape[0] != len(self.vols):
            return None
        return self.padded[0][:, :self.max_nv].contiguous()


def wnf_batch_to_meshes_gpu(wnf_all, iso_surface_level=0.5, sigma=0.5, gradient_direction="ascent"):
    """wnf_to_mesh_gpu for a whole (B,Q,Q,Q) batch with ONE host synchronisation (IsoBatchJob: enqueue everything, finish).  Same
    results and the same error contract as the one-garment function: -> list of B entries, each a mesh dict or the exception
    (ValueError / RuntimeError) scikit-image would have raised."""
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


def largest_connected_component(mc_faces, num_verts):
    """eval.py:497-503 / :536-540 (igl.adjacency_matrix + igl.connected_components + np.argmax(cc_sizes)): -> is_cc_vert, a bool mask over the
    vertices.  Device kernel (csrc/mesh_cc.hip gn_mesh_largest_component); host tensors are taken to the current GPU and the mask comes back
    to where the faces live."""
    if mc_faces.is_cuda:
        return ops.mesh_largest_component(mc_faces, num_verts)[0]
    mask, _ = ops.mesh_largest_component(mc_faces.cuda(), num_verts)
    return mask.to(mc_faces.device)


def remove_holes(mc_verts, mc_faces, pred_value, value_threshold, extra_verts=()):
    """the reference's hole removal, eval.py:529-548: vertices whose predicted on-surface value exceeds the threshold, the faces made of them
    only (delete_invalid_verts), then the largest connected component of what is left (delete_invalid_verts again).  extra_verts: further
    per-vertex arrays carried through both compactions (eval.py's pred_mc_sim_verts).  -> (verts, faces, [extras...]) of the component.

    Empty result: when no face survives the threshold there is no component to pick -- the reference's np.argmax(cc_sizes) raises
    ``ValueError: attempt to get argmax of an empty sequence`` there (eval.py:540), and so does this function, with that message; callers that
    prefer an empty mesh catch ValueError."""
    on_surface = pred_value > value_threshold
    v1, f1 = delete_invalid_verts(mc_verts, mc_faces, on_surface)
    e1 = [delete_invalid_verts(e, mc_faces, on_surface)[0] for e in extra_verts]
    is_cc = largest_connected_component(f1, v1.shape[0])
    v2, f2 = delete_invalid_verts(v1, f1, is_cc)
    e2 = [delete_invalid_verts(e, f1, is_cc)[0] for e in e1]
    return (v2, f2, *e2)
