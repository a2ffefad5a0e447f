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


def delete_invalid_verts(mc_verts, mc_faces, is_vert_on_surface):
    """common/marching_cubes_util.py:38-52 on torch tensors (any device)."""
    keep_face = is_vert_on_surface[mc_faces.long()].all(dim=1)
    faces = mc_faces[keep_face].long()
    used = torch.unique(faces.flatten())
    remap = torch.zeros(mc_verts.shape[0], dtype=torch.long, device=mc_verts.device)
    remap[used] = torch.arange(used.numel(), device=mc_verts.device)
    return mc_verts[used], remap[faces]
