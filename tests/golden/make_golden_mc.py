#!/opt/conda/bin/python3.9
"""Golden vectors for the isosurface stage from the reference's own third-party dependencies
(scikit-image 0.18.3 `marching_cubes(method='lewiner')`, scipy `gaussian_gradient_magnitude`), called exactly as
/root/reference/predict.py:160-181 calls them.  Run in the build container:

    /opt/conda/bin/python3.9 tests/golden/make_golden_mc.py

Writes tests/golden/mc_*.npz (inputs + expected outputs, data only).
"""
import hashlib
import os
import warnings

import numpy as np
import scipy.ndimage as ni
from skimage.measure import marching_cubes

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))


def run(vol, level, spacing=(1.0, 1.0, 1.0)):
    v, f, n, a = marching_cubes(vol, level=level, spacing=spacing, gradient_direction="ascent", method="lewiner")
    return v, f.astype(np.int32), n, a


def shell(Q):
    ax = np.arange(Q, dtype=np.float64) / (Q - 1)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    rho = np.sqrt((X - 0.5) ** 2 + (Y - 0.5) ** 2)
    sig = lambda t: 1.0 / (1.0 + np.exp(-t))
    return (sig(80 * (0.3 - rho)) * sig(80 * (0.4 - np.abs(Z - 0.5)))).astype(np.float32)


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


rng = np.random.default_rng(2024)
out = {}

# (1) every sign pattern of a single cell x 8 magnitude seeds, packed: vols (2048,2,2,2); per-case face/vert lists
vols, nfs, nvs, faces, verts, norms, vals = [], [], [], [], [], [], []
for pat in range(256):
    for seed in range(8):
        mag = rng.uniform(0.05, 1.0, 8).astype(np.float32)
        sign = np.array([(pat >> i) & 1 for i in range(8)]) * 2 - 1
        vol = (mag * sign).reshape(2, 2, 2).astype(np.float32)
        vols.append(vol)
        try:
            v, f, n, a = run(vol, 0.0)
        except (RuntimeError, ValueError):
            v = np.zeros((0, 3), np.float32); f = np.zeros((0, 3), np.int32); n = v; a = np.zeros(0, np.float32)
        nfs.append(len(f)); nvs.append(len(v)); faces.append(f); verts.append(v); norms.append(n); vals.append(a)
out.update(cell_vols=np.stack(vols), cell_nf=np.array(nfs, np.int32), cell_nv=np.array(nvs, np.int32),
           cell_faces=np.concatenate(faces), cell_verts=np.concatenate(verts).astype(np.float32),
           cell_normals=np.concatenate(norms).astype(np.float32), cell_values=np.concatenate(vals).astype(np.float32))

# (2) exact-level corners (v == level), 3-valued cells
vols, nfs, faces = [], [], []
for it in range(3000):
    vol = rng.integers(0, 3, (2, 2, 2)).astype(np.float32) * 0.5
    vols.append(vol)
    try:
        v, f, n, a = run(vol, 0.5)
    except (RuntimeError, ValueError):
        f = np.zeros((0, 3), np.int32)
    nfs.append(len(f)); faces.append(f)
out.update(exact_vols=np.stack(vols), exact_nf=np.array(nfs, np.int32), exact_faces=np.concatenate(faces))

# (3) volumes
cases = {
    "noise14": (rng.uniform(0, 1, (14, 14, 14)).astype(np.float32), 0.5),
    "smooth24": (ni.gaussian_filter(rng.normal(size=(24, 24, 24)), 2.0).astype(np.float32), None),
    "aniso": (ni.gaussian_filter(rng.normal(size=(20, 33, 27)), 1.5).astype(np.float32), 0.0),
    "exact12": (rng.integers(0, 3, (12, 12, 12)).astype(np.float32) * 0.5, 0.5),
    "shell32": (shell(32), 0.5),
}
for name, (vol, level) in cases.items():
    if level is None:
        level = float(np.median(vol))
    Q = vol.shape[-1]
    sp = 1 / (Q - 1)
    v, f, n, a = run(vol, level, (sp,) * 3)
    ggm = ni.gaussian_gradient_magnitude(vol, sigma=0.5, mode="nearest")
    idx = (v / sp).astype(np.uint32)
    out.update({f"{name}_vol": vol, f"{name}_level": np.float64(level), f"{name}_verts": v, f"{name}_faces": f,
                f"{name}_normals": n, f"{name}_values": a, f"{name}_ggm": ggm,
                f"{name}_verts_ggm": ggm[idx[:, 0], idx[:, 1], idx[:, 2]]})
    print(name, vol.shape, "V", len(v), "F", len(f))

# (4) checksum-only: BASELINE-size shells (the volume is regenerated analytically by the tests)
for Q in (128,):
    vol = shell(Q)
    sp = 1 / (Q - 1)
    v, f, n, a = run(vol, 0.5, (sp,) * 3)
    ggm = ni.gaussian_gradient_magnitude(vol, sigma=0.5, mode="nearest")
    out.update({f"shell{Q}_nv": np.int64(len(v)), f"shell{Q}_nf": np.int64(len(f)),
                f"shell{Q}_faces_sha": digest(f), f"shell{Q}_verts_sha": digest(v.astype(np.float32)),
                f"shell{Q}_values_sha": digest(a), f"shell{Q}_ggm_sha": digest(ggm),
                f"shell{Q}_faces_head": f[:64], f"shell{Q}_verts_head": v[:64],
                f"shell{Q}_normals_probe": n[::97], f"shell{Q}_vol_sha": digest(vol)})
    print("shell", Q, "V", len(v), "F", len(f))

path = os.path.join(HERE, "mc_golden.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path) // 1024, "KiB")
