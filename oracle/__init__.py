"""CPU ORACLE -- test infrastructure, NOT product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  ``garmentnets_amd`` never does (it fails loudly when its HIP library is missing).

Two halves:
  * ``oracle/gn_oracle.c``  -> ``libgn_oracle.so``: plain-C restatement of the index / geometry steps
    (fps, ball query, k-NN interpolation, Gaussian gradient magnitude, Lewiner MC33), see that file's
    header for the reference call sites and pinning status.
  * ``oracle/pipeline.py``: fp32 torch-CPU restatement of the dense steps (MLPs, GroupNorm/Conv3d UNet,
    trilinear sampling + decoder) and of the composition of the whole path, i.e. the ATen ops the
    reference dispatches to on its CPU path.

Pinning: MC33 / GGM are pinned against scikit-image 0.18.3 / scipy goldens; the dense composition is
pinned against goldens produced by the reference's own modules (tests/golden/make_golden_ref.py);
torch_cluster / PyG / torch_scatter ops are "parity unpinned" (packages absent) -- DESIGN.md section 3.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libgn_oracle.so")
    src = os.path.join(_HERE, "gn_oracle.c")
    hdr = os.path.join(_HERE, "..", "garmentnets_amd", "csrc", "mc33_luts.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgn_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libgn_oracle.so")
        override = os.environ.get("GN_ORACLE_LIB")         # tests/test_oracle_asan.py: the sanitizer build of the same source (make -C oracle asan)
        if override:
            so = override
        elif not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        vp, i64, f64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int
        L.gno_fps_count.restype = i64
        L.gno_fps_count.argtypes = [i64, f64]
        L.gno_fps.argtypes = [vp, vp, i32, f64, vp, vp]
        L.gno_ball_query.argtypes = [vp, vp, vp, vp, i32, f64, i32, vp, vp]
        L.gno_knn_interpolate.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
        L.gno_ggm.argtypes = [vp, i64, i64, i64, f64, vp]
        L.gno_mc33.argtypes = [vp, i64, i64, i64, f64, vp, vp, vp, vp, i64, i64, vp, vp]
        L.gno_gather_nn.argtypes = [vp, i64, i64, i64, vp, i64, f64, vp]
        _LIB = L
    return _LIB


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def batch_to_ptr(batch, B=None):
    batch = np.asarray(batch, dtype=np.int64)
    if B is None:
        B = int(batch.max()) + 1 if batch.size else 0
    assert np.all(np.diff(batch) >= 0), "batch vector must be sorted"
    return np.concatenate([[0], np.cumsum(np.bincount(batch, minlength=B))]).astype(np.int64)


def fps(pos, ptr, ratio):
    """-> (idx int64 global indices, out_ptr)"""
    pos = _c(pos, np.float32)
    ptr = _c(ptr, np.int64)
    B = len(ptr) - 1
    L = lib()
    m = sum(L.gno_fps_count(int(ptr[b + 1] - ptr[b]), float(ratio)) for b in range(B))
    idx = np.zeros(m, np.int64)
    optr = np.zeros(B + 1, np.int64)
    rc = L.gno_fps(pos.ctypes.data, ptr.ctypes.data, B, float(ratio), idx.ctypes.data, optr.ctypes.data)
    assert rc == 0
    return idx, optr


def ball_query(pos, ptr, centre_idx, centre_ptr, r, max_nbr=64):
    """-> (nbr (M,max_nbr) int32 global point idx, -1 padded; cnt (M) int32)"""
    pos = _c(pos, np.float32)
    ptr = _c(ptr, np.int64)
    centre_idx = _c(centre_idx, np.int64)
    centre_ptr = _c(centre_ptr, np.int64)
    M = len(centre_idx)
    nbr = np.zeros((M, max_nbr), np.int32)
    cnt = np.zeros(M, np.int32)
    rc = lib().gno_ball_query(pos.ctypes.data, ptr.ctypes.data, centre_idx.ctypes.data, centre_ptr.ctypes.data,
                              len(ptr) - 1, float(r), int(max_nbr), nbr.ctypes.data, cnt.ctypes.data)
    assert rc == 0
    return nbr, cnt


def knn_interpolate(xs, ps, ptr_s, pq, ptr_q, k, return_knn=False):
    xs = _c(xs, np.float32)
    ps = _c(ps, np.float32)
    pq = _c(pq, np.float32)
    ptr_s = _c(ptr_s, np.int64)
    ptr_q = _c(ptr_q, np.int64)
    Nq, C = pq.shape[0], xs.shape[1]
    out = np.zeros((Nq, C), np.float32)
    kidx = np.zeros((Nq, k), np.int32)
    kw = np.zeros((Nq, k), np.float32)
    rc = lib().gno_knn_interpolate(xs.ctypes.data, ps.ctypes.data, ptr_s.ctypes.data, pq.ctypes.data, ptr_q.ctypes.data,
                                   len(ptr_s) - 1, C, int(k), out.ctypes.data, kidx.ctypes.data, kw.ctypes.data)
    assert rc == 0
    return (out, kidx, kw) if return_knn else out


def ggm(vol, sigma):
    vol = _c(vol, np.float32)
    out = np.zeros_like(vol)
    rc = lib().gno_ggm(vol.ctypes.data, *vol.shape, float(sigma), out.ctypes.data)
    assert rc == 0
    return out


def marching_cubes_raw(vol, level):
    """Lewiner MC33 in voxel units. -> verts f32 (V,3) axis order, faces i32 (F,3), normals f32 (V,3), values f32 (V)."""
    vol = _c(vol, np.float32)
    capv = max(1024, vol.size // 4)
    capf = 2 * capv
    while True:
        v = np.zeros((capv, 3), np.float32)
        f = np.zeros((capf, 3), np.int32)
        n = np.zeros((capv, 3), np.float32)
        a = np.zeros(capv, np.float32)
        nv, nf = ctypes.c_int64(), ctypes.c_int64()
        rc = lib().gno_mc33(vol.ctypes.data, *vol.shape, float(level), v.ctypes.data, f.ctypes.data, n.ctypes.data,
                            a.ctypes.data, capv, capf, ctypes.byref(nv), ctypes.byref(nf))
        if rc == 0:
            return v[:nv.value].copy(), f[:nf.value].copy(), n[:nv.value].copy(), a[:nv.value].copy()
        assert rc == 1, rc
        capv, capf = max(capv, nv.value) + 16, max(capf, nf.value) + 16


def marching_cubes(vol, level, spacing=(1.0, 1.0, 1.0), gradient_direction="ascent"):
    """Same contract as skimage.measure.marching_cubes(method='lewiner') as used at predict.py:172-177:
    ValueError if level outside [min,max]; RuntimeError if no surface; verts float64 = float32 verts * spacing."""
    vol = _c(vol, np.float32)
    level = float(level)
    if level < vol.min() or level > vol.max():
        raise ValueError("Surface level must be within volume data range.")
    v, f, n, a = marching_cubes_raw(vol, level)
    if not len(v):
        raise RuntimeError("No surface found at the given iso value.")
    if gradient_direction == "descent":
        f = np.fliplr(f)
    elif gradient_direction != "ascent":
        raise ValueError("Incorrect input %s in `gradient_direction`" % gradient_direction)
    if not np.array_equal(spacing, (1, 1, 1)):
        v = v * np.r_[spacing]
    return v, f, n, a


def gather_nn(vol, verts, spacing):
    """predict.py:179-181: vol[(verts/spacing).astype(uint32)]"""
    vol = _c(vol, np.float32)
    verts = _c(verts, np.float64)
    out = np.zeros(len(verts), np.float32)
    rc = lib().gno_gather_nn(vol.ctypes.data, *vol.shape, verts.ctypes.data, len(verts), float(spacing), out.ctypes.data)
    assert rc == 0, rc
    return out
