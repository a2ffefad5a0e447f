"""Minimal Zarr v2 directory-store writer / reader (pure numpy, no `zarr` dependency) for the `prediction.zarr` contract.

/root/reference/predict.py:74-84,191-279 writes, per sample, the groups
    samples/<key>/marching_cubes_mesh/{verts,faces,normals,volume_value,volume_gradient_magnitude,warp_field[,is_on_surface,
                                        is_on_surface_logits]}
    samples/<key>/point_cloud/{pred_nocs,pred_nocs_confidence,pred_nocs_logits,input_points,input_rgb[,gt_nocs]}
    samples/<key>/misc/{pred_nocs_grip_point,pred_global_nocs_grip_point,pred_global_confidence,global_feature[,gt_nocs_grip_point]}
each array stored as ONE chunk (chunks == shape) and read back by eval.py through `zarr` (eval.py:58-102,185-257,904-935).
This writer emits spec-conformant Zarr v2 metadata (`.zgroup`, `.zattrs`, `.zarray`, chunk files "0.0...") with
`compressor: null` or the stdlib `zlib` codec (`{"id": "zlib", "level": n}`), both readable by any Zarr v2 implementation;
the reference's Blosc/zstd codec needs `numcodecs`, which is not available offline -- the codec is a storage detail, the
group / array / dtype / shape contract is what eval.py depends on.
"""
import json
import os
import zlib

import numpy as np


def _write_json(path, obj):
    with open(path, "w") as f:
        json.dump(obj, f, indent=4, sort_keys=True)


class Group:
    def __init__(self, path, create=True):
        self.path = path
        if create:
            os.makedirs(path, exist_ok=True)
            zg = os.path.join(path, ".zgroup")
            if not os.path.exists(zg):
                _write_json(zg, {"zarr_format": 2})

    # -- write ---------------------------------------------------------------------------------------------
    def require_group(self, name, overwrite=False):
        g = self
        for part in name.strip("/").split("/"):
            g = Group(os.path.join(g.path, part))
        return g

    def put_attrs(self, attrs):
        _write_json(os.path.join(self.path, ".zattrs"), attrs)

    def array(self, name, data, chunks=None, compressor=None, overwrite=True):
        """one chunk per array (chunks == data.shape), C order; compressor: None or ("zlib", level)"""
        data = np.ascontiguousarray(data)
        apath = os.path.join(self.path, name)
        os.makedirs(apath, exist_ok=True)
        comp = None
        raw = data.tobytes()
        if compressor is not None:
            cid, level = compressor
            assert cid == "zlib"
            comp = {"id": "zlib", "level": int(level)}
            raw = zlib.compress(raw, int(level))
        shape = list(data.shape)
        meta = {"chunks": shape if shape else [], "compressor": comp, "dtype": data.dtype.str, "fill_value": None if data.dtype.kind == "f" else 0,
                "filters": None, "order": "C", "shape": shape, "zarr_format": 2}
        if data.dtype.kind == "f":
            meta["fill_value"] = 0.0
        _write_json(os.path.join(apath, ".zarray"), meta)
        key = ".".join("0" for _ in shape) if shape else "0"
        with open(os.path.join(apath, key), "wb") as f:
            f.write(raw)

    # -- read ----------------------------------------------------------------------------------------------
    def __getitem__(self, name):
        path = os.path.join(self.path, *name.strip("/").split("/"))
        if os.path.exists(os.path.join(path, ".zarray")):
            meta = json.load(open(os.path.join(path, ".zarray")))
            shape = tuple(meta["shape"])
            key = ".".join("0" for _ in shape) if shape else "0"
            raw = open(os.path.join(path, key), "rb").read()
            if meta["compressor"] is not None:
                assert meta["compressor"]["id"] == "zlib"
                raw = zlib.decompress(raw)
            return np.frombuffer(raw, dtype=np.dtype(meta["dtype"])).reshape(shape).copy()
        if os.path.exists(os.path.join(path, ".zgroup")):
            return Group(path, create=False)
        raise KeyError(name)

    def keys(self):
        return sorted(d for d in os.listdir(self.path) if not d.startswith("."))

    @property
    def attrs(self):
        p = os.path.join(self.path, ".zattrs")
        return json.load(open(p)) if os.path.exists(p) else {}


def open_group(path):
    return Group(path)


def write_sample(samples_group, key, mesh, point_cloud, misc, attrs=None, compressor=("zlib", 1)):
    """Write one prediction sample in the reference's layout (predict.py:211-279)."""
    g = samples_group.require_group(key)
    if attrs:
        g.put_attrs(attrs)
    for gname, data in (("marching_cubes_mesh", mesh), ("point_cloud", point_cloud), ("misc", misc)):
        sub = g.require_group(gname)
        for k, v in data.items():
            sub.array(k, np.asarray(v), compressor=compressor)
    return g
