"""Minimal Zarr v2 directory-store writer / reader (pure numpy, no `zarr` dependency) for the `prediction.zarr` contract.

/root/reference/predict.py:74-84,191-279 writes, per sample, the groups
    samples/<key>/marching_cubes_mesh/{verts,faces,normals,volume_value,volume_gradient_magnitude,warp_field[,is_on_surface,
                                        is_on_surface_logits]}
    samples/<key>/point_cloud/{pred_nocs,pred_nocs_confidence,pred_nocs_logits,input_points,input_rgb,gt_nocs}
    samples/<key>/misc/{gt_nocs_grip_point,pred_nocs_grip_point,pred_global_nocs_grip_point,pred_global_confidence,global_feature}
    samples/<key>/gt_marching_cubes_mesh/   (zarr.copy of the input sample's marching_cube_mesh group: copy_group)
    samples/<key>/gt_mesh/{cloth_verts (rotated by the augmentation matrix),cloth_nocs_verts,cloth_faces_tri,...}
    samples/<key>/.zattrs  {scale, gender, sample_id, garment_name, grip_vertex_idx, batch_idx}
(predict.write_prediction_sample assembles a sample; the gt_* parts exist for dataset samples only),
each array stored as ONE chunk (chunks == shape) and read back by eval.py through `zarr` (eval.py:58-102,185-257,904-935).
This writer emits spec-conformant Zarr v2 metadata (`.zgroup`, `.zattrs`, `.zarray`, chunk files "0.0...").  Codecs:
`compressor: null` and the stdlib `zlib` codec (`{"id": "zlib", "level": n}`) are handled here; every other codec (and any filter
chain) goes through `numcodecs.get_codec(config)` when that package is importable -- so on a machine with numcodecs the real
GarmentNets dataset (Blosc chunks) is readable, and ``REFERENCE_COMPRESSOR`` writes predict.py:77's
Blosc(cname='zstd', clevel=6, shuffle=BITSHUFFLE).  ``default_compressor()`` picks that codec when numcodecs is there and zlib
level 1 otherwise (numcodecs is not installable in the offline build image; tests exercise the branch with a stand-in module).
"""
import importlib
import json
import os
import zlib

import numpy as np

# predict.py:77 -- Blosc(cname='zstd', clevel=6, shuffle=Blosc.BITSHUFFLE) as a Zarr v2 codec config (BITSHUFFLE == 2)
REFERENCE_COMPRESSOR = {"id": "blosc", "cname": "zstd", "clevel": 6, "shuffle": 2, "blocksize": 0}


def _numcodecs():
    """the numcodecs module or None (looked up at call time: an environment that gains the package needs no re-import of this one)"""
    try:
        return importlib.import_module("numcodecs")
    except ImportError:
        return None


def default_compressor():
    """the reference's codec when it can be written here, the stdlib one otherwise"""
    return dict(REFERENCE_COMPRESSOR) if _numcodecs() is not None else ("zlib", 1)


def _codec_config(compressor):
    """None | ("zlib", level) | a Zarr v2 codec config dict -> config dict or None"""
    if compressor is None:
        return None
    if isinstance(compressor, dict):
        return dict(compressor)
    cid, level = compressor
    if cid != "zlib":
        raise ValueError(f"compressor {compressor!r}: pass a codec config dict for anything but ('zlib', level)")
    return {"id": "zlib", "level": int(level)}


class _Codec:
    """encode / decode for one codec config: zlib from the standard library, the rest from numcodecs"""

    def __init__(self, config, what=""):
        self.config, self.impl = config, None
        if config is not None and config.get("id") != "zlib":
            nc = _numcodecs()
            if nc is None:
                raise NotImplementedError(f"{what}: codec {config.get('id')!r} needs the numcodecs package (only zlib / uncompressed chunks "
                                          f"are handled without it)")
            self.impl = nc.get_codec(config)

    def encode(self, block):
        """block: a C-contiguous ndarray chunk.  numcodecs codecs get the TYPED array (as zarr hands it over: Blosc takes its typesize -- what
        BITSHUFFLE shuffles over -- from the array's itemsize; a bytes object would be typesize 1: readable, but neither byte-comparable with the
        reference's chunks nor as well compressed)"""
        if self.impl is not None:
            return bytes(self.impl.encode(block))
        raw = block.tobytes()
        if self.config is None:
            return raw
        return zlib.compress(raw, self.config.get("level", 1))

    def decode(self, raw):
        if self.config is None:
            return raw
        if self.impl is not None:
            out = self.impl.decode(raw)
            return out.tobytes() if isinstance(out, np.ndarray) else bytes(out)
        return zlib.decompress(raw)


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
        """merge `attrs` into the group's .zattrs"""
        merged = dict(self.attrs)
        merged.update(attrs)
        _write_json(os.path.join(self.path, ".zattrs"), merged)

    def array(self, name, data, chunks=None, compressor=None, overwrite=True):
        """C-order Zarr v2 array; chunks=None -> one chunk (chunks == data.shape: what the reference's prediction.zarr uses), else a
        chunk grid with full-size (fill-padded) edge chunks; compressor: None, ("zlib", level) or a Zarr v2 codec config dict
        (e.g. REFERENCE_COMPRESSOR; anything but zlib needs numcodecs)"""
        data = np.ascontiguousarray(data)
        apath = os.path.join(self.path, name)
        comp = _codec_config(compressor)
        codec = _Codec(comp, apath)
        os.makedirs(apath, exist_ok=True)
        shape = list(data.shape)
        cshape = shape if chunks is None else [int(c) for c in chunks]
        assert len(cshape) == len(shape)
        fill = 0.0 if data.dtype.kind == "f" else 0
        meta = {"chunks": cshape, "compressor": comp, "dtype": data.dtype.str, "fill_value": fill,
                "filters": None, "order": "C", "shape": shape, "zarr_format": 2}
        _write_json(os.path.join(apath, ".zarray"), meta)
        grid = list(np.ndindex(*[-(-n // c) for n, c in zip(shape, cshape)])) if shape else [()]
        for ci in grid:
            if shape:
                block = np.full(cshape, fill, dtype=data.dtype)
                sl = tuple(slice(i * c, min((i + 1) * c, n)) for i, c, n in zip(ci, cshape, shape))
                block[tuple(slice(0, s.stop - s.start) for s in sl)] = data[sl]
            else:
                block = data
            raw = codec.encode(np.ascontiguousarray(block))
            with open(os.path.join(apath, ".".join(str(i) for i in ci) if ci else "0"), "wb") as f:
                f.write(raw)

    # -- read ----------------------------------------------------------------------------------------------
    def __getitem__(self, name):
        path = os.path.join(self.path, *name.strip("/").split("/"))
        if os.path.exists(os.path.join(path, ".zarray")):
            return _read_array(path)
        if os.path.exists(os.path.join(path, ".zgroup")):
            return Group(path, create=False)
        raise KeyError(name)

    def keys(self):
        return sorted(d for d in os.listdir(self.path) if not d.startswith("."))

    def __contains__(self, name):
        path = os.path.join(self.path, *name.strip("/").split("/"))
        return os.path.exists(os.path.join(path, ".zarray")) or os.path.exists(os.path.join(path, ".zgroup"))

    def arrays(self):
        """(name, numpy array) of every array directly in this group, sorted by name (zarr's Group.arrays())"""
        for k in self.keys():
            if os.path.exists(os.path.join(self.path, k, ".zarray")):
                yield k, _read_array(os.path.join(self.path, k))

    @property
    def attrs(self):
        p = os.path.join(self.path, ".zattrs")
        return json.load(open(p)) if os.path.exists(p) else {}


def _read_array(path):
    """Zarr v2 array -> numpy: any chunk grid (C order, '.' or '/' chunk keys, edge chunks stored full-size, missing chunks =
    fill_value), compressor None / zlib natively, any other codec and any filter chain through numcodecs when importable (reported,
    not guessed, when it is not)."""
    meta = json.load(open(os.path.join(path, ".zarray")))
    shape, chunks = tuple(meta["shape"]), tuple(meta["chunks"])
    dtype = np.dtype(meta["dtype"])
    comp = meta.get("compressor")
    codec = _Codec(comp, path)
    filters = [_Codec(f, path) for f in (meta.get("filters") or [])]
    if meta.get("order", "C") != "C":
        raise NotImplementedError(f"{path}: only C-order chunks are supported")
    sep = meta.get("dimension_separator", ".")
    fill = meta.get("fill_value")
    out = np.full(shape, 0 if fill is None else fill, dtype=dtype)
    if not shape:
        grid = [()]
    else:
        grid = list(np.ndindex(*[-(-n // c) for n, c in zip(shape, chunks)]))
    for ci in grid:
        f = os.path.join(path, sep.join(str(i) for i in ci) if ci else "0")
        if not os.path.exists(f):
            continue
        raw = codec.decode(open(f, "rb").read())
        for flt in reversed(filters):
            raw = flt.decode(raw)
        block = np.frombuffer(raw, dtype=dtype).reshape(chunks if shape else ())
        if not shape:
            out[...] = block
            continue
        sl = tuple(slice(i * c, min((i + 1) * c, n)) for i, c, n in zip(ci, chunks, shape))
        out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]
    return out


CODEC_NOTE = ("written without numcodecs: chunks use Zarr v2's stdlib codec {'id': 'zlib'} (or none) instead of the reference's "
              "Blosc(cname='zstd', clevel=6, shuffle=BITSHUFFLE) (predict.py:77); group / array / dtype / shape layout is the reference's")


def open_group(path, create=True):
    """root group of a store; a store created here records the one deviation from the reference's on-disk format in its .zattrs"""
    fresh = create and not os.path.exists(os.path.join(path, ".zgroup"))
    g = Group(path, create=create)
    if fresh and _numcodecs() is None:
        g.put_attrs({"codec_note": CODEC_NOTE})
    return g


def copy_group(src_group, dst_parent, name):
    """zarr.copy(src_group, dst_parent, name=name, if_exists='replace') for directory stores (predict.py:236-239): the source group's
    metadata and chunk files are copied as they are -- shapes, dtypes, chunking AND codec (Blosc chunks included: nothing is decoded)"""
    import shutil
    dst = os.path.join(dst_parent.path, name)
    if os.path.exists(dst):
        shutil.rmtree(dst)
    shutil.copytree(src_group.path, dst)
    return Group(dst, create=False)


def write_sample(samples_group, key, mesh, point_cloud, misc, attrs=None, compressor="default"):
    """Write one prediction sample in the reference's layout (predict.py:211-279); compressor "default" = default_compressor()."""
    if compressor == "default":
        compressor = default_compressor()
    g = samples_group.require_group(key)
    if attrs:
        g.put_attrs(attrs)
    for gname, data in (("marching_cubes_mesh", mesh), ("point_cloud", point_cloud), ("misc", misc)):
        sub = g.require_group(gname)
        for k, v in data.items():
            sub.array(k, np.asarray(v), compressor=compressor)
    return g
