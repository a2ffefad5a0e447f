"""Drop-in for the compute slice of /root/reference/predict.py:138-209, and (``main``) for its I/O shell :55-279.

``predict_batch`` runs, for every garment of a batch, exactly the reference's stages
    pointnet2_forward -> unet3d_forward -> (Q,Q,Q) WNF decode -> gaussian_gradient_magnitude -> marching cubes (Lewiner)
    -> nearest-voxel GGM at the vertices -> surface_decoder_forward (warp field) [-> mc_surface_decoder (hole head)]
entirely on the GPU; a ValueError from marching cubes (level outside the volume range) yields the reference's NaN
placeholder mesh (predict.py:165-171,188-189).  Results stay on the device; ``to_host`` converts one result to the
numpy dict predict.py writes to zarr (predict.py:191-209).

The CLI reproduces the reference's config keys (config/predict_default.yaml: main.gpu_id, prediction.volume_size,
gradient_sigma, iso_surface_level, gradient_direction, use_hole_prediction) with argparse.  Inputs: a checkpoint (or seeded synthetic
weights) and either synthetic clouds or a garmentnets dataset store (``--zarr_in``, io/dataset.py); ``--zarr_out`` writes the
reference's ``prediction.zarr`` layout INCLUDING the ground-truth half eval.py reads (gt_marching_cubes_mesh, gt_mesh with the
augmentation rotation, point_cloud/gt_nocs, misc/gt_nocs_grip_point, per-sample attrs: predict.py:120-136,226,236-250,269).
Hydra / wandb are out of scope.
"""
import argparse
import collections
import json
import threading
import time

import numpy as np
import torch

from . import synthetic
from .batch import Batch
from .common import marching_cubes_util as mcu
from .common.torch_util import to_numpy
from .networks.conv_implicit_wnf import ConvImplicitWNFPipeline


def nan_placeholder(device):
    """predict.py:165-170"""
    nan3 = torch.full((1, 3), float("nan"), device=device)
    return dict(verts=nan3.double(), verts_f32=nan3.clone(), faces=torch.zeros((1, 3), dtype=torch.int32, device=device),
                normals=nan3.clone(), volume_value=torch.full((1,), float("nan"), device=device),
                volume_gradient_magnitude=torch.full((1,), float("nan"), device=device), warp_field=nan3.clone())


class _Counter:
    """fp32 re-runs seen by this process (any thread); ``["count"]`` reads it"""

    def __init__(self):
        self._lock, self._n = threading.Lock(), 0

    def bump(self):
        with self._lock:
            self._n += 1
            return self._n

    def __getitem__(self, key):
        if key != "count":
            raise KeyError(key)
        with self._lock:
            return self._n

    def __setitem__(self, key, value):          # tests reset it
        if key != "count":
            raise KeyError(key)
        with self._lock:
            self._n = int(value)


_FALLBACKS = _Counter()
_CAPS_LOCK = threading.Lock()
ISO_CAP_LATTICES = 4         # _iso_capacity remembers at most this many lattice sizes per model


def _iso_capacity(model, Q, needed=None):
    """per-model memory of how many vertices the iso-surfaces of this model's volumes have asked for (keyed by lattice size): the batched
    MC33 launch sizes its output buffers from it (+25 %), so that a checkpoint whose surfaces are larger than the garment-like default
    (6 Q^2 vertices) pays the one-volume-at-a-time redo once, not every batch.  needed: record a finished batch's demand"""
    with _CAPS_LOCK:
        caps = model.__dict__.setdefault("_iso_caps", collections.OrderedDict())
        if needed is not None:
            # grows at once, decays slowly (1/8 of the gap per batch): one outlier batch does not pin tens of MB of output buffers for good
            have = caps.pop(Q, 0)
            caps[Q] = int(needed) if needed >= have else have - (have - int(needed)) // 8
            while len(caps) > ISO_CAP_LATTICES:
                caps.popitem(last=False)
            return None
        return int(caps[Q] * 1.25) + 1024 if Q in caps else None


def _warn_fallback():
    import warnings
    if _FALLBACKS.bump() == 1:
        warnings.warn("garmentnets_amd: NaN in the WNF volume / warp field / hole logits under the split-operand arithmetic (a value left "
                      "fp16's range, or the input holds NaN): re-running the batch with the fp32 kernels")


def predict_batch(model, batch, volume_size=128, iso_surface_level=0.5, gradient_sigma=0.5, gradient_direction="ascent",
                  use_hole_prediction=False, auto_level=False, arith=None):
    """-> list (one per garment) of dicts of device tensors.

    Arithmetic: ``arith`` (default: the model's, ``model.arith``; an immutable arith.Arith passed down every call -- nothing global is
    switched).  The default f16x2 operand split covers fp32's range through power-of-two scales (per output channel, per sample, per
    hidden unit; csrc/unet_split.hip, csrc/decode_split.hip).  What those cannot cover -- a value beyond fp16's range in the scaled
    units -- surfaces as NaN in the WNF volume, the warp field or the hole logits (each decoder has its own weights and scales), never
    as a wrong finite number; such a batch is re-run here with ``arith.strict_fp32()`` (one warning; the count is in
    predict._FALLBACKS)."""
    arith = arith or model.arith
    results, bad = _predict_batch_once(model, batch, volume_size, iso_surface_level, gradient_sigma, gradient_direction, use_hole_prediction, auto_level,
                                       arith.split, arith)
    if arith.split and (results is None or bool(bad)):
        _warn_fallback()
        results, _ = _predict_batch_once(model, batch, volume_size, iso_surface_level, gradient_sigma, gradient_direction, use_hole_prediction, auto_level,
                                         False, arith.strict_fp32())
    return results


def _record_stream(obj, stream):
    """every CUDA tensor reachable from obj (tensor / dict / list / Batch) was allocated on another stream and is about to be used on `stream`"""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)
    elif isinstance(obj, Batch):
        for k in obj.keys:
            _record_stream(getattr(obj, k), stream)


def _dense_phase(model, batch, volume_size, iso_surface_level, gradient_sigma, gradient_direction, auto_level, arith, front=None, device=None):
    """everything that needs no host synchronisation, queued on the current stream: PointNet++ -> gridding + UNet -> WNF lattice ->
    (fixed level) GGM / MC33 of the whole batch -> the batch's grip-point post-processing.  -> state dict for _tail_phase.
    front: a second stream for the batch's FRONT -- the H2D copy of a host batch and PointNet++ (serial farthest-point sampling: 16
    workgroups for 3 ms, a few small GEMMs) run there, i.e. beside whatever the current stream still has queued (the previous batch's UNet);
    the current stream joins at the gridding.  The batch's tensors must be ready for the front stream: a host batch is copied on it, a device
    batch made by Batch.to() carries its own readiness event AS LONG AS its fields are untouched since (Batch.ready_event checks object identity and
    the tensors' version counters), anything else -- incl. a batch whose fields were re-bound or edited in place after .to() -- is ordered behind
    the current stream (correct, no overlap)"""
    with torch.no_grad():
        try:
            if front is None:
                if not batch.pos.is_cuda:
                    batch = batch.to(device, non_blocking=True)
                pointnet2_result = model.pointnet2_forward(batch, prefetch_volume=True)
            else:
                cur = torch.cuda.current_stream(device)
                # (Batch.ready_event: None once a field was re-bound or edited in place after .to() -- that work sits on the caller's stream)
                ready = batch.ready_event() if (batch.pos.is_cuda and hasattr(batch, "ready_event")) else None
                if batch.pos.is_cuda and ready is None:
                    front.wait_stream(cur)
                elif ready is not None:
                    front.wait_event(ready)
                if batch.pos.is_cuda:
                    _record_stream(batch, front)                 # allocated on the caller's stream, read on the front stream
                with torch.cuda.stream(front):
                    if not batch.pos.is_cuda:
                        batch = batch.to(device, non_blocking=True)
                    pointnet2_result = model.pointnet2_forward(batch, prefetch_volume=True)
                    done = torch.cuda.Event()
                    done.record(front)
                cur.wait_event(done)
                _record_stream(pointnet2_result, cur)            # allocated on the front stream, consumed from here on
                _record_stream(batch, cur)
            unet3d_result = model.unet3d_forward(pointnet2_result, arith)
        finally:
            model.volume_agg.drop_prefetch()
        nocs_data = pointnet2_result["nocs_data"]
        B = nocs_data.num_graphs
        wnf_all = model.volume_lattice_forward(unet3d_result, volume_size, arith)["pred_volume"]     # (B,Q,Q,Q)
        # iso-surfaces of the whole batch with one host synchronisation (fixed level; auto_level needs each volume's range first)
        job = None
        if not auto_level:
            job = mcu.IsoBatchJob(volume_size, iso_surface_level, gradient_sigma, gradient_direction, cap_v=_iso_capacity(model, volume_size),
                                  ggm_fp32=arith.ggm_fp32)
            job.enqueue(wnf_all)
        # (read after the batch's own host synchronisation; with a job: off its NaN-propagating range records -- no second pass over the volumes)
        bad = job.any_nan() if job is not None and B > 0 else torch.isnan(wnf_all).any()
        # grip-point post-processing, predict.py:254-274, for the whole batch at once (a handful of launches instead of ~10 per garment)
        bins = model.pointnet2_nocs.nocs_bins
        glog_all = pointnet2_result["global_logits"].reshape(B, bins, 3)
        grip_global = torch.argmax(glog_all, dim=1).to(torch.float32) * (1.0 / (bins - 1))
        conf_global = torch.softmax(glog_all, dim=1)
        sizes = list(nocs_data.sizes)
        grip_nocs = None
        if len(set(sizes)) == 1 and sizes[0] > 0:          # equal clouds: one batched arg-min (ragged batches fall back to per-garment)
            grip_idx = torch.argmin(torch.norm(batch.pos.view(B, sizes[0], 3), dim=2), dim=1) + torch.arange(B, device=batch.pos.device) * sizes[0]
            grip_nocs = nocs_data.pos[grip_idx]
        return dict(pointnet2_result=pointnet2_result, unet3d_result=unet3d_result, wnf_all=wnf_all, job=job, bad=bad,
                    grip_global=grip_global, conf_global=conf_global, grip_nocs=grip_nocs, batch=batch)


def _tail_phase(model, batch, st, iso_surface_level, gradient_sigma, gradient_direction, use_hole_prediction, auto_level, stop_on_nan, arith):
    """the host-synchronising rest, on the current stream: vertex / face counts of the batch (one device-to-host copy), per garment
    the mesh slices, the GGM look-up and the surface decoders.  -> (results, device bool: the WNF, the warp field or the hole logits
    hold a NaN); stop_on_nan: -> (None, True) before the per-garment tail when the WNF does"""
    with torch.no_grad():
        pointnet2_result, unet3d_result, wnf_all, bad = st["pointnet2_result"], st["unet3d_result"], st["wnf_all"], st["bad"]
        nocs_data = pointnet2_result["nocs_data"]
        B = nocs_data.num_graphs
        meshes = st["job"].finish() if st["job"] is not None else None
        if st["job"] is not None:
            _iso_capacity(model, st["job"].Q, st["job"].need_v)
        ptr = np.concatenate([[0], np.cumsum(nocs_data.sizes)])
        results = []
        if stop_on_nan and bool(bad):                # (after the batch's own synchronisation above: no extra stall on the common path)
            return None, True
        grip_global, conf_global, grip_nocs = st["grip_global"], st["conf_global"], st["grip_nocs"]
        # surface decoders: one launch for the whole batch on the padded vertex buffer (rows past a garment's vertex count are zeros and
        # are sliced away) when the iso-surfaces came out of the batched path; per garment otherwise
        q_all = st["job"].padded_queries() if st["job"] is not None else None
        warp_all = hole_all = None
        if q_all is not None:
            warp_all = model.surface_decoder_forward(unet3d_result, q_all, arith)["out_features"]
            bad = bad | torch.isnan(warp_all).any()  # the surface decoders run the split kernel with their OWN weights and scales
            if use_hole_prediction:
                hole_all = model.mc_surface_decoder_forward(unet3d_result, q_all, arith)["out_features"]
                bad = bad | torch.isnan(hole_all).any()
        for b in range(B):
            wnf = wnf_all[b]
            res = dict(wnf_volume=wnf)
            level = iso_surface_level
            if auto_level:
                mm = torch.stack([wnf.min(), wnf.max()]).cpu()
                level = 0.5 * (float(mm[0]) + float(mm[1]))
            try:
                if meshes is None:
                    mesh = mcu.wnf_to_mesh_gpu(wnf, level, gradient_sigma, gradient_direction)
                elif isinstance(meshes[b], Exception):       # ValueError -> placeholder below; RuntimeError propagates, as in predict.py
                    raise meshes[b]
                else:
                    mesh = meshes[b]
                nv = mesh["verts_f32"].shape[0]
                if warp_all is not None and nv <= warp_all.shape[1]:
                    mesh["warp_field"] = warp_all[b, :nv]
                    logits = hole_all[b, :nv, 0] if use_hole_prediction else None
                else:
                    u3_b = unet3d_result.select(b, b + 1)          # the 128-channel volume is never materialised on this path
                    q = mesh["verts_f32"].view(1, -1, 3)
                    mesh["warp_field"] = model.surface_decoder_forward(u3_b, q, arith)["out_features"].view(-1, 3)
                    bad = bad | torch.isnan(mesh["warp_field"]).any()
                    logits = None
                    if use_hole_prediction:
                        logits = model.mc_surface_decoder_forward(u3_b, q, arith)["out_features"].reshape(-1)
                        bad = bad | torch.isnan(logits).any()
                if use_hole_prediction:
                    mesh["is_on_surface_logits"] = logits
                    mesh["is_on_surface"] = logits > 0
                mesh.pop("ggm", None)
                res.update(mesh)
            except ValueError:
                res.update(nan_placeholder(wnf.device))
            sl = slice(int(ptr[b]), int(ptr[b + 1]))
            res.update(pred_nocs=nocs_data.pos[sl], pred_nocs_confidence=nocs_data.pred_confidence[sl],
                       pred_nocs_logits=pointnet2_result["per_point_logits"][sl], input_points=batch.pos[sl], input_rgb=batch.x[sl])
            res.update(pred_global_nocs_grip_point=grip_global[b], pred_global_confidence=conf_global[b],
                       pred_nocs_grip_point=grip_nocs[b] if grip_nocs is not None else nocs_data.pos[sl][torch.argmin(torch.norm(batch.pos[sl], dim=1))],
                       global_feature=pointnet2_result["global_feature"][b])
            results.append(res)
        return results, bad


def _predict_batch_once(model, batch, volume_size, iso_surface_level, gradient_sigma, gradient_direction, use_hole_prediction, auto_level, stop_on_nan, arith):
    """-> (results, device bool: NaN seen); stop_on_nan: -> (None, True) before the per-garment tail when the WNF holds one"""
    st = _dense_phase(model, batch, volume_size, iso_surface_level, gradient_sigma, gradient_direction, auto_level, arith)
    return _tail_phase(model, batch, st, iso_surface_level, gradient_sigma, gradient_direction, use_hole_prediction, auto_level, stop_on_nan, arith)


_TAIL = threading.local()         # per host thread: {device: its tail stream} -- two threads driving PredictJobs never share one


def _tail_stream(device, kind="tail"):
    streams = _TAIL.__dict__.setdefault("streams", {})
    key = (kind, str(device))
    if key not in streams:
        streams[key] = torch.cuda.Stream(device=device)
    return streams[key]


class PredictJob:
    """predict_batch in two halves, for a stream of batches: the constructor queues everything that needs no host synchronisation
    (PointNet++, UNet, WNF lattice, the batch's GGM / MC33 launches on the current stream) and returns; ``finish()`` does the rest --
    the one device-to-host copy of the vertex / face counts, the mesh slices, the surface decoders -- on a tail stream of its own, so
    that it waits for THIS batch's iso-surfaces only, not for whatever the caller has queued on the main stream since.  Begin batch
    k+1, then finish batch k: the latency-bound tail of k (small grids, host round trip) and the 16-workgroup farthest-point sampling
    of k+1 fill each other's idle CUs, and the host's launch gaps disappear behind queued work.  Every job owns its buffers.
    Same results as predict_batch, bit for bit (tests/test_gpu_api.py)."""

    def __init__(self, model, batch, volume_size=128, iso_surface_level=0.5, gradient_sigma=0.5, gradient_direction="ascent",
                 use_hole_prediction=False, arith=None, front_stream=True):
        """batch: on the device, or on the HOST (pinned for a truly asynchronous copy) -- the job then copies it itself, on its front stream.
        front_stream: run the batch's copy + PointNet++ on a second stream, beside the previous batch's UNet (see _dense_phase)"""
        self.model = model
        self.args = (volume_size, iso_surface_level, gradient_sigma, gradient_direction, use_hole_prediction)
        self.arith = arith or model.arith
        self.device = batch.pos.device if batch.pos.is_cuda else model.device
        self.main = torch.cuda.current_stream(self.device)
        front = _tail_stream(self.device, "front") if front_stream else None
        self.state = _dense_phase(model, batch, volume_size, iso_surface_level, gradient_sigma, gradient_direction, False, self.arith, front=front,
                                  device=self.device)
        self.batch = self.state["batch"]            # (the device copy of a host batch)
        self.ready = torch.cuda.Event()
        self.ready.record(self.main)

    def finish(self, host=False):
        """-> what predict_batch returns; host=True: -> [to_host(r) for r in results], copied on the tail stream (the copies wait for this
        batch only, not for the next batch's dense path on the caller's stream)"""
        volume_size, level, sigma, direction, hole = self.args
        split = self.arith.split
        tail = _tail_stream(self.device)
        tail.wait_event(self.ready)
        with torch.cuda.stream(tail):
            results, bad = _tail_phase(self.model, self.batch, self.state, level, sigma, direction, hole, False, split, self.arith)
            nan_seen = split and (results is None or bool(bad))
            if host and not nan_seen:
                results = to_host_batch(results)
                self.state = None
                return results
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(tail)                       # the caller's stream sees finished results; main-stream memory the tail read is safe to recycle
        self.state = None
        if nan_seen:                                # the fp32 re-run of THIS batch: its own arith value travels down its own calls
            _warn_fallback()
            results, _ = _predict_batch_once(self.model, self.batch, volume_size, level, sigma, direction, hole, False, False, self.arith.strict_fp32())
            return to_host_batch(results) if host else results
        for r in results:                           # allocated on the tail stream, consumed on the caller's
            for v in r.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)
        return results


def predict_stream(model, batches, volume_size=128, iso_surface_level=0.5, gradient_sigma=0.5, gradient_direction="ascent",
                   use_hole_prediction=False, arith=None, host=False):
    """generator over an iterable of batches -> the predict_batch result of each, in order, with one batch in flight behind the one
    being finished (PredictJob).  host=True: every batch comes back as [to_host(r) for r in results] -- the numpy mesh dicts predict.py
    writes -- copied to pinned host memory on the tail stream while the next batch's dense path runs"""
    prev = None
    for batch in batches:
        job = PredictJob(model, batch, volume_size, iso_surface_level, gradient_sigma, gradient_direction, use_hole_prediction, arith)
        if prev is not None:
            yield prev.finish(host=host)
        prev = job
    if prev is not None:
        yield prev.finish(host=host)


def _d2h(tensors):
    """device tensors -> numpy arrays through pinned staging buffers (torch's caching host allocator: after the first batch the blocks
    are recycled), all copies queued on the current stream, ONE synchronisation"""
    out, dev = [], None
    for t in tensors:
        if not t.is_cuda:
            out.append(t)
            continue
        dev = t.device
        buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t, non_blocking=True)
        out.append(buf)
    if dev is not None:
        torch.cuda.current_stream(dev).synchronize()
    return [t.numpy() for t in out]


def _mesh_items(res):
    keys = ["verts", "faces", "normals", "volume_value", "volume_gradient_magnitude", "warp_field"]
    keys += [k for k in ("is_on_surface", "is_on_surface_logits") if k in res]
    dts = {"faces": torch.int32, "is_on_surface": torch.bool}
    return keys, [res[k].detach().to(dts.get(k, torch.float32)) for k in keys]


def to_host(res):
    """numpy dict with the dtypes predict.py:191-199 writes (conversions on the device, before the copy)."""
    keys, tensors = _mesh_items(res)
    return dict(zip(keys, _d2h(tensors)))


def to_host_batch(results):
    """[to_host(r) for r in results] with every copy of the batch queued before the ONE synchronisation"""
    items = [_mesh_items(r) for r in results]
    flat = _d2h([t for _, ts in items for t in ts])
    out, i = [], 0
    for keys, ts in items:
        out.append(dict(zip(keys, flat[i:i + len(ts)])))
        i += len(ts)
    return out


def to_host_groups(res, data=None):
    """(marching_cubes_mesh, point_cloud, misc) numpy dicts = the three zarr groups predict.py:211-279 writes per sample.
    data: the (one-garment) input Batch of a dataset sample -- adds its ground truth as the reference does: point_cloud/gt_nocs =
    batch.y (predict.py:226), misc/gt_nocs_grip_point = batch.nocs_grip_point[0] (predict.py:269)"""
    pc = {"pred_nocs": to_numpy(res["pred_nocs"]), "pred_nocs_confidence": to_numpy(res["pred_nocs_confidence"]),
          "pred_nocs_logits": to_numpy(res["pred_nocs_logits"]), "input_points": to_numpy(res["input_points"]),
          "input_rgb": to_numpy((res["input_rgb"] * 255).to(torch.uint8))}
    misc = {}
    if data is not None and hasattr(data, "y"):
        pc["gt_nocs"] = to_numpy(data.y)
    if data is not None and hasattr(data, "nocs_grip_point"):
        misc["gt_nocs_grip_point"] = to_numpy(data.nocs_grip_point)[0]
    misc.update({k: to_numpy(res[k]) for k in ("pred_nocs_grip_point", "pred_global_nocs_grip_point", "pred_global_confidence", "global_feature")})
    return to_host(res), pc, misc


SAMPLE_ATTR_KEYS = ("scale", "gender", "sample_id", "garment_name", "grip_vertex_idx")      # predict.py:124-130


def write_prediction_sample(output_samples_group, group_key, res, data=None, input_group=None, batch_idx=0, compressor="default"):
    """one sample of prediction.zarr, predict.py:120-136,211-279: the predicted marching_cubes_mesh / point_cloud / misc groups and -- for
    a dataset sample (`data` = its one-garment Batch, `input_group` = its group in the INPUT store) -- everything eval.py reads next to
    them (eval.py:58-66,106-110,147-152,193-208): the per-sample attrs, ``gt_marching_cubes_mesh`` (a copy of the input sample's
    ``marching_cube_mesh`` group, chunk files and codec as they are: what zarr.copy does) and ``gt_mesh`` (the input ``mesh`` arrays,
    ``cloth_verts`` rotated by the augmentation matrix the cloud was rotated with)."""
    from .io import zarr_store
    if compressor == "default":              # predict.py:77's Blosc(zstd, 6, BITSHUFFLE) when numcodecs is importable, zlib otherwise
        compressor = zarr_store.default_compressor()
    attrs = {"batch_idx": int(batch_idx)}
    if input_group is not None:
        src = input_group.attrs
        for k in SAMPLE_ATTR_KEYS:
            if k in src:
                attrs[k] = int(src[k]) if k in ("gender", "grip_vertex_idx") else src[k]
    mesh, pc, misc = to_host_groups(res, data)
    g = zarr_store.write_sample(output_samples_group, group_key, mesh, pc, misc, attrs=attrs, compressor=compressor)
    if input_group is not None:
        if "marching_cube_mesh" in input_group:
            zarr_store.copy_group(input_group["marching_cube_mesh"], g, "gt_marching_cubes_mesh")
        rot = np.eye(3, dtype=np.float32)
        if data is not None and hasattr(data, "input_aug_rot_mat"):
            rot = np.squeeze(to_numpy(data.input_aug_rot_mat))
        out_mesh = g.require_group("gt_mesh")
        for key, value in input_group["mesh"].arrays():
            out_mesh.array(key, value @ rot.T if key == "cloth_verts" else value, compressor=compressor)
    return g


def _batches_of(indices, batch_size):
    for i in range(0, len(indices), batch_size):
        yield indices[i:i + batch_size]


def main(argv=None):
    ap = argparse.ArgumentParser(description="GarmentNets predict (MI355X-native): predict.py's loop over a dataset store or synthetic clouds")
    ap.add_argument("--checkpoint_path", default=None, help="Lightning-style .ckpt; default: seeded synthetic weights")
    ap.add_argument("--gpu_id", type=int, default=0)
    ap.add_argument("--volume_size", type=int, default=128)
    ap.add_argument("--gradient_sigma", type=float, default=0.5)
    ap.add_argument("--iso_surface_level", type=float, default=0.5)
    ap.add_argument("--gradient_direction", default="ascent")
    ap.add_argument("--use_hole_prediction", action="store_true")
    ap.add_argument("--auto_level", action="store_true", help="synthetic weights: iso level = mid(min, max) of each WNF volume instead of the fixed level")
    ap.add_argument("--num_samples", type=int, default=None, help="at most this many samples (default: the whole subset of a store; 4 synthetic clouds)")
    ap.add_argument("--num_pc_sample", type=int, default=6000)
    ap.add_argument("--grid", type=int, default=32)
    ap.add_argument("--reduce_method", default="max")
    ap.add_argument("--subset", default="test", choices=("train", "val", "test", "all"),
                    help="prediction.subset of the reference config (predict.py:63-66): the samples of the data module's seeded instance split "
                         "(datamodule.dataset_split / split_seed); val and test read the store with static_epoch_seed=True, as the reference's "
                         "val_dataset does.  'all' = every sample of the store, in key order")
    ap.add_argument("--dataset_split", type=float, nargs=3, default=(8, 1, 1), help="datamodule.dataset_split (predict_default.yaml: [8,1,1])")
    ap.add_argument("--split_seed", type=int, default=0, help="datamodule.split_seed (predict_default.yaml: 0)")
    ap.add_argument("--batch_size", type=int, default=1,
                    help="garments per forward pass.  1 = the reference's loop (predict.py:62 asserts it).  > 1: the batched device tail (one "
                         "marching-cubes / surface-decode launch set per batch); PointConv's bipartite self-loop quirk then links centre i to point i "
                         "of the WHOLE batch, as the reference's forward would at that batch size -- see --self_loop_scope")
    ap.add_argument("--self_loop_scope", default="example", choices=("example", "batch"),
                    help="example (default): PointConv's added self-loop of centre i is point i of the centre's OWN cloud, so every garment of a batch "
                         "gets exactly its batch_size=1 result; batch: PyG's literal behaviour on a batched graph (point i of the concatenated cloud)")
    ap.add_argument("--in_flight", type=int, default=2, choices=(1, 2), help="2 (default): batch k+1's dense path is queued before batch k's tail is "
                                                                              "finished (predict_stream); 1: one batch at a time (predict_batch)")
    ap.add_argument("--out", default=None, help="optional .npz with the last mesh")
    ap.add_argument("--zarr_out", default=None, help="optional prediction.zarr directory (reference group layout, Zarr v2)")
    ap.add_argument("--zarr_in", default=None, help="garmentnets dataset (Zarr v2; Blosc chunks need numcodecs): read the clouds through "
                                                    "io.dataset.GarmentInputDataset instead of synthetic ones; with --zarr_out the ground-truth groups eval.py "
                                                    "reads are written too")
    ap.add_argument("--num_views", type=int, default=4)
    ap.add_argument("--no_augmentation", action="store_true", help="datamodule.enable_augumentation=False (predict_default.yaml: True)")
    ap.add_argument("--random_rot_range", type=float, nargs=2, default=(-180.0, 180.0))
    ap.add_argument("--static_epoch_seed", action="store_true", help="force static_epoch_seed=True for --subset train / all too")
    a = ap.parse_args(argv)
    device = torch.device("cuda:{}".format(a.gpu_id))
    torch.cuda.set_device(device)       # main.gpu_id of the reference config: every allocation, stream and kernel of this process goes there
    if a.checkpoint_path:
        model = ConvImplicitWNFPipeline.load_from_checkpoint(a.checkpoint_path)
    else:
        hp = synthetic.default_hparams(grid=a.grid, reduce_method=a.reduce_method, mc_surface=a.use_hole_prediction)
        model = ConvImplicitWNFPipeline(**hp)
        model.load_state_dict(synthetic.synthetic_state_dict(hp, 0))
    model = model.to(device).eval().requires_grad_(False)
    model.pointnet2_nocs.set_self_loop_scope(a.self_loop_scope)
    dataset = None
    if a.zarr_in:
        from .io.dataset import GarmentInputDataset
        dataset = GarmentInputDataset(a.zarr_in, num_pc_sample=a.num_pc_sample, num_views=a.num_views,
                                      static_epoch_seed=a.static_epoch_seed or a.subset in ("val", "test"),
                                      enable_augumentation=not a.no_augmentation, random_rot_range=tuple(a.random_rot_range),
                                      volume_task_space=model.volume_task_space, dataset_split=tuple(a.dataset_split), split_seed=a.split_seed)
        indices = list(range(len(dataset))) if a.subset == "all" else [int(i) for i in dataset.subset_indices(a.subset)]
    else:
        indices = list(range(4 if a.num_samples is None else a.num_samples))
    if a.num_samples is not None:
        indices = indices[:a.num_samples]
    out_samples = None
    if a.zarr_out:
        from .io import zarr_store
        root = zarr_store.open_group(a.zarr_out)
        root.put_attrs({"subset": a.subset if dataset is not None else "synthetic"})
        out_samples = root.require_group("samples")

    def load(chunk):
        """-> (per-garment (key, one-garment Batch, input group), the batch on the device)"""
        items = []
        for i in chunk:
            if dataset is not None:
                key = dataset.keys[i]
                items.append((key, GarmentInputDataset.collate([dataset[i]]), dataset.samples_group[key]))
            else:
                x, pos, batch = synthetic.synthetic_cloud(1, a.num_pc_sample, seed=i)
                items.append((f"synthetic_{i:05d}", Batch(sizes=[a.num_pc_sample], x=x, pos=pos, batch=batch), None))
        if len(items) == 1:
            return items, items[0][1].to(device)
        sizes = [it[1].sizes[0] for it in items]
        joint = Batch(sizes=sizes, x=torch.cat([it[1].x for it in items]), pos=torch.cat([it[1].pos for it in items]),
                      batch=torch.repeat_interleave(torch.arange(len(items)), torch.tensor(sizes)))
        return items, joint.to(device)

    kw = dict(volume_size=a.volume_size, iso_surface_level=a.iso_surface_level, gradient_sigma=a.gradient_sigma,
              gradient_direction=a.gradient_direction, use_hole_prediction=a.use_hole_prediction)
    pending = collections.deque()

    def inputs():
        for chunk in _batches_of(indices, a.batch_size):
            items, dev_batch = load(chunk)
            pending.append((items, time.time()))
            yield dev_batch

    if a.in_flight == 2 and not a.auto_level:
        results_iter = predict_stream(model, inputs(), **kw)
    else:
        results_iter = (predict_batch(model, b, auto_level=a.auto_level, **kw) for b in inputs())
    last, batch_idx = None, 0
    for results in results_iter:
        items, t0 = pending.popleft()
        for (key, data, input_group), res in zip(items, results):
            last = to_host(res)
            if out_samples is not None:
                write_prediction_sample(out_samples, key, res, data if dataset is not None else None, input_group, batch_idx=batch_idx)
            print(json.dumps({"sample": batch_idx, "key": key, "verts": int(last["verts"].shape[0]), "faces": int(last["faces"].shape[0]),
                              "seconds_since_load": round(time.time() - t0, 4)}))
            batch_idx += 1
    torch.cuda.synchronize()
    if a.out and last is not None:
        np.savez_compressed(a.out, **last)


if __name__ == "__main__":
    main()
