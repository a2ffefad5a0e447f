"""Input side of the hot path: garment sample -> view / point sub-sampling -> noise / z-rotation
augmentation -> Batch, as ConvImplicitWNFDataset does for inference (/root/reference/datasets/conv_implicit_wnf_dataset.py:
data_io 134-181, get_base_data 183-229, rotation_augumentation 370-406, noise_augumentation 408-423, __getitem__ 431-461).

Random streams are the reference's: np.random.RandomState(seed = idx when static_epoch_seed else None), drawn in the same order
(view choice, point choice; one fresh stream per augmentation), so a given (idx, static_epoch_seed=True) yields the same points.
Volume / surface / marching-cubes-surface query sampling (training targets) is not on the inference path and is not here.
Reads the dataset's Zarr v2 layout through garmentnets_amd.io.zarr_store (uncompressed or zlib chunks; Blosc needs numcodecs).
"""
import numpy as np
import torch
from scipy.spatial.transform import Rotation

from ..batch import Batch
from . import zarr_store


def data_io(sample_group):
    """one sample group -> the arrays the inference path needs (data_io: 134-181, without the volume / MC-surface targets)"""
    attrs = sample_group.attrs
    pc, mesh = sample_group["point_cloud"], sample_group["mesh"]
    return {
        "cloth_sim_verts": mesh["cloth_verts"][:],
        "cloth_nocs_verts": mesh["cloth_nocs_verts"][:],
        "cloth_faces_tri": mesh["cloth_faces_tri"][:],
        "pc_nocs": pc["nocs"][:],
        "pc_sim": pc["point"][:],
        "pc_sim_rgb": pc["rgb"][:],
        "pc_sizes": pc["sizes"][:],
        "scale": attrs["scale"],
        "grip_vertex_idx": attrs["grip_vertex_idx"],
    }


def get_base_data(idx, data_in, num_pc_sample=6000, num_views=4, static_epoch_seed=False, cloth_sim_aabb=None):
    """get_base_data: 183-229"""
    rs = np.random.RandomState(seed=idx if static_epoch_seed else None)
    all_idxs = np.arange(len(data_in["pc_sim"]))
    all_num_views = len(data_in["pc_sizes"])
    if num_views < all_num_views:
        idxs_mask = np.zeros_like(all_idxs, dtype=bool)
        selected_view_idxs = np.sort(rs.choice(all_num_views, size=num_views, replace=False))
        view_idxs = np.concatenate([[0], np.cumsum(data_in["pc_sizes"])])
        for i in selected_view_idxs:
            idxs_mask[view_idxs[i]:view_idxs[i + 1]] = True
        all_idxs = all_idxs[idxs_mask]
    selected_idxs = rs.choice(all_idxs, size=num_pc_sample, replace=False)

    pc_sim_rgb = data_in["pc_sim_rgb"][selected_idxs].astype(np.float32) / 255
    pc_sim = data_in["pc_sim"][selected_idxs].astype(np.float32)
    pc_nocs = data_in["pc_nocs"][selected_idxs].astype(np.float32)
    grip_idx = data_in["grip_vertex_idx"]
    sim_grip_point = data_in["cloth_sim_verts"][grip_idx].reshape((1, 3))
    nocs_grip_point = data_in["cloth_nocs_verts"][grip_idx].reshape((1, 3))
    dists = np.linalg.norm(pc_sim - sim_grip_point[0], axis=1)
    data = {
        "x": pc_sim_rgb,
        "y": pc_nocs,
        "pos": pc_sim,
        "scale": np.array([data_in["scale"]]),
        "sim_grip_point": sim_grip_point,
        "nocs_grip_point": nocs_grip_point,
        "grip_pc_idx": np.array([np.argmin(dists)]),
        "dataset_idx": np.array([idx]),
    }
    if cloth_sim_aabb is not None:
        aabb = np.asarray(cloth_sim_aabb)
        data["cloth_sim_aabb"] = aabb.reshape((1,) + aabb.shape)
    return data


def noise_augmentation(idx, data, pc_noise_std, static_epoch_seed=False):
    """noise_augumentation: 408-423 (the sum is float64, as in the reference)"""
    rs = np.random.RandomState(seed=idx if static_epoch_seed else None)
    noise = rs.normal(loc=(0,) * 3, scale=(pc_noise_std,) * 3, size=data["pos"].shape)
    out = dict(data)
    out["pos"] = data["pos"] + noise
    return out


def rotation_augmentation(idx, data, random_rot_range=(-90, 90), static_epoch_seed=False, volume_task_space=False):
    """rotation_augumentation: 370-406"""
    assert len(random_rot_range) == 2 and random_rot_range[0] <= random_rot_range[-1]
    rs = np.random.RandomState(seed=idx if static_epoch_seed else None)
    rot_angle = rs.uniform(*random_rot_range)
    rot_mat = Rotation.from_euler("z", rot_angle, degrees=True).as_matrix().astype(np.float32)
    out = dict(data)
    for key in (("pos", "sim_grip_point") if volume_task_space else ("pos", "sim_grip_point", "gt_sim_points")):
        if key in data:
            out[key] = (data[key] @ rot_mat.T).astype(np.float32)
    if volume_task_space:
        offset_vec = np.array([0.5, 0.5, 0], dtype=np.float32)
        for key in ("volume_query_points", "surf_query_points"):
            if key in data:
                out[key] = ((data[key] - offset_vec) @ rot_mat.T + offset_vec).astype(np.float32)
    out["input_aug_rot_mat"] = rot_mat.reshape((1,) + rot_mat.shape)
    return out


class GarmentInputDataset:
    """dataset[idx] -> dict of numpy arrays (the fields of the reference's Data object that the inference path and predict.py
    read); collate() -> Batch.  Constructor arguments carry the reference's names and defaults."""

    def __init__(self, zarr_path, num_pc_sample=6000, enable_augumentation=True, random_rot_range=(-90, 90), num_views=4,
                 pc_noise_std=0, static_epoch_seed=False, volume_task_space=False, **kwargs):
        assert num_views > 0
        root = zarr_store.open_group(zarr_path, create=False)
        self.samples_group = root["samples"]
        self.keys = sorted(self.samples_group.keys())
        self.num_pc_sample, self.enable_augumentation, self.random_rot_range = num_pc_sample, enable_augumentation, tuple(random_rot_range)
        self.num_views, self.pc_noise_std, self.static_epoch_seed, self.volume_task_space = num_views, pc_noise_std, static_epoch_seed, volume_task_space
        self.cloth_sim_aabb = root["summary"]["cloth_aabb_union"][:].astype(np.float32)

    def __len__(self):
        return len(self.keys)

    def __getitem__(self, idx):
        data_in = data_io(self.samples_group[self.keys[idx]])
        data = get_base_data(idx, data_in, self.num_pc_sample, self.num_views, self.static_epoch_seed, self.cloth_sim_aabb)
        data["input_aug_rot_mat"] = np.expand_dims(np.eye(3, dtype=np.float32), axis=0)
        if self.pc_noise_std > 0:
            data = noise_augmentation(idx, data, self.pc_noise_std, self.static_epoch_seed)
        if self.enable_augumentation:
            data = rotation_augmentation(idx, data, self.random_rot_range, self.static_epoch_seed, self.volume_task_space)
        return data

    @staticmethod
    def collate(samples):
        """PyG-style batching: every field concatenated along dim 0, plus the `batch` vector of the per-point fields"""
        sizes = [len(s["pos"]) for s in samples]
        cat = {k: torch.from_numpy(np.concatenate([np.asarray(s[k]) for s in samples], axis=0)) for k in samples[0]}
        cat["pos"], cat["x"] = cat["pos"].float(), cat["x"].float()
        batch = torch.repeat_interleave(torch.arange(len(samples)), torch.tensor(sizes))
        return Batch(sizes=sizes, batch=batch, **cat)
