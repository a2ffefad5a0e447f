#!/usr/bin/env python3
"""Golden vectors for the input side from the REFERENCE's own dataset class (build container only).

Run:  python tests/golden/make_golden_dataset.py        (needs /root/reference; never runs on the GPU box)

/root/reference/datasets/conv_implicit_wnf_dataset.py is imported with sys.modules stubs for its absent third-party packages
(zarr, igl, torch_geometric, pytorch_lightning ...: make_golden_ref.install_stubs + three more); its get_base_data /
noise_augumentation / rotation_augumentation methods then run unmodified on a synthetic sample (bound to a bare namespace that
carries the attributes __init__ would have set).  Only DATA (the synthetic sample, the parameters, the methods' outputs) is written to ref_dataset.npz.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ref as G  # noqa: E402


def synthetic_sample(seed, views=4):
    rs = np.random.RandomState(seed)
    sizes = rs.randint(900, 1400, size=views).astype(np.int64)
    n = int(sizes.sum())
    nv = 500
    return {
        "cloth_sim_verts": rs.normal(size=(nv, 3)).astype(np.float32) * 0.3,
        "cloth_nocs_verts": rs.uniform(size=(nv, 3)).astype(np.float32),
        "cloth_faces_tri": rs.randint(0, nv, size=(900, 3)).astype(np.int32),
        "pc_nocs": rs.uniform(size=(n, 3)).astype(np.float16),
        "pc_sim": (rs.normal(size=(n, 3)) * 0.3).astype(np.float16),
        "pc_sim_rgb": rs.randint(0, 256, size=(n, 3)).astype(np.uint8),
        "pc_sizes": sizes,
        "scale": float(rs.uniform(0.5, 2.0)),
        "grip_vertex_idx": int(rs.randint(0, nv)),
    }


def main():
    G.install_stubs()
    for name in ("zarr", "igl"):
        sys.modules[name] = types.ModuleType(name)
    import importlib.util                            # by path: the name `datasets` belongs to an installed package here
    spec = importlib.util.spec_from_file_location("ref_conv_implicit_wnf_dataset", os.path.join(G.REF, "datasets", "conv_implicit_wnf_dataset.py"))
    ref_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_mod)
    Ref = ref_mod.ConvImplicitWNFDataset

    out = {}
    cases = [dict(idx=3, num_pc_sample=600, num_views=4, pc_noise_std=0.0, rot=(-90, 90), task_space=False),
             dict(idx=7, num_pc_sample=500, num_views=2, pc_noise_std=0.01, rot=(-180, 180), task_space=False),
             dict(idx=11, num_pc_sample=256, num_views=3, pc_noise_std=0.0, rot=(10, 10), task_space=True)]
    for ci, c in enumerate(cases):
        sample = synthetic_sample(100 + ci)
        aabb = np.array([[-1, -1, -1], [1, 1, 1]], dtype=np.float32) * (1 + ci)
        self = types.SimpleNamespace(num_pc_sample=c["num_pc_sample"], static_epoch_seed=True, num_views=c["num_views"], cloth_sim_aabb=aabb,
                                     pc_noise_std=c["pc_noise_std"], random_rot_range=c["rot"], volume_task_space=c["task_space"])
        base = Ref.get_base_data(self, c["idx"], data_in=sample)
        data = dict(base)
        data["input_aug_rot_mat"] = np.expand_dims(np.eye(3, dtype=np.float32), axis=0)
        if c["task_space"]:
            data["surf_query_points"] = np.random.RandomState(5).uniform(size=(40, 3)).astype(np.float32)
        if c["pc_noise_std"] > 0:
            data = Ref.noise_augumentation(self, c["idx"], data=data)
        final = Ref.rotation_augumentation(self, c["idx"], data=data)
        for k, v in sample.items():
            out[f"c{ci}/in/{k}"] = np.asarray(v)
        out[f"c{ci}/aabb"] = aabb
        out[f"c{ci}/params"] = np.array([c["idx"], c["num_pc_sample"], c["num_views"], c["pc_noise_std"], c["rot"][0], c["rot"][1], float(c["task_space"])])
        if c["task_space"]:
            out[f"c{ci}/surf_in"] = data["surf_query_points"]
        for k, v in base.items():
            out[f"c{ci}/base/{k}"] = np.asarray(v)
        for k, v in final.items():
            out[f"c{ci}/final/{k}"] = np.asarray(v)
    # the data module's seeded instance split (prepare_data: 478-529), run unmodified; only the dataset class it instantiates is swapped
    # for a bare object carrying the `groups_df` / `static_epoch_seed` attributes prepare_data reads (the real one needs zarr)
    import pandas as pd
    split_cases = [dict(n_inst=37, views=(1, 5), split=(8, 1, 1), seed=0), dict(n_inst=10, views=(2, 3), split=(8, 1, 1), seed=0),
                   dict(n_inst=123, views=(1, 4), split=(6, 3, 1), seed=42)]
    for si, c in enumerate(split_cases):
        rs = np.random.RandomState(900 + si)
        ids = np.concatenate([np.full(rs.randint(*c["views"]), i) for i in rs.permutation(c["n_inst"])])
        rs.shuffle(ids)                                                    # instances interleaved in dataset order
        sample_ids = np.array([f"{v:05d}_Dress" for v in ids])
        keys = [f"{k:06d}" for k in range(len(sample_ids))]

        class FakeDataset:
            def __init__(self, **kw):
                self.static_epoch_seed = kw.get("static_epoch_seed", False)
                self.groups_df = pd.DataFrame({"sample_id": sample_ids, "group_key": keys, "idx": np.arange(len(keys))}, index=keys)

        ref_mod.ConvImplicitWNFDataset = FakeDataset
        dm = ref_mod.ConvImplicitWNFDataModule(dataset_split=c["split"], split_seed=c["seed"], batch_size=1, num_workers=0)
        dm.prepare_data()
        out[f"s{si}/sample_ids"] = sample_ids
        out[f"s{si}/params"] = np.array(list(c["split"]) + [c["seed"]])
        out[f"s{si}/train"], out[f"s{si}/val"], out[f"s{si}/test"] = (np.asarray(v, dtype=np.int64) for v in (dm.train_idxs, dm.val_idxs, dm.test_idxs))
    np.savez_compressed(os.path.join(HERE, "ref_dataset.npz"), **out)
    print("wrote ref_dataset.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
