"""PointNet2NOCS (inference) -- API twin of /root/reference/networks/pointnet2_nocs.py:58-195.

Same constructor kwargs, sub-module names (``sa1_module`` ... ``global_lin2``: the checkpoint schema), ``forward(data)``
result keys and ``logits_to_nocs`` / ``get_virtual_grid`` helpers; training code (losses, Lightning steps,
visualisation, :197-448) is out of scope.  All arithmetic runs in HIP kernels (garmentnets_amd.ops).
"""
import torch
from torch import nn

from .. import ops
from ..components.gridding import VirtualGrid
from ..components.mlp import MLP, HipLinear
from ..components.pointnet2 import FPModule, GlobalSAModule, SAModule, Segments


class PointNet2NOCS(nn.Module):
    def __init__(self, feature_dim, batch_norm, dropout, sa1_ratio, sa1_r, sa2_ratio, sa2_r, fp3_k, fp2_k, fp1_k,
                 symmetry_axis=None, nocs_bins=None, learning_rate=1e-4, nocs_loss_weight=1, grip_point_loss_weight=1,
                 vis_per_items=0, max_vis_per_epoch_train=0, max_vis_per_epoch_val=0, batch_size=None):
        super().__init__()
        self.hparams = dict(feature_dim=feature_dim, batch_norm=batch_norm, dropout=dropout, sa1_ratio=sa1_ratio, sa1_r=sa1_r,
                            sa2_ratio=sa2_ratio, sa2_r=sa2_r, fp3_k=fp3_k, fp2_k=fp2_k, fp1_k=fp1_k, symmetry_axis=symmetry_axis,
                            nocs_bins=nocs_bins)
        self.sa1_module = SAModule(sa1_ratio, sa1_r, MLP([3 + 3, 64, 64, 128], batch_norm=batch_norm))
        self.sa2_module = SAModule(sa2_ratio, sa2_r, MLP([128 + 3, 128, 128, 256], batch_norm=batch_norm))
        self.sa3_module = GlobalSAModule(nn=MLP([256 + 3, 256, 512, 1024], batch_norm=batch_norm))
        self.fp3_module = FPModule(k=fp3_k, nn=MLP([1024 + 256, 256, 256], batch_norm=batch_norm))
        self.fp2_module = FPModule(k=fp2_k, nn=MLP([256 + 128, 256, 128], batch_norm=batch_norm))
        self.fp1_module = FPModule(k=fp1_k, nn=MLP([128 + 3, 128, 128, 128], batch_norm=batch_norm))
        output_dim = 3 if nocs_bins is None else nocs_bins * 3
        self.lin1 = HipLinear(128, 128)
        self.lin2 = HipLinear(128, feature_dim)
        self.lin3 = HipLinear(feature_dim, output_dim)
        self.global_lin1 = HipLinear(1024, 1024)
        self.global_lin2 = HipLinear(1024, output_dim)
        self.nocs_bins = nocs_bins
        self.symmetry_axis = symmetry_axis
        self.batch_size = batch_size

    @property
    def device(self):
        return self.lin1.weight.device

    def set_self_loop_scope(self, scope):
        """"batch" (default; PyG's literal PointConv rule on a batched graph) or "example" (every garment of a batch gets its batch-of-one
        result: what the reference's predict.py, which asserts batch_size == 1, produces) -- components/pointnet2.py PointConv"""
        if scope not in ("batch", "example"):
            raise ValueError(f"self_loop_scope={scope!r}")
        self.sa1_module.conv.self_loop_scope = self.sa2_module.conv.self_loop_scope = scope
        return self

    def forward(self, data, seg=None):
        """data: .x (N,3) rgb, .pos (N,3), .batch (N,) sorted int64 [, .sizes host list] -- eval mode (dropout = identity).
        seg: the batch's Segments when the caller already has them (ConvImplicitWNFPipeline.pointnet2_forward: sizes travel with the call,
        nothing about a batch is parked on the shared module)"""
        if seg is None:
            sizes = data._sizes if hasattr(data, "_sizes") else getattr(data, "sizes", None)
            seg = Segments.of(data.batch, sizes)
        x = data.x.float().contiguous()
        pos = data.pos.float().contiguous()
        sa0 = (x, pos, seg)
        sa1 = self.sa1_module(*sa0)
        sa2 = self.sa2_module(*sa1)
        sa3 = self.sa3_module(*sa2)
        fp3 = self.fp3_module(*sa3, *sa2)
        fp2 = self.fp2_module(*fp3, *sa1)
        h, _, _ = self.fp1_module(*fp2, *sa0)
        h = self.lin1(h, relu=True)
        features = self.lin2(h)
        logits = self.lin3(features)
        global_feature = sa3[0]
        g = self.global_lin1(torch.relu(global_feature))
        global_logits = self.global_lin2(g)
        return {
            "per_point_features": features,
            "per_point_logits": logits,
            "per_point_batch_idx": data.batch,
            "global_logits": global_logits,
            "global_feature": global_feature,
        }

    def logits_to_nocs(self, logits):
        if self.nocs_bins is None:
            return logits
        lg = logits.reshape(-1, self.nocs_bins * 3)
        _, _, nocs = ops.nocs_head(lg, self.nocs_bins)
        return nocs.reshape(logits.shape[:-1] + (3,)) if logits.dim() == 2 else nocs.reshape(3)

    def get_virtual_grid(self):
        return VirtualGrid(lower_corner=(0, 0, 0), upper_corner=(1, 1, 1), grid_shape=(self.nocs_bins,) * 3, batch_size=1,
                           device=self.device, int_dtype=torch.int64, float_dtype=torch.float32)
