"""Chamfer metrics on the GPU -- the definitions of /root/reference/eval.py:259-271 (chamfer) and
:381-402 (hybrid chamfer: nearest neighbours in NOCS space, distances in simulation space), with the two cKDTree queries
replaced by the exact brute-force gn_nearest_neighbor kernel."""
import torch

from .. import ops


def chamfer(pred_points, gt_points):
    """-> dict(chamfer_forward, chamfer_backward, chamfer_symmetrical) (means of Euclidean NN distances)"""
    _, d2f = ops.nearest_neighbor(pred_points, gt_points)
    _, d2b = ops.nearest_neighbor(gt_points, pred_points)
    fwd = torch.sqrt(d2f.double()).mean()
    bwd = torch.sqrt(d2b.double()).mean()
    return {"chamfer_forward": fwd, "chamfer_backward": bwd, "chamfer_symmetrical": 0.5 * (fwd + bwd)}


def hybrid_chamfer(pred_nocs_points, gt_nocs_points, pred_sim_points, gt_sim_points):
    fi, _ = ops.nearest_neighbor(pred_nocs_points, gt_nocs_points)
    bi, _ = ops.nearest_neighbor(gt_nocs_points, pred_nocs_points)
    fwd = torch.norm(pred_sim_points.double() - gt_sim_points.double()[fi.long()], dim=1).mean()
    bwd = torch.norm(gt_sim_points.double() - pred_sim_points.double()[bi.long()], dim=1).mean()
    return {"hybrid_chamfer_forward": fwd, "hybrid_chamfer_backward": bwd, "hybrid_chamfer_symmetrical": 0.5 * (fwd + bwd)}
