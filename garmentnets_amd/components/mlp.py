"""Point-wise MLP on the HIP dense-layer kernel.

API / checkpoint-schema twin of /root/reference/components/mlp.py:3-20: ``MLP(channels, batch_norm)`` is a stack of
``[Linear, ReLU, PointBatchNorm1D]`` blocks (keys ``{i}.0.weight``, ``{i}.2.running_mean`` ...), BatchNorm applied
AFTER the ReLU and also on the last layer.  The torch layers only HOLD the parameters; ``forward`` folds the eval-mode
BatchNorm into a per-channel (scale, shift) and runs ``gn_linear`` (fp32 MFMA, fused bias+ReLU+affine epilogue).
"""
import threading

import torch
from torch import nn

from .. import ops


class ParamCache:
    """kernel-side packs derived from a module's parameters, shared by the host threads that run the module: `generation` names the
    parameter state (device, tensor versions) -- a new one drops every older entry -- and `key` the variant (arithmetic, split point ...).
    One lock per cache; entries are immutable once built; a reader that already holds an entry keeps it alive whatever happens to the table."""

    def __init__(self):
        self._lock, self._generation, self._items = threading.RLock(), None, {}

    def get(self, generation, key, build):
        with self._lock:
            if generation != self._generation:
                self._generation, self._items = generation, {}
            v = self._items.get(key)
            if v is None:
                with torch.no_grad():
                    v = self._items[key] = build()
            return v

    def __deepcopy__(self, memo):          # a copied / unpickled module starts with an empty cache
        return ParamCache()

    def __reduce__(self):
        return (ParamCache, ())


def param_cache(module, name):
    """the module's ParamCache `name` (created on first use; not a parameter, not in the state dict)"""
    c = module.__dict__.get(name)
    if c is None:
        c = module.__dict__.setdefault(name, ParamCache())
    return c


class PackedModule(nn.Module):
    """Caches kernel-friendly parameter packs; dropped whenever parameters move or are reloaded."""

    def _invalidate(self):
        object.__setattr__(self, "_packed", None)

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self._invalidate()
        return r

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._invalidate()

    def packed(self):
        p = getattr(self, "_packed", None)
        if p is None:
            with torch.no_grad():
                p = self._pack()
            object.__setattr__(self, "_packed", p)
        return p


def pack_wb(weight, bias):
    """weight [N][K] -> fp32 [N][pad4(K)] zero padded (16-byte rows for the aligned loader); bias fp32 or None."""
    w = weight.detach().float()
    n, k = w.shape
    wp = torch.zeros((n, ops.pad4(k)), dtype=torch.float32, device=w.device)
    wp[:, :k] = w
    b = None if bias is None else bias.detach().float().contiguous()
    return wp, b, k


def pack_linear(lin):
    return pack_wb(lin.weight, lin.bias)


def fold_batchnorm(bn):
    """eval-mode BatchNorm1d -> y = x*scale + shift"""
    scale = (bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps))
    shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    return scale.float().contiguous(), shift.float().contiguous()


class PointBatchNorm1D(nn.BatchNorm1d):
    """Parameter holder (components/mlp.py:3-7); evaluated inside gn_linear's epilogue."""


class MLPStack(PackedModule, nn.Sequential):
    def _pack(self):
        layers = []
        for block in self:
            wp, b, k = pack_linear(block[0])
            sc = sh = None
            if len(block) > 2:
                sc, sh = fold_batchnorm(block[2])
            layers.append((wp, b, sc, sh, k))
        return layers

    def forward(self, x):
        lead = x.shape[:-1]
        h = x.reshape(-1, x.shape[-1]) if x.dim() != 2 else x
        if h.stride(-1) != 1:
            h = h.contiguous()
        for wp, b, sc, sh, k in self.packed():
            h = ops.linear(h, wp, b, sc, sh, relu=True, K=k)
        if x.dim() != 2:
            h = h.reshape(*lead, h.shape[-1])
        return h


def MLP(channels, batch_norm=True):
    blocks = []
    for i in range(1, len(channels)):
        mods = [nn.Linear(channels[i - 1], channels[i]), nn.ReLU()]
        if batch_norm:
            mods.append(PointBatchNorm1D(channels[i]))
        blocks.append(nn.Sequential(*mods))
    return MLPStack(*blocks)


class HipLinear(PackedModule, nn.Linear):
    """nn.Linear evaluated by gn_linear (optionally with a fused ReLU)."""

    def _pack(self):
        return pack_linear(self)

    def forward(self, x, relu=False):
        wp, b, k = self.packed()
        return ops.linear(x, wp, b, None, None, relu=relu, K=k)
