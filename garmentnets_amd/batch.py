"""Minimal stand-in for torch_geometric.data.Batch: attribute bag with ``num_graphs`` and ``.to()``.

The reference builds ``Batch(x=, pos=, batch=, sim_points=, pred_confidence=)`` (networks/conv_implicit_wnf.py:232-237)
and reads ``num_graphs`` (:51), which PyG derives as ``int(batch.max()) + 1`` (a device sync).  Here the per-example
sizes can be attached once on the host (``sizes``) so that the hot path never synchronises.
"""
import torch


class Batch:
    def __init__(self, sizes=None, **kwargs):
        self._sizes = None if sizes is None else [int(s) for s in sizes]
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    @property
    def sizes(self):
        """points per example (host list); computed with ONE device sync if it was not provided."""
        if self._sizes is None:
            b = self.batch
            n = int(b.max().item()) + 1 if b.numel() else 0
            self._sizes = torch.bincount(b, minlength=n).cpu().tolist()
        return self._sizes

    @property
    def num_graphs(self):
        return len(self.sizes)

    def to(self, device, **kw):
        out = Batch(sizes=self._sizes)
        for k in self.keys:
            v = getattr(self, k)
            setattr(out, k, v.to(device, **kw) if torch.is_tensor(v) else v)
        dev = torch.device(device)
        if dev.type == "cuda" and torch.cuda.is_available():
            # when the copies were queued: a consumer on ANOTHER stream (predict.PredictJob's front stream) waits for this event instead of
            # for everything the producing stream has queued since.  The event vouches for the tensors AS THEY ARE NOW: ready_event() hands it
            # out only while every field still is the same tensor object at the same version (a field re-bound or edited in place after
            # .to() was produced by later work on the caller's stream -- the consumer then has to order itself behind that stream)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            out._ready = (ev, out._fingerprint())
        return out

    def _fingerprint(self):
        return tuple((k, id(v), v._version) for k, v in ((k, getattr(self, k)) for k in self.keys) if torch.is_tensor(v))

    def ready_event(self):
        """the event recorded behind the copies of .to(), or None when a field has been re-bound / modified in place since (or the batch was not
        made by .to()): the caller must then wait for the producing stream itself"""
        r = self.__dict__.get("_ready")
        if r is None or r[1] != self._fingerprint():
            return None
        return r[0]

    def __repr__(self):
        return "Batch(" + ", ".join(f"{k}={tuple(getattr(self, k).shape) if torch.is_tensor(getattr(self, k)) else getattr(self, k)}" for k in self.keys) + ")"
