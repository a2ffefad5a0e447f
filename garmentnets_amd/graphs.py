"""HIP-graph replay of the dense, shape-static part of the predict step.

pointnet2_forward -> unet3d_forward -> volume_lattice_forward launch ~150 kernels whose shapes depend only on (batch size, points
per garment, grid, lattice size); at small batch (the reference's predict.py runs one garment at a time) the step is launch-bound.
GraphedDenseStages captures that kernel sequence once into a HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm; the C-ABI kernels
launch on torch's current stream, so they are captured like any other work) and replays it for every new batch of the same shape:
one graph launch instead of ~150 kernel launches.  The data-dependent tail (iso-surface extraction: vertex counts, per-garment
allocations) stays eager.
"""
import torch

from .batch import Batch


class GraphedDenseStages:
    def __init__(self, model, example_batch, volume_size, warmup=2):
        dev = example_batch.pos.device
        self.model, self.volume_size = model, int(volume_size)
        self.sizes = list(example_batch.sizes)
        self.static = Batch(sizes=self.sizes, x=example_batch.x.clone(), pos=example_batch.pos.clone(), batch=example_batch.batch.clone())
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                  # builds every weight pack / cache outside the capture
                self._run()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.outputs = self._run()

    def _run(self):
        p2 = self.model.pointnet2_forward(self.static)
        u3 = self.model.unet3d_forward(p2)
        wnf = self.model.volume_lattice_forward(u3, self.volume_size)["pred_volume"]
        return p2, u3, wnf

    def matches(self, batch):
        return list(batch.sizes) == self.sizes

    def __call__(self, batch):
        """-> (pointnet2_result, unet3d_result, wnf_all) for a batch with the captured shape; the results live in the graph's static
        buffers and are overwritten by the next call"""
        if not self.matches(batch):
            raise ValueError(f"graph captured for garment sizes {self.sizes}, got {list(batch.sizes)}")
        self.static.x.copy_(batch.x, non_blocking=True)
        self.static.pos.copy_(batch.pos, non_blocking=True)
        self.static.batch.copy_(batch.batch, non_blocking=True)
        self.graph.replay()
        return self.outputs
