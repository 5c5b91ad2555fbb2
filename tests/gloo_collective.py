"""CPU stand-in of ``renormalizer_amd.parallel.RcclCollective`` for the world_size-2 tests: the same three
operations (barrier, max, all-gather) on torch.distributed's gloo backend.  Test infrastructure only - the product
package talks to librccl.so through ctypes and never imports torch."""
import numpy as np


class GlooCollective:
    kind = "gloo"

    def __init__(self):
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo")
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self.dist.barrier()

    def allreduce_max(self, value: float) -> float:
        import torch
        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allgather(self, row: np.ndarray) -> np.ndarray:
        import torch
        t = torch.as_tensor(np.ascontiguousarray(row, dtype=np.float64).ravel())
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return np.stack([p.numpy() for p in parts])

    def close(self):
        self.dist.destroy_process_group()
