"""Host-array collectives over an initialised ``torch.distributed`` group (gloo, CPU) -- test infrastructure: the product package
never imports torch (north_star: no PyTorch at run time); its own collectives are RCCL's, driven from libd2d.so."""

from typing import Optional

import numpy as np


class GlooHostComm:
    """Host-array all-gather / all-reduce over an initialised ``torch.distributed`` group (CPU tests)."""

    def __init__(self):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather(self, local: np.ndarray) -> np.ndarray:
        t = self.torch.from_numpy(np.ascontiguousarray(local))
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return np.stack([o.numpy() for o in outs])

    def gather(self, local: np.ndarray, root: int = 0) -> Optional[np.ndarray]:
        """``[world, ...]`` on ``root``, ``None`` elsewhere."""
        t = self.torch.from_numpy(np.ascontiguousarray(local))
        outs = [self.torch.empty_like(t) for _ in range(self.world)] if self.rank == root else None
        self.dist.gather(t, outs, dst=root)
        return np.stack([o.numpy() for o in outs]) if self.rank == root else None

    def allreduce_sum(self, local: np.ndarray) -> np.ndarray:
        t = self.torch.from_numpy(np.array(local, dtype=np.float64))
        self.dist.all_reduce(t)
        return t.numpy()

    def barrier(self):
        self.dist.barrier()
