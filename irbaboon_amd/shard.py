"""Instance sharding across GPUs: every (instance, channel) stream is an
independent convolution (fp/convolution.cpp has no statics; the plug-in's state is per processor,
Source/PluginProcessor.h:194-217), so GPU g simply owns a contiguous range of instances.  There is
no data-path collective; torch.distributed is used only to line ranks up and to take the max of
their timings."""
import os


def instance_range(n_total, rank, world):
    """Contiguous, balanced split of instances [0, n_total) over `world` ranks."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(instance, n_total, world):
    for r in range(world):
        lo, hi = instance_range(n_total, r, world)
        if lo <= instance < hi:
            return r
    raise ValueError("instance out of range")


class Ranks:
    """Rank bookkeeping from the torchrun environment; barrier and max-over-ranks."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self._dist = None
        self._device = "cpu"
        self.backend = None
        if self.world > 1 or os.environ.get("IRB_FORCE_DIST") == "1":      # the env hook rehearses the RCCL path with one rank
            import torch
            import torch.distributed as dist
            ndev = torch.cuda.device_count()
            if backend is None:
                # one rank per GPU over RCCL; ranks sharing a GPU (rehearsals on a 1-GPU box) use gloo
                backend = "nccl" if ndev >= int(os.environ.get("LOCAL_WORLD_SIZE", self.world)) and ndev > 0 else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self._device = f"cuda:{self.local_rank}"
            self.backend = backend
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            if self.backend == "nccl":
                self._dist.barrier(device_ids=[self.local_rank])
            else:
                self._dist.barrier()

    def max(self, value):
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self._dist is not None:
            self._dist.destroy_process_group()
            self._dist = None
