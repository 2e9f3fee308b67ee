"""STUB of tapstark_amd.dist.TorchComm for the CPU dry-run."""
import torch.distributed as dist


class TorchComm:
    def __init__(self, device=None, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    def exchange(self):
        dist.barrier(self.group)

    def close(self):
        pass
