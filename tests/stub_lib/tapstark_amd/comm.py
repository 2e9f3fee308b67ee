"""STUB of tapstark_amd.comm (the library's native RCCL communicator) for the CPU dry-run: the unique
id and the rank bookkeeping are real, the collective is a torch.distributed (gloo) barrier."""
import os


def rccl_unique_id() -> bytes:
    return os.urandom(128)


class RcclComm:
    backend = "stub rccl"

    def __init__(self, ctx, unique_id, rank, world):
        assert isinstance(unique_id, (bytes, bytearray)) and len(unique_id) == 128, "every rank needs its group's id"
        self.rank, self.world, self.uid = rank, world, bytes(unique_id)

        class _C:
            abort = None
            user = None
        self.c = _C()

    def info(self):
        return {"comm_count": self.world, "user_rank": self.rank, "world": self.world, "checked": 1}

    def exchange(self):
        import torch.distributed as dist
        ids = [None] * dist.get_world_size()
        dist.all_gather_object(ids, (self.uid, self.rank))
        mine = [r for u, r in ids if u == self.uid]
        assert sorted(mine) == list(range(self.world)), f"group of {self.world}: ranks {mine}"

    def close(self):
        pass
