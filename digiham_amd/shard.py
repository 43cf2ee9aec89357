"""Channel sharding across the GPUs of one node.

Channels are independent (each is one `rrc_filter | gfsk_demodulator | dmr_decoder` pipe in the
reference), so the only partitioning is by channel range and the data path needs no collective.
`torch.distributed` is used for two things only, both outside the timed region: a barrier, and a
MAX/SUM reduction of the per-rank wall time and unit counts for reporting.
"""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def channel_range(total_channels, rank, world):
    """[lo, hi) owned by `rank` when `total_channels` are split as evenly as possible (strong scaling)."""
    base, extra = divmod(total_channels, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (nccl == RCCL on ROCm; gloo on CPU)."""
    import torch
    import torch.distributed as dist
    rank, world, local = rank_world()
    # DH_FORCE_DIST=1 initialises the process group even for a single rank (exercises the RCCL path on a 1-GPU box)
    if (world > 1 or os.environ.get("DH_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:             # only without a launcher (DH_FORCE_DIST on one rank): any free port
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            # one process per GPU: bind the device before RCCL creates its communicator
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def barrier():
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def reduce_report(seconds, units, device=None):
    """(max seconds over ranks, total units over ranks)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seconds, units
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
