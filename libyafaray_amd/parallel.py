"""Multi-GPU composition of one frame: pixel tiles are dealt round-robin to the ranks
(tile t -> rank t % world, yafaray_setShard), every rank renders its tiles into zero-initialised
full-frame film planes, combines them into one [H][W][5] film (yafgpu_film_combine), and ONE sum-reduce
of that film (20 MB at 1024^2) to rank 0 assembles the frame — the semantics of the
reference's own film merge (sum colour, sum weight, normalise afterwards: imagefilm.cc:1467-1557).
A reduce(sum) rather than a gather because a sample near a pixel's right/lower edge also lands on
the neighbouring pixel (imagefilm.cc:933-936), which may belong to another rank's tile."""
import torch
import torch.distributed as dist


def shard_of_tile(tile_index, world_size):
    return tile_index % world_size


def reduce_film(film: torch.Tensor, dst: int = 0):
    """Sum the per-rank films (or film planes) onto rank `dst` (RCCL over xGMI on GPUs, gloo on CPU tests).
    Interior pixels are non-zero on exactly one rank, so their sum is exact (x + 0); a tile-border pixel also
    carries the neighbouring rank's splat, added here instead of inside the combine (<= 1 ulp of order effect)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


reduce_planes = reduce_film      # the same collective on the four uncombined splat planes


class _DeviceFloats:
    """a float32 device array by address, for torch.as_tensor (the CUDA array interface; zero copy)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def plane_exchange(device=None):
    """The function yafaray_setPlaneExchange wants (Interface.setPlaneExchange): all-reduce(sum) of a float32 device array
    over the process group, in place — RCCL over xGMI with the nccl backend, staged through the host with gloo.  Used
    between the passes of a multi-pass (adaptive anti-aliasing) render of a sharded frame: every rank gets every rank's
    splat planes, so every rank takes the single-GPU render's decision about which pixels to sample again."""
    def exchange(ptr, n):
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return
        t = torch.as_tensor(_DeviceFloats(ptr, n), device=device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        if dist.get_backend() == "gloo":          # CPU rehearsals / one-GPU boxes: through the host
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    return exchange
