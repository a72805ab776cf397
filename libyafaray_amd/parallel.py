"""Multi-GPU composition of one frame: pixel tiles are dealt round-robin to the ranks
(tile t -> rank t % world, yafaray_setShard), every rank renders its tiles into zero-initialised
full-frame film planes, combines them into one [H][W][5] film (yafgpu_film_combine), and ONE sum-reduce
of that film (20 MB at 1024^2) to rank 0 assembles the frame — the semantics of the
reference's own film merge (sum colour, sum weight, normalise afterwards: imagefilm.cc:1467-1557).
A reduce(sum) rather than a gather because a sample near a pixel's right/lower edge also lands on
the neighbouring pixel (imagefilm.cc:933-936), which may belong to another rank's tile."""
import ctypes as C

import torch
import torch.distributed as dist

from . import interface as _iface


class FilmComm:
    """The C ABI's RCCL communicator (include/yafaray_c_api.h "multi-GPU", csrc/yafaray_reduce.cpp): what a C/C++ host uses
    to assemble a sharded frame — ncclReduce(sum) of the [H][W][5] film, no torch on the data path.  This wrapper only moves
    the 128-byte unique id from rank 0 to the others (here through the process group's store; a C host uses a file, MPI ...)."""

    ID_BYTES = 128

    def __init__(self, rank, world, device_index, unique_id=None, share=None):
        L = _iface.load()
        self._L = L
        if unique_id is None:
            buf = C.create_string_buffer(self.ID_BYTES)
            if rank == 0 and not L.yafaray_commGetUniqueId(buf):
                raise _iface.YafaRayError("commGetUniqueId: " + L.yafaray_commLastError().decode())
            unique_id = share(buf.raw if rank == 0 else None) if share is not None else buf.raw
        self.handle = L.yafaray_commCreate(C.c_char_p(unique_id), rank, world, device_index)
        if not self.handle:
            raise _iface.YafaRayError("commCreate: " + L.yafaray_commLastError().decode())
        self.rank, self.world = rank, world

    @classmethod
    def from_process_group(cls, device_index):
        """one communicator over the ranks of torch.distributed's default group; the id travels through its store"""
        rank, world = dist.get_rank(), dist.get_world_size()

        def share(raw):
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        return cls(rank, world, device_index, share=share)

    @property
    def backend(self):
        return self._L.yafaray_commBackend().decode()

    def reduce_film(self, film: torch.Tensor, dst=0, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        if not self._L.yafaray_reduceFilm(self.handle, film.data_ptr(), film.numel(), dst, st):
            raise _iface.YafaRayError("reduceFilm: " + self._L.yafaray_commLastError().decode())
        return film

    def all_reduce(self, t: torch.Tensor, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        if not self._L.yafaray_allReduce(self.handle, t.data_ptr(), t.numel(), st):
            raise _iface.YafaRayError("allReduce: " + self._L.yafaray_commLastError().decode())
        return t

    def close(self):
        if self.handle:
            self._L.yafaray_commDestroy(self.handle)
            self.handle = None


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
