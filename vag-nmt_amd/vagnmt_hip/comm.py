"""RCCL communicator behind the C ABI (include/vag_nmt.h: vag_comm_*; SURVEY 8b "C1", 8e).

One process per GPU.  Rank 0 draws the 128-byte id, the other ranks receive it over whatever channel the launcher has
(here: torch.distributed's object broadcast on the default group, any backend -- gloo is enough; only the id travels that
way), then every rank binds a communicator to its current device.  all_reduce() enqueues an in-place fp32 sum on a side
stream that waits for the caller's stream, so the exchange of a finished gradient bucket runs beside the rest of backward;
the returned handle's wait() makes the caller's stream wait for it (no host synchronisation anywhere).  The tensor handed to
all_reduce() must stay alive until its handle has been waited on."""
import ctypes as C

import torch

from . import _lib as L


class _Pending:
    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        return True


class Comm:
    def __init__(self, rank=0, world_size=1, exchange=None):
        """exchange(bytes_or_None) -> bytes: returns rank 0's id on every rank (default: torch.distributed object broadcast;
        not needed when world_size == 1)."""
        self.rank, self.world = int(rank), int(world_size)
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            L.call("vag_comm_unique_id", ident)
        if self.world > 1:
            if exchange is None:
                import torch.distributed as dist

                def exchange(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            raw = exchange(bytes(ident.raw) if self.rank == 0 else None)
            ident = C.create_string_buffer(raw, 128)
        self._h = C.c_void_p()
        L.call("vag_comm_init", C.byref(self._h), self.world, self.rank, ident)
        self._side = torch.cuda.Stream()

    def all_reduce(self, t):
        """In-place sum of a contiguous fp32 HIP tensor across the ranks, beside the caller's stream."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        ready = torch.cuda.Event()
        ready.record()                                   # the bucket is final at this point of the caller's stream
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            L.call("vag_comm_allreduce", self._h, C.c_void_p(t.data_ptr()), t.numel(),
                   C.c_void_p(self._side.cuda_stream))
            done = torch.cuda.Event()
            done.record()
        # (no t.record_stream(side): the caller keeps `t` alive until wait() -- a gradient bucket lives as long as its driver --
        # and a storage marked that way makes the allocator record an event on the side stream when it is freed, which
        # aborts the process if that happens inside somebody's graph capture)
        return _Pending(done)

    def close(self):
        if self._h:
            torch.cuda.synchronize()
            L.call("vag_comm_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
