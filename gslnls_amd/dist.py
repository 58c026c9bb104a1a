"""One process per GPU: shard multi-start's sample points over the ranks of a torch.distributed job.

The only exchange step of the path is one all-gather of the per-point records per batch
(SURVEY.md 8(e)); with backend "nccl" that is an RCCL all-gather over xGMI on device buffers,
with "gloo" (CPU tests) the same call on host tensors.  The C library writes this rank's shard
into `shard`, calls back into `_allgather`, then reads the completed `all` buffer.
"""
import ctypes as C

from . import _lib

ALLGATHER_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)
_state = {}


def init_multistart_comm(max_points, p, device=None):
    """Register the communicator of the default process group with libgslnls_hip.so.

    max_points: largest batch (mstart_n) that will be sharded; p: number of parameters.
    """
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        _lib.lib().gslnls_set_comm(0, 1, None, None, None, None, 0, 0)
        return None
    rank, world = dist.get_rank(), dist.get_world_size()
    K = _lib.lib().gslnls_mstart_record_size(p)
    per = (max_points + world - 1) // world
    on_device = dist.get_backend() == "nccl"
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if on_device else "cpu")
    shard = torch.zeros(per * K, dtype=torch.float64, device=dev)
    allb = torch.zeros(world * per * K, dtype=torch.float64, device=dev)
    calls = {"n": 0}

    def _allgather(_ctx, per_points, k):
        try:
            dist.all_gather_into_tensor(allb[:world * per_points * k], shard[:per_points * k])
            if on_device:
                torch.cuda.current_stream().synchronize()
            calls["n"] += 1
            return 0
        except Exception:  # noqa
            return -1
    cb = ALLGATHER_T(_allgather)
    rc = _lib.lib().gslnls_set_comm(rank, world, C.cast(cb, C.c_void_p), None, C.c_void_p(shard.data_ptr()),
                                    C.c_void_p(allb.data_ptr()), world * per, int(on_device))
    if rc != 0:
        raise RuntimeError("gslnls_set_comm failed: %d" % rc)
    _state.update(cb=cb, shard=shard, allb=allb, calls=calls)  # keep alive
    return calls


def reset_comm():
    _lib.lib().gslnls_set_comm(0, 1, None, None, None, None, 0, 0)
    _state.clear()
