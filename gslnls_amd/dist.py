"""One process per GPU: shard multi-start's sample points (and batched robust fits) over the ranks of a job.

The only exchange step of the path is one all-gather of the per-point records per batch.

* init_library_comm(): the production form.  libgslnls_hip.so owns an RCCL communicator and issues the
  ncclAllGather itself, on its own stream, right behind the batch kernel; Python only carries the 128-byte
  bootstrap id from rank 0 to the others (any channel would do: an R host uses gslnls_comm_init_file).
* init_multistart_comm(): the callback form -- the library calls back into torch.distributed.  Used with the
  "gloo" backend by the CPU tests (world_size 2 on one box), where no RCCL communicator can exist.
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


def init_library_comm():
    """Bind the in-library RCCL communicator to the ranks of the default process group (backend "nccl").
    Returns the communicator's description; raises when RCCL cannot be bound."""
    import torch
    import torch.distributed as dist
    L = _lib.lib()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        L.gslnls_comm_destroy()
        return "one rank"
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    buf = C.create_string_buffer(128)
    # every rank makes an id (only rank 0's is used): RCCL is thereby bound on each of them before any rank enters
    # ncclCommInitRank, which is itself a collective
    ok = torch.tensor([1 if L.gslnls_comm_get_unique_id(buf) == 0 else 0], dtype=torch.int32, device=dev)
    t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone().to(dev)
    dist.broadcast(t, 0)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) != 1:
        raise RuntimeError("RCCL unavailable: %s" % L.gslnls_comm_last_error().decode())
    rc = L.gslnls_comm_init_rank(bytes(t.cpu().numpy().tobytes()), rank, world)
    flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        L.gslnls_comm_destroy()
        raise RuntimeError("ncclCommInitRank failed: %s" % L.gslnls_comm_last_error().decode())
    return "RCCL communicator inside libgslnls_hip.so, %d ranks" % world


def reset_comm():
    _lib.lib().gslnls_comm_destroy()
    _lib.lib().gslnls_set_comm(0, 1, None, None, None, None, 0, 0)
    _state.clear()
