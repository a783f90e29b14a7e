"""Multi-GPU driver of the sharded bake (include/omm_mi355x_ext.h: ommxSharded*).

One process per GPU.  All compute is inside libomm-lib.so; this module only moves the two exchange buffers between ranks with
torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests):

    begin  ->  all_reduce(SUM) of 4 uint32 words per active work item  ->  tail  ->  all_gather of the padded block
    contributions  ->  finish (every rank ends with the identical merged arrayData / descArray / indexBuffer in HBM)
"""
import ctypes as C


class DevicePointer:
    """Wraps a raw HBM pointer for torch via __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, nbytes, typestr="|u1", itemsize=1):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_tensor(torch, ptr, count, dtype):
    """torch view of `count` elements of `dtype` at device address `ptr`."""
    typestr = {torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    itemsize = {torch.int32: 4, torch.uint8: 1}[dtype]
    return torch.as_tensor(DevicePointer(ptr, count * itemsize, typestr, itemsize), device="cuda")


def allreduce_words_(dist, words):
    """SUM all-reduce of the per-item metadata words (int32 view; foreign entries are zero so the sum is a merge)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(words, op=dist.ReduceOp.SUM)
    return words


def allgather_padded(dist, torch, contribution, world):
    """all-gather of equally sized (padded) uint8 contributions -> tensor [world * stride]."""
    if dist is None or not dist.is_initialized() or world == 1:
        return contribution
    out = torch.empty(world * contribution.numel(), dtype=contribution.dtype, device=contribution.device)
    if dist.get_backend() == "gloo":   # (tests: gloo has no all_gather_into_tensor for device tensors)
        dist.all_gather(list(out.view(world, -1).unbind(0)), contribution)
    else:
        dist.all_gather_into_tensor(out, contribution)
    return out


def bind(dll):
    dll.ommxShardedBegin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    dll.ommxShardedGetMeta.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    dll.ommxShardedTail.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    dll.ommxShardedFinish.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    dll.ommxShardedDestroy.argtypes = [C.c_void_p]
    return dll


def sharded_bake(dll, baker, desc_ptr, rank, world, torch, dist):
    """Runs one sharded bake on this rank; returns the ommxDeviceBakeResult handle (destroy with ommxDestroyDeviceBakeResult)."""
    bind(dll)
    h = C.c_void_p()
    r = dll.ommxShardedBegin(baker, desc_ptr, rank, world, C.byref(h))
    if r != 0:
        raise RuntimeError("ommxShardedBegin failed: %d" % r)
    try:
        words, n = C.c_void_p(), C.c_uint64()
        assert dll.ommxShardedGetMeta(h, C.byref(words), C.byref(n)) == 0
        if n.value and world > 1:
            allreduce_words_(dist, device_tensor(torch, words.value, n.value, torch.int32))
            torch.cuda.current_stream().synchronize()
        contrib, nbytes, stride = C.c_void_p(), C.c_uint64(), C.c_uint64()
        r = dll.ommxShardedTail(h, C.byref(contrib), C.byref(nbytes), C.byref(stride))
        if r != 0:
            raise RuntimeError("ommxShardedTail failed: %d" % r)
        gathered_ptr = contrib.value
        keep = None
        if world > 1 and stride.value:   # stride is the global maximum: zero means no rank has any block to contribute
            keep = allgather_padded(dist, torch, device_tensor(torch, contrib.value, stride.value, torch.uint8), world)
            torch.cuda.current_stream().synchronize()
            gathered_ptr = keep.data_ptr()
        out = C.c_void_p()
        r = dll.ommxShardedFinish(h, gathered_ptr, C.byref(out))
        if r != 0:
            raise RuntimeError("ommxShardedFinish failed: %d" % r)
        del keep
        return out
    finally:
        dll.ommxShardedDestroy(h)
