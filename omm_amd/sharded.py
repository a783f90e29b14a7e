"""Python callers of the sharded bake (include/omm_mi355x_ext.h).  One process per GPU; all compute is inside libomm-lib.so.

  * native path (what bench.py --gpus N runs): ommxShardedBakeRccl -- the library itself issues the two RCCL collectives from C++ on its own
    streams.  This module only helps with the communicator bootstrap: rank 0 asks the library for an RCCL unique id, the 128 bytes travel
    through torch.distributed, every rank joins with ommxRcclCommInitRank (rccl_comm()).
  * the same one call over another transport: CollectivesComm -- the library runs the identical sequence and calls back into
    torch.distributed (any backend) for its all-reduce and all-gather (ommxCommFromCollectives).
  * caller-driven path (tests, transports other than RCCL): the four ommxSharded* phases with the two exchanges done here through
    torch.distributed ("gloo" in the CPU tests):

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


def rccl_comm(dll, torch, dist, rank, world):
    """RCCL communicator owned by the library: the unique id is created by rank 0's library and broadcast through torch.distributed
    (any initialised backend), then every rank joins.  Destroy with dll.ommxRcclCommDestroy."""
    dll.ommxRcclGetUniqueId.argtypes = [C.c_void_p, C.c_size_t]
    dll.ommxRcclCommInitRank.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    dll.ommxRcclCommDestroy.argtypes = [C.c_void_p]
    buf = (C.c_uint8 * 128)()
    ok = 1
    if rank == 0:
        ok = 1 if dll.ommxRcclGetUniqueId(buf, 128) == 0 else 0
    dev = "cuda" if (world > 1 and dist.get_backend() == "nccl") else "cpu"
    if world > 1:
        # the id travels with rank 0's status byte, so that a failure there is an exception on EVERY rank instead of a hang in the broadcast
        t = torch.tensor(list(buf) + [ok], dtype=torch.uint8, device=dev)
        dist.broadcast(t, 0)
        vals = t.cpu().tolist()
        ok = vals[128]
        for i in range(128):
            buf[i] = vals[i]
    if not ok:
        raise RuntimeError("ommxRcclGetUniqueId failed on rank 0 (is librccl.so.1 loadable?)")
    comm = C.c_void_p()
    r = dll.ommxRcclCommInitRank(buf, 128, rank, world, C.byref(comm))
    joined = 1 if r == 0 else 0
    if world > 1:
        # every rank learns whether ALL joined: a caller can fall back to its own transport collectively
        t = torch.tensor([joined], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        joined_all = int(t.item())
    else:
        joined_all = joined
    if not joined_all:
        if joined:
            dll.ommxRcclCommDestroy(comm)
        raise RuntimeError("ommxRcclCommInitRank failed (this rank: %d)" % r)
    return comm


class CollectivesComm:
    """Communicator for ommxShardedBakeRccl whose two collectives run through torch.distributed on ANY initialised backend
    (ommxCommFromCollectives): the library hands device pointers and its stream to the callbacks below.  Blocking implementation: the
    library's stream is synchronised before the exchange and torch's after it.  `.handle` goes where an ommxRcclComm goes; destroy()
    when done (the object keeps the ctypes callbacks alive)."""

    def __init__(self, dll, torch, dist, rank, world):
        ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
        ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

        class Collectives(C.Structure):
            _fields_ = [("allReduceU32", ALLREDUCE), ("allGatherBytes", ALLGATHER), ("user", C.c_void_p)]

        def wait_for(stream):
            if stream:
                torch.cuda.ExternalStream(int(stream)).synchronize()
            else:
                torch.cuda.synchronize()

        def all_reduce(_user, send, recv, count, op, stream):
            try:
                if count == 0:
                    return 0
                wait_for(stream)
                src = device_tensor(torch, send, count, torch.int32)
                dst = src if recv == send else device_tensor(torch, recv, count, torch.int32)
                if op == 0:   # SUM: the int32 view wraps exactly like uint32
                    t = src if recv == send else src.clone()
                    if world > 1:
                        dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    if recv != send:
                        dst.copy_(t)
                else:         # MAX / MIN compare as UNSIGNED words: through int64
                    t = src.to(torch.int64) & 0xFFFFFFFF
                    if world > 1:
                        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.MIN)
                    dst.copy_(torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32))
                torch.cuda.current_stream().synchronize()
                return 0
            except Exception:   # (an exception must not unwind through the C frames)
                import traceback
                traceback.print_exc()
                return 1

        def all_gather(_user, send, recv, nbytes, stream):
            try:
                if nbytes == 0:
                    return 0
                wait_for(stream)
                src = device_tensor(torch, send, nbytes, torch.uint8)
                out = device_tensor(torch, recv, nbytes * world, torch.uint8)
                if world == 1:
                    out.copy_(src)
                elif dist.get_backend() == "gloo":
                    dist.all_gather(list(out.view(world, -1).unbind(0)), src)
                else:
                    dist.all_gather_into_tensor(out, src)
                torch.cuda.current_stream().synchronize()
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        self._callbacks = (ALLREDUCE(all_reduce), ALLGATHER(all_gather))
        self._table = Collectives(self._callbacks[0], self._callbacks[1], None)
        self._dll = dll
        dll.ommxCommFromCollectives.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        dll.ommxRcclCommDestroy.argtypes = [C.c_void_p]
        self.handle = C.c_void_p()
        r = dll.ommxCommFromCollectives(C.byref(self._table), rank, world, C.byref(self.handle))
        if r != 0:
            raise RuntimeError("ommxCommFromCollectives failed: %d" % r)

    def destroy(self):
        if self.handle:
            self._dll.ommxRcclCommDestroy(self.handle)
            self.handle = C.c_void_p()


def sharded_bake_rccl(dll, baker, desc_ptr, comm):
    """One sharded bake with the collectives inside the library; returns the ommxDeviceBakeResult handle."""
    dll.ommxShardedBakeRccl.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    out = C.c_void_p()
    r = dll.ommxShardedBakeRccl(baker, desc_ptr, comm, C.byref(out))
    if r != 0:
        raise RuntimeError("ommxShardedBakeRccl failed: %d" % r)
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
