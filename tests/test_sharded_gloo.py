"""world_size-2 gloo test (CPU) of the multi-GPU exchange protocol driven by omm_amd/sharded.py.

No GPU here, so the per-rank compute is emulated with numpy on synthetic work items; what is tested is the protocol the
library's ommxSharded* phases rely on: (1) a SUM all-reduce of metadata words that are zero outside the owner's range is a
merge, (2) the padded all-gather of per-rank block contributions + scatter by (owner, contribution offset) reproduces the
single-process arrayData.  The layout arithmetic mirrors tail_kernels.hip (owner_of_position, run_shard_layout)."""
import os
import socket
import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def level_bounds(active_start, world):
    """bounds[l][r] of bake_core: rank r owns [a + cnt*r/world, a + cnt*(r+1)/world) of each level group"""
    b = []
    for l in range(len(active_start) - 1):
        a, cnt = active_start[l], active_start[l + 1] - active_start[l]
        b.append([a + cnt * r // world for r in range(world + 1)])
    return b


def owner_of_position(bounds, p, world):
    l = 0
    while l + 1 < len(bounds) and p >= bounds[l + 1][0]:
        l += 1
    r = 0
    while r + 1 < world and p >= bounds[l][r + 1]:
        r += 1
    return r


def _worker(rank, world, port, result_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import omm_amd.sharded as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(99)  # same stream on every rank
    # synthetic active lists: 3 level groups
    active_start = [0, 0, 137, 137, 600, 1000]
    A = active_start[-1]
    bounds = level_bounds(active_start, world)
    owners = np.array([owner_of_position(bounds, p, world) for p in range(A)])
    full_meta = rng.integers(1, 2 ** 32, size=4 * A, dtype=np.uint64).astype(np.uint32)
    mine = np.tile(owners == rank, 4)
    words = torch.from_numpy(np.where(mine, full_meta, 0).astype(np.uint32).view(np.int32).copy())
    sh.allreduce_words_(dist, words)
    ok_meta = np.array_equal(words.numpy().view(np.uint32), full_meta)

    # surviving blocks in final order: item position, size; per-rank contribution layout (run_shard_layout)
    E = 300
    pos = rng.permutation(A)[:E]
    sizes = (4 ** rng.integers(1, 5, size=E)).astype(np.int64)
    blocks = [rng.integers(0, 256, size=int(s), dtype=np.uint8) for s in sizes]
    dst_ofs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    own = owners[pos]
    totals = [int(sizes[own == r].sum()) for r in range(world)]
    stride = (max(totals) + 255) // 256 * 256
    cofs = np.zeros(E, np.int64)
    for r in range(world):
        m = own == r
        cofs[m] = np.concatenate([[0], np.cumsum(sizes[m])[:-1]])
    contrib = np.zeros(stride, np.uint8)
    for j in range(E):
        if own[j] == rank:
            contrib[cofs[j]:cofs[j] + sizes[j]] = blocks[j]
    gathered = sh.allgather_padded(dist, torch, torch.from_numpy(contrib), world).numpy()
    merged = np.zeros(int(sizes.sum()), np.uint8)
    for j in range(E):
        src = own[j] * stride + cofs[j]
        merged[dst_ofs[j]:dst_ofs[j] + sizes[j]] = gathered[src:src + sizes[j]]
    ok_data = np.array_equal(merged, np.concatenate(blocks))
    open(os.path.join(result_dir, "rank%d" % rank), "w").write("%d %d" % (ok_meta, ok_data))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_protocol(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "1 1"


def _bootstrap_worker(rank, world, port, result_dir):
    import sys, ctypes as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import omm_amd.sharded as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dll = C.CDLL(os.path.join(root, "omm_amd", "lib", "libomm-lib.so"))
    try:
        comm = sh.rccl_comm(dll, torch, dist, rank, world)
        outcome = "joined"
        dll.ommxRcclCommDestroy(comm)
    except RuntimeError as e:
        outcome = "raised"
    open(os.path.join(result_dir, "boot%d" % rank), "w").write(outcome)
    dist.barrier()
    dist.destroy_process_group()


def test_communicator_bootstrap_fails_on_every_rank_or_on_none(tmp_path):
    """rccl_comm() on a box without GPUs: the RCCL communicator cannot be created, and that must surface as the same exception on BOTH
    ranks (bench.py then falls back to torch.distributed collectively) -- never as one rank raising while the other waits in a collective"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_bootstrap_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [open(os.path.join(str(tmp_path), "boot%d" % r)).read() for r in range(2)]
    assert got[0] == got[1], got
    if not torch.cuda.is_available():
        assert got == ["raised", "raised"], got


def test_bench_gpus_n_relaunches_itself_under_the_launcher(monkeypatch):
    """`python bench.py --gpus 8` as typed (no WORLD_SIZE in the environment) must not die on an assert: it re-executes the same command line
    under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 (the driver's own launcher command line sets WORLD_SIZE and is untouched)"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    seen = {}

    class Stop(Exception):
        pass

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, list(argv)
        raise Stop()
    monkeypatch.setattr(os, "execv", fake_execv)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    with pytest.raises(Stop):
        bench.main()
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[1:3] == ["-m", "torch.distributed.run"]
    assert a[a.index("--nproc-per-node") + 1] == "8" and a[a.index("--master-addr") + 1] == "127.0.0.1" and int(a[a.index("--master-port") + 1]) > 0
    assert os.path.samefile(a[a.index("--master-port") + 2], os.path.join(root, "bench.py"))
    assert a[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
