"""N > 1 path on CPU: submap partitioning + the best-match all-reduce on gloo,
world_size 2 (the GPU path uses the same code on RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cartographer_amd import sharding


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 512, 513):
        for w in (1, 2, 3, 8):
            ranges = [sharding.shard_range(n, r, w) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_key_roundtrip_and_order():
    a = sharding.pack_best_key([0, 1, 1], [0.9, 0.61, 0.75], 40)
    b = sharding.pack_best_key([1, 0, 0], [0.74, 0.99, 0.99], 3)
    assert sharding.unpack_best_key(a) == (pytest.approx(0.75), 42)
    assert sharding.unpack_best_key(b) == (pytest.approx(0.74), 3)
    assert a > b                      # keys order like the scores
    assert sharding.pack_best_key([0, 0], [0.5, 0.6], 0) == sharding.NOT_FOUND
    assert sharding.unpack_best_key(sharding.NOT_FOUND) == (None, None)
    assert sharding.NOT_FOUND < min(a, b)          # any found match beats "nothing found"
    # a match at submap 0 with the smallest positive score is still distinct from NOT_FOUND
    z = sharding.pack_best_key([1], [np.float32(1e-45)], 0)
    assert z > sharding.NOT_FOUND and sharding.unpack_best_key(z)[1] == 0
    # equal scores: the LOWER submap id wins the max, as it does within a rank -- the result
    # does not depend on how the submaps are sharded
    c = sharding.pack_best_key([1], [0.75], 100)
    assert max(a, c) == a and sharding.unpack_best_key(max(a, c))[1] == 42


def test_all_gather_results_single_process():
    found, scores = np.array([1, 0, 1], np.int32), np.array([0.7, 0.2, 0.61], np.float32)
    poses = np.arange(9, dtype=np.float64).reshape(3, 3)
    f, s, p = sharding.all_gather_results(found, scores, poses, 3, 0, 1)
    assert np.array_equal(f, found) and np.array_equal(s, scores) and np.array_equal(p, poses)


def _worker(rank, world_size, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    num_submaps = 9
    begin, end = sharding.shard_range(num_submaps, rank, world_size)
    rng = np.random.default_rng(5)
    all_scores = rng.uniform(0.3, 0.9, num_submaps).astype(np.float32)
    all_found = (all_scores > 0.55).astype(np.int32)
    key = sharding.pack_best_key(all_found[begin:end], all_scores[begin:end], begin)
    best = sharding.all_reduce_best(key)
    expect_id = int(np.argmax(np.where(all_found > 0, all_scores, -1)))
    score, gid = sharding.unpack_best_key(best)
    ok = (gid == expect_id) and abs(score - float(all_scores[expect_id])) < 1e-7
    # every rank learns every submap's optional constraint, bit-exactly
    all_poses = rng.uniform(-5, 5, (num_submaps, 3))
    f, sc, po = sharding.all_gather_results(all_found[begin:end], all_scores[begin:end],
                                            all_poses[begin:end], num_submaps, rank, world_size)
    ok = ok and np.array_equal(f, all_found) and np.array_equal(sc, all_scores) and \
        np.array_equal(po, all_poses)
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_all_reduce_best_gloo_world_size_2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] and out[1]


def test_c_abi_partition_and_key_equal_the_python_rules():
    """cmx_shard_range / cmx_pack_best_key / cmx_unpack_best_key (plain host arithmetic in the
    product library, no device needed) against cartographer_amd.sharding: the single-process
    multi-GPU path (cmx_comm) and the one-process-per-GPU path (torch.distributed) partition and
    merge identically."""
    import ctypes as C
    from cartographer_amd import _lib
    L = _lib.lib()
    for n in (0, 1, 7, 512, 513):
        for w in (1, 2, 3, 8):
            for r in range(w):
                b, e = C.c_int64(), C.c_int64()
                L.cmx_shard_range(n, r, w, C.byref(b), C.byref(e))
                assert (b.value, e.value) == sharding.shard_range(n, r, w)
    rng = np.random.default_rng(2)
    for trial in range(200):
        m = int(rng.integers(1, 20))
        found = (rng.random(m) < 0.5).astype(np.int32)
        scores = rng.uniform(0.01, 1.0, m).astype(np.float32)
        if trial % 5 == 0 and m > 2:
            scores[1:] = scores[0]              # ties: the lowest index must win
        first = int(rng.integers(0, 1000))
        key = L.cmx_pack_best_key(found.ctypes.data, scores.ctypes.data, m, first)
        assert key == sharding.pack_best_key(found, scores, first)
        f, s, g = C.c_int32(), C.c_float(), C.c_int64()
        L.cmx_unpack_best_key(key, C.byref(f), C.byref(s), C.byref(g))
        score, gid = sharding.unpack_best_key(key)
        if score is None:
            assert f.value == 0 and g.value == -1
        else:
            assert f.value == 1 and g.value == gid and np.float32(s.value) == np.float32(score)
    # merging per-block keys with max equals the key of the whole list, for every partition
    found = np.array([0, 1, 1, 0, 1, 1, 1], np.int32)
    scores = np.array([0.9, 0.7, 0.8, 0.95, 0.8, 0.8, 0.6], np.float32)
    whole = L.cmx_pack_best_key(found.ctypes.data, scores.ctypes.data, 7, 0)
    for w in (1, 2, 3, 7):
        keys = []
        for r in range(w):
            b, e = sharding.shard_range(7, r, w)
            fb, sb = np.ascontiguousarray(found[b:e]), np.ascontiguousarray(scores[b:e])
            keys.append(L.cmx_pack_best_key(fb.ctypes.data, sb.ctypes.data, e - b, b))
        assert max(keys) == whole and sharding.unpack_best_key(whole)[1] == 2
