"""CPU tests (gloo, world_size 2) of the row-sharded multi-GPU path: nnz-prefix partition,
shard extraction (rebased rowptr, global columns), equal and unequal all-gather assembly.
The local SpMV is injected (the oracle) so that the sharding/collective logic runs without a
GPU; on GPUs the default local_spmv is the HIP path (spblas-reference_amd/sharded.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import spblas_reference_amd as sp
from oracle import oracle
from spblas_reference_amd import generate, sharded


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_local(info, a_local, x, y_local):
    y = oracle.spmv(tuple(a_local.shape()), a_local.rowptr().numpy(), a_local.colind().numpy(),
                    a_local.values().numpy(), x.numpy())
    y_local.copy_(torch.from_numpy(y))


def _worker(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n, nnz, by_nnz, dtype = case[:5]
        gather = case[5] if len(case) > 5 else "auto"
        values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz, seed=3, dtype=dtype)
        if by_nnz:  # skew the matrix: make the first rows heavy so nnz-balanced shards are unequal
            lens = np.diff(rowptr)
            order = np.argsort(-lens, kind="stable")
            # rebuild CSR with rows sorted by decreasing length (still a valid test matrix)
            new_rp = np.concatenate([[0], np.cumsum(lens[order])]).astype(np.int32)
            idx = np.concatenate([np.arange(rowptr[r], rowptr[r + 1]) for r in order]) if nnz else np.zeros(0, np.int64)
            values, colind, rowptr = values[idx], colind[idx], new_rp
        t = torch.from_numpy
        rp_t = t(rowptr)
        bounds = sharded.partition_rows_by_nnz(rp_t, world) if by_nnz else sharded.partition_rows_even(m, world)
        a_local = sharded.shard_csr(t(values), rp_t, t(colind), shape, bounds[rank], bounds[rank + 1])
        assert int(a_local.rowptr()[0]) == 0 and a_local.shape() == (bounds[rank + 1] - bounds[rank], n)
        op = sharded.ShardedSpMV(a_local, bounds, local_spmv=_oracle_local, gather=gather)
        assert op.gather_mode == ("inplace" if len({bounds[i + 1] - bounds[i] for i in range(world)}) == 1 else
                                  ("p2p" if gather == "auto" else gather))
        x = t(np.random.default_rng(5).random(n).astype(dtype))
        y = op.step(x).numpy().copy()
        y2 = op.step(x).numpy().copy()  # a second step reuses the buffers
        y_ref = oracle.spmv(shape, rowptr, colind, values, x.numpy())
        # gathered result == single-process result bit for bit (same per-row arithmetic)
        assert np.array_equal(y, y_ref) and np.array_equal(y2, y_ref)
        np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array(bounds))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", [(1000, 100, 100, False, np.float32), (100, 1000, 10000, False, np.float64),
                                  (999, 640, 20000, True, np.float32), (40, 40, 1000, True, np.float64),
                                  (999, 640, 20000, True, np.float32, "padded"), (40, 40, 1000, True, np.float64, "p2p")])
def test_row_sharded_spmv_world2_gloo(case, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    b0, b1 = np.load(tmp_path / "ok_0.npy"), np.load(tmp_path / "ok_1.npy")
    assert np.array_equal(b0, b1) and b0[0] == 0 and b0[-1] == case[0]


def _worker_pipelined(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n, nnz, chunks = 960, 500, 15000, 4
        values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz, seed=8)
        t = torch.from_numpy
        ranges = sharded.striped_row_ranges(m, world, chunks)
        assert ranges is not None and ranges[0][0] == (0, 120) and ranges[-1][-1] == (840, 960)
        a_chunks = [sharded.shard_csr(t(values), t(rowptr), t(colind), shape, *ranges[c][rank]) for c in range(chunks)]
        op = sharded.PipelinedShardedSpMV(a_chunks, ranges, local_spmv=_oracle_local)
        x = t(np.random.default_rng(6).random(n).astype(np.float32))
        for _ in range(2):
            y = op.step(x).numpy().copy()
            assert np.array_equal(y, oracle.spmv(shape, rowptr, colind, values, x.numpy()))
        np.save(os.path.join(out_dir, f"pipe_{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


def _worker_overlapped(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n, nnz, chunks = 960, 500, 15000, 4
        values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz, seed=9)
        t = torch.from_numpy
        ranges = sharded.striped_row_ranges(m, world, chunks)
        L = ranges[0][0][1] - ranges[0][0][0]
        parts = [sharded.shard_csr(t(values), t(rowptr), t(colind), shape, *ranges[c][rank]) for c in range(chunks)]
        # the rank's stripes back to back = its local matrix
        lens = torch.cat([p.rowptr()[1:] - p.rowptr()[:-1] for p in parts])
        rp = torch.zeros(L * chunks + 1, dtype=torch.int32)
        rp[1:] = torch.cumsum(lens, 0)
        a_local = sp.csr_view(torch.cat([p.values() for p in parts]), rp, torch.cat([p.colind() for p in parts]),
                              (L * chunks, n), int(rp[-1]))
        calls = []

        def stages(x):  # oracle-backed stand-in for plan.bind_stages
            state = {}

            def expand():
                state["y"] = oracle.spmv(tuple(a_local.shape()), a_local.rowptr().numpy(), a_local.colind().numpy(),
                                         a_local.values().numpy(), x.numpy())
                calls.append("E")

            def reduce(c, y_stripe_local):
                y_stripe_local.copy_(torch.from_numpy(state["y"][c * L:(c + 1) * L]))
                calls.append(c)

            return expand, reduce

        op = sharded.OverlappedShardedSpMV(a_local, ranges, stages=stages)
        x = t(np.random.default_rng(6).random(n).astype(np.float32))
        y = op.step(x).numpy().copy()
        assert calls == ["E", 0, 1, 2, 3]
        assert np.array_equal(y, oracle.spmv(shape, rowptr, colind, values, x.numpy()))
        np.save(os.path.join(out_dir, f"ovl_{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


def test_overlapped_single_plan_striped_world2_gloo(tmp_path):
    mp.spawn(_worker_overlapped, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ovl_0.npy").exists() and (tmp_path / "ovl_1.npy").exists()


def test_pipelined_striped_all_gather_world2_gloo(tmp_path):
    mp.spawn(_worker_pipelined, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "pipe_0.npy").exists() and (tmp_path / "pipe_1.npy").exists()
    assert sharded.striped_row_ranges(10, 2, 4) is None  # not divisible -> caller falls back to one gather


def test_partition_rows_by_nnz_balances_nnz():
    rowptr = torch.tensor(np.concatenate([[0], np.cumsum([1000] * 4 + [1] * 4000)]))
    b = sharded.partition_rows_by_nnz(rowptr, 8, row_weight=0)
    assert b[0] == 0 and b[-1] == 4004 and all(b[i] <= b[i + 1] for i in range(8))
    per = [int(rowptr[b[i + 1]] - rowptr[b[i]]) for i in range(8)]
    assert max(per) <= 1000 + 1  # no shard exceeds nnz/P by more than one (heavy) row
    assert sharded.partition_rows_even(10, 4) == [0, 2, 5, 7, 10]
    # the default weighs a row like one entry (its offset + its element of y cost what an entry costs): the four heavy rows
    # and the 4 000 single-entry rows hold 8 000 entries + 4 004 rows = 12 004 units, ~1 500 per shard
    b = sharded.partition_rows_by_nnz(rowptr, 8)
    assert b[0] == 0 and b[-1] == 4004 and all(b[i] <= b[i + 1] for i in range(8))
    work = [int(rowptr[b[i + 1]] - rowptr[b[i]]) + (b[i + 1] - b[i]) for i in range(8)]
    assert max(work) <= 12004 // 8 + 1001 and min(work) >= 500
    # (the shards of single-entry rows now hold fewer entries than an equal split of the entries would give them)
    assert int(rowptr[b[8]] - rowptr[b[7]]) < 1000


def test_sharded_default_compute_is_the_hip_path():
    assert sharded._hip_local_spmv.__module__.endswith("sharded")
    a = sp.csr_view(torch.ones(1), torch.tensor([0, 1], dtype=torch.int32), torch.zeros(1, dtype=torch.int32), (1, 1), 1)
    op = sharded.ShardedSpMV(a, [0, 1], inspect=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        op.step(torch.ones(1))


# --------------------------------------------------------------------------- SpMM / SpGEMM over row shards
def _oracle_local_spmm(info, a_local, b, c_local):
    c = oracle.spmm(tuple(a_local.shape()), a_local.rowptr().numpy(), a_local.colind().numpy(), a_local.values().numpy(),
                    b.numpy())
    c_local.copy_(torch.from_numpy(np.asarray(c).reshape(c_local.shape)))


def _oracle_local_spgemm(a_local, b):
    ash, bsh = tuple(a_local.shape()), tuple(b.shape())
    ar, ac, av = a_local.rowptr().numpy(), a_local.colind().numpy(), a_local.values().numpy()
    br, bc, bv = b.rowptr().numpy(), b.colind().numpy(), b.values().numpy()
    n_ref, _ = oracle.spgemm_symbolic(ash, ar, ac, bsh, br, bc)
    cr, cc, cv = oracle.spgemm_numeric(ash, ar, ac, av, bsh, br, bc, bv, capacity=n_ref)
    return torch.from_numpy(cr), torch.from_numpy(cc[:n_ref].copy()), torch.from_numpy(cv[:n_ref].copy())


def _worker_spmm_spgemm(rank, world, port, by_nnz, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, k, n, nnz = 300, 200, 12, 4000
        values, rowptr, colind, shape, _ = generate.generate_csr(m, k, nnz, seed=7)
        t = torch.from_numpy
        rp_t = t(rowptr)
        bounds = sharded.partition_rows_by_nnz(rp_t, world) if by_nnz else sharded.partition_rows_even(m, world)
        a_local = sharded.shard_csr(t(values), rp_t, t(colind), shape, bounds[rank], bounds[rank + 1])
        # SpMM: B replicated, C row blocks, optional gather
        b = t(np.random.default_rng(9).random((k, n)).astype(np.float32))
        op = sharded.ShardedSpMM(a_local, bounds, n, local_spmm=_oracle_local_spmm)
        c_loc = op.local(b).numpy().copy()
        c_ref = np.asarray(oracle.spmm(shape, rowptr, colind, values, b.numpy())).reshape(m, n)
        assert np.array_equal(c_loc, c_ref[bounds[rank]:bounds[rank + 1]])
        assert np.array_equal(op.gather_c().numpy(), c_ref)
        # SpGEMM: A row-sharded, B (k x m) replicated, C blocks disjoint, nnz offsets by exclusive scan
        bv, br, bc, bsh, _ = generate.generate_csr(k, m, nnz, seed=8)
        b_csr = sp.csr_view(t(bv), t(br), t(bc), bsh, nnz)
        g = sharded.ShardedSpGEMM(a_local, b_csr, bounds, local_spgemm=_oracle_local_spgemm)
        (cr, cc, cv), (off, total) = g.compute()
        n_ref, _ = oracle.spgemm_symbolic(shape, rowptr, colind, bsh, br, bc)
        fr, fc, fv = oracle.spgemm_numeric(shape, rowptr, colind, values, bsh, br, bc, bv, capacity=n_ref)
        lo, hi = int(fr[bounds[rank]]), int(fr[bounds[rank + 1]])
        assert total == n_ref and off == lo
        assert np.array_equal(cr.numpy(), fr[bounds[rank]:bounds[rank + 1] + 1] - lo)
        assert np.array_equal(cc.numpy(), fc[lo:hi]) and np.array_equal(cv.numpy(), fv[lo:hi])
        np.save(os.path.join(out_dir, f"ok_mm_{rank}.npy"), np.array(bounds))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("by_nnz", [False, True])
def test_row_sharded_spmm_and_spgemm_world2_gloo(by_nnz, tmp_path):
    mp.spawn(_worker_spmm_spgemm, args=(2, _free_port(), by_nnz, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), f"ok_mm_{r}.npy")) for r in range(2))
