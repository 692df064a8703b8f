"""Worker of tests/test_gpu_fused_sharding.py: run with torch.distributed.run, WORLD_SIZE processes that
ALL use cuda:0 (the GPU box has one device).  Control plane: gloo; data plane: hipIpc-mapped buffers and
peer stores from the reduce kernels -- exactly the code path of an 8-GPU node, minus xGMI."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp  # noqa: E402
from oracle import oracle  # noqa: E402
from spblas_reference_amd import generate, sharded  # noqa: E402


def main():
    dist.init_process_group(os.environ.get("FUSED_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    m, n, per_row = 64000 * world, 90000, 9
    dtype = torch.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else torch.float32
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, n, per_row, dtype=dtype, seed=0, device=dev)
    bounds = sharded.partition_rows_even(m, world)
    a_loc = sharded.shard_csr(values, rowptr, colind, shape, bounds[rank], bounds[rank + 1])
    stripes = int(os.environ.get("FUSED_STRIPES", "1"))
    op = sharded.FusedShardedSpMV(a_loc, bounds, alg=sp._capi.SPMV_SLICED, timeout_ms=15000, stripes=stripes)
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.rand(n, dtype=dtype, device=dev, generator=g)
    vh, rh, ch, xh = values.cpu().numpy(), rowptr.cpu().numpy(), colind.cpu().numpy(), x.cpu().numpy()
    ref = oracle.spmv(shape, rh, ch, vh, xh)
    absrow = oracle.spmv_absrow(rh, ch, vh, xh)
    tol = 1e-6 if dtype == torch.float32 else 1e-12
    outs = []
    for it in range(4):  # exercises both y buffers twice
        y = op.step(x)
        torch.cuda.synchronize()
        op.check_status()
        yh = y.cpu().numpy()
        err = np.abs(yh.astype(np.float64) - ref.astype(np.float64))
        assert (err <= tol * absrow + 1e-30).all(), f"rank {rank} step {it}: parity failed ({(err / (absrow + 1e-30)).max()})"
        outs.append(yh.copy())
        x = x.clone()  # a new tensor object every step: the x pointer is re-bound
    assert all(np.array_equal(outs[0], o) for o in outs[1:]), "steps disagree"
    # a DIFFERENT x on every step: a row a peer failed to deliver would still hold the previous step's value
    for it, scale in enumerate((2.0, -0.5, 3.0)):
        y = op.step(x * scale)
        torch.cuda.synchronize()
        op.check_status()
        err = np.abs(y.cpu().numpy().astype(np.float64) - scale * ref.astype(np.float64))
        assert (err <= tol * abs(scale) * absrow + 1e-30).all(), f"rank {rank} scaled step {it}: parity failed"
    # every rank must hold the SAME bits (each row is computed once, by its owner)
    gathered = [None] * world
    dist.all_gather_object(gathered, outs[0].tobytes())
    assert all(gb == gathered[0] for gb in gathered), "ranks hold different y"
    op.close()
    if os.environ.get("FUSED_VFREE"):
        # Round 6: the fused exchange on a VALUE-FREE local plan (what a plain inspected csr_view gets at cfg2's size; forced here
        # through the test hooks).  The plan holds no copy of A's values: every step multiplies with the caller's array as it is
        # THEN -- so values rewritten in place between two steps must show in every rank's copy of y, with no update call.
        os.environ["SPBLAS_GFX950_PB_VFREE"] = "2"
        os.environ["SPBLAS_GFX950_PB_VF_ROWS"] = "500"
        y_loc = torch.empty(bounds[rank + 1] - bounds[rank], dtype=dtype, device=dev)
        info_vf = sp.multiply_inspect(a_loc, x, y_loc, alg=sp._capi.SPMV_SLICED)
        del os.environ["SPBLAS_GFX950_PB_VFREE"], os.environ["SPBLAS_GFX950_PB_VF_ROWS"]
        assert info_vf.state_.sliced_info()["value_free"] == 1
        saved_vals = a_loc.values().clone()
        op_vf = sharded.FusedShardedSpMV(a_loc, bounds, info=info_vf, timeout_ms=15000, stripes=stripes)
        assert op_vf.value_free
        for it, vscale in enumerate((1.0, -0.5, 3.0)):
            if vscale != 1.0:
                a_loc.values().mul_(vscale)      # in place, every rank its own rows; no update_values anywhere
            total = float(np.prod((1.0, -0.5, 3.0)[:it + 1]))
            y = op_vf.step(x)
            torch.cuda.synchronize()
            op_vf.check_status()
            err = np.abs(y.cpu().numpy().astype(np.float64) - total * ref.astype(np.float64))
            assert (err <= tol * abs(total) * absrow + 1e-30).all(), f"rank {rank} value-free step {it}: parity failed"
        gathered = [None] * world
        dist.all_gather_object(gathered, y.cpu().numpy().tobytes())
        assert all(gb == gathered[0] for gb in gathered), "value-free plans: ranks hold different y"
        a_loc.values().copy_(saved_vals)                 # (bit-exact restore: the rest of the worker compares bits)
        op_vf.close()
        print("FUSED_VFREE_OK", flush=True)
    # the selection helper bench.py uses: adopted only when bit-identical to a reference path on every rank
    plans = []
    for q in range(world):
        a_q = sharded.shard_csr(values, rowptr, colind, shape, bounds[q], bounds[q + 1])
        y_q = torch.empty(bounds[q + 1] - bounds[q], dtype=dtype, device=dev)
        plans.append((sp.multiply_inspect(a_q, x, y_q, alg=sp._capi.SPMV_SLICED), a_q, y_q))

    def reference_step(xk=None, corrupt=False):
        xk = x if xk is None else xk
        for info_q, a_q, y_q in plans:
            sp.multiply(info_q, a_q, xk, y_q)
        y = torch.cat([p[2] for p in plans])
        if corrupt and rank == world - 1:
            y[5] += 1
        return y

    chosen = sharded.try_fused(a_loc, bounds, x, reference_step, alg=sp._capi.SPMV_SLICED,
                               log=lambda msg: print(f"[rank {rank}] {msg}", flush=True))
    assert chosen is not None, "try_fused rejected a correct fused path"
    assert torch.allclose(chosen.step(x), reference_step(), rtol=1e-4)
    chosen.close()
    # reusing the reference path's own plan, against a reference that gathers every rank's OWN shard
    # (what the RCCL all-gather path produces): the two must agree bit for bit
    def reference_step_gathered(xk=None):
        xk = x if xk is None else xk
        info_r, a_r, y_r = plans[rank]
        sp.multiply(info_r, a_r, xk, y_r)
        torch.cuda.synchronize()
        parts = [None] * world
        dist.all_gather_object(parts, y_r.cpu().numpy())
        return torch.from_numpy(np.concatenate(parts)).to(dev)

    chosen = sharded.try_fused(a_loc, bounds, x, reference_step_gathered, info=plans[rank][0], stripes=stripes)
    assert chosen is not None, "try_fused rejected the plan-sharing fused path"
    assert torch.equal(chosen.step(x), reference_step_gathered())
    # throughput form (independent right-hand sides): the wait for step k-1 sits inside step k, results after flush()
    # are the same bits; a different vector per step, mixed with dependent steps, so that a late or lost peer store shows
    refs = {sc: reference_step_gathered(x * sc) for sc in (1.0, -2.0, 0.25, 4.0)}
    for sc in (1.0, -2.0, 0.25):
        chosen.step_pipelined(x * sc)
    y_last = chosen.flush()
    torch.cuda.synchronize()
    chosen.check_status()
    assert torch.equal(y_last, refs[0.25]), "pipelined steps: last y differs"
    assert torch.equal(chosen.step(x * 4.0), refs[4.0]), "dependent step after pipelined ones differs"
    chosen.step_pipelined(x * -2.0)
    assert torch.equal(chosen.step(x), refs[1.0]), "step() must flush a pending pipelined step first"
    chosen.close()
    # a disagreement on ONE rank must make EVERY rank fall back
    assert sharded.try_fused(a_loc, bounds, x, lambda xk: reference_step(xk, True), alg=sp._capi.SPMV_SLICED) is None
    # a rank that cannot build a SLICED plan (forced row-block here) makes every rank fall back, no hang
    bad_alg = sp._capi.SPMV_ROWBLOCK if rank == 0 else sp._capi.SPMV_SLICED
    assert sharded.try_fused(a_loc, bounds, x, reference_step, alg=bad_alg) is None
    # The dependent iteration without a step barrier (round 4): y_{k+1} = alpha * A y_k, the peers' rows of step k
    # arriving chunk by chunk behind the expand of step k + 1 (spblas_gfx950_spmv_step_bcast_chunked).  Needs a square
    # matrix; compared bit for bit with the same chain through step() (full barrier after every step) on the same plan;
    # then once more with chunk 1 of every step deliberately late on ONE rank: the other ranks' expands have to wait
    # for it (and report that they did), the bits stay the same.
    chunks = int(os.environ.get("FUSED_CHUNKS", "0"))
    if chunks:
        msq = 64000 * world
        v2, rp2, ci2, shape2, _ = generate.uniform_csr_device(msq, msq, per_row, dtype=dtype, seed=3, device=dev)
        b2 = sharded.partition_rows_even(msq, world)
        if os.environ.get("FUSED_TRIANGULAR", "0") == "1":
            # lower BLOCK triangular: the rows of rank q only have columns below the end of q's own block, so rank 0 reads
            # no peer's rows at all and rank q never reads those of the ranks above it -- no x slice of its expand waits for
            # them.  With the last rank's chunk made late below, the lower ranks run ahead unless the step itself orders
            # them behind EVERY peer (round-4 advisor finding: a write-after-read race on the peers' copies of y).
            row_of = torch.repeat_interleave(torch.arange(msq, device=dev), (rp2[1:] - rp2[:-1]).long())
            limit = (torch.div(row_of, msq // world, rounding_mode="floor") + 1) * (msq // world)
            ci2 = (ci2.long() % limit).to(torch.int32)
        a2 = sharded.shard_csr(v2, rp2, ci2, shape2, b2[rank], b2[rank + 1])
        op2 = sharded.FusedShardedSpMV(a2, b2, alg=sp._capi.SPMV_SLICED, timeout_ms=20000, chunks=chunks, shared_device=True)
        x0 = torch.rand(msq, dtype=dtype, device=dev, generator=g)
        nsteps = 7  # (entries in [0, 1): y grows by ~2.2x per step, far from overflow)
        y = op2.step(x0).clone()
        ref_chain = [y.clone()]
        for _ in range(nsteps - 1):
            y = op2.step(y.clone()).clone()
            ref_chain.append(y.clone())
        torch.cuda.synchronize()
        op2.check_status()
        for delay in ("0", "4000"):
            if rank == world - 1:
                os.environ["SPBLAS_GFX950_CHUNK_DELAY_US"] = delay
            dist.barrier()
            y = op2.step_dependent(x0, alpha=1.0)
            for _ in range(nsteps - 1):
                y = op2.step_dependent(alpha=1.0)
            y = op2.flush_chain()
            torch.cuda.synchronize()
            op2.check_status()
            assert torch.equal(y, ref_chain[-1]), f"rank {rank}: chunked chain differs from the barrier chain (delay {delay})"
            waited = op2.chunk_wait_us()
            # Some expand must have waited for the late chunk (and report it).  Not EVERY rank's: with the ranks sharing one
            # GPU their kernels are time-sliced, and a rank whose expand was only scheduled after the late chunk had landed
            # never waits (seen once in five runs of the 8-rank case) -- so the largest wait over the other ranks is checked.
            w_all = torch.tensor([waited if rank != world - 1 else 0.0], dtype=torch.float64)
            dist.all_reduce(w_all, op=dist.ReduceOp.MAX)
            if delay != "0" and world > 1 and chunks > 1:
                assert float(w_all.item()) >= 2000.0, f"no expand on any rank waited for the late chunk ({float(w_all.item())} us)"
            # a step() after the chain flushes it; both copies of y are in use again afterwards
            assert torch.equal(op2.step(x0), ref_chain[0])
        os.environ.pop("SPBLAS_GFX950_CHUNK_DELAY_US", None)
        # every rank holds the same bits
        parts = [None] * world
        dist.all_gather_object(parts, ref_chain[-1].cpu().numpy().tobytes())
        assert all(pb == parts[0] for pb in parts)
        op2.close()
    dist.barrier()
    if rank == 0:
        print("FUSED_OK", world, str(dtype), "stripes", stripes, "chunks", chunks)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
