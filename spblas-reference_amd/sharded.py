"""Row-sharded multi-GPU SpMV: one process per GPU, y all-gathered over RCCL/xGMI.

The reference is single-device (no collectives anywhere); this is
new capability layered on the same multiply() path.  CSR rows are independent
(/root/reference/include/spblas/algorithms/multiply_impl.hpp:48-52), so each rank owns a
contiguous row range chosen by NNZ PREFIX on rowptr (mandatory for power-law inputs),
keeps x replicated, computes its slice of y with the single-GPU kernel and exchanges
slices with ONE all-gather per step -- the only data-path collective.  In an iterative
solver y is the next x, which is why the gather belongs to the step.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): an all-gather moves each shard
once over each link, so bigger, fewer collectives win; y shards are gathered in a single
call, in place when the shards are equal-sized.
"""
import os

import torch
import torch.distributed as dist

from . import api


def partition_rows_by_nnz(rowptr, world_size, row_weight=1):
    """Row boundaries b[0..P] that balance the WORK of the shards: b[g] = first row whose start offset + row_weight * row index
    reaches g / P of the total.  A row costs what its entries cost plus a fixed part -- its offset and its element of y: 12 B
    against 12 B per entry for fp64 values with int32 indices, 8 against 8 for fp32 -- so the default weighs a row like one
    entry: measured on the eight shards of R-MAT scale 24 (bench.py --workload spmv_rmat_shards) the local steps take
    9.3 ps per entry + 8.5 ps per row, and the split by entries alone (row_weight = 0) left the last shard, 7.3 M short rows,
    20 % behind the first, 70 k long ones.  `rowptr` is a 1-D integer tensor (host or device) of m+1 offsets."""
    m = rowptr.numel() - 1
    key = rowptr.to(torch.int64).contiguous()
    if row_weight:
        key = key + int(row_weight) * torch.arange(m + 1, dtype=torch.int64, device=rowptr.device)
    total = int(key[-1].item())
    targets = torch.tensor([(g * total) // world_size for g in range(world_size + 1)], dtype=torch.int64,
                           device=rowptr.device)
    b = torch.searchsorted(key, targets, right=False).clamp_(max=m)
    b[0] = 0
    b[-1] = m
    b = torch.cummax(b, 0).values
    return [int(v) for v in b.tolist()]


def partition_rows_even(m, world_size):
    return [(g * m) // world_size for g in range(world_size + 1)]


def shard_csr(values, rowptr, colind, shape, row_begin, row_end):
    """Local shard of a global CSR: rows [row_begin, row_end), rowptr rebased to 0,
    column indices stay global (x is replicated)."""
    p0 = int(rowptr[row_begin].item())
    p1 = int(rowptr[row_end].item())
    local_rowptr = (rowptr[row_begin:row_end + 1] - p0).to(rowptr.dtype).contiguous()
    return api.csr_view(values[p0:p1].contiguous(), local_rowptr, colind[p0:p1].contiguous(),
                        (row_end - row_begin, shape[1]), p1 - p0)


def _hip_local_spmv(info, a_local, x, y_local):
    api.multiply(info, a_local, x, y_local)


def _order_before_collective(group, tensor):
    """RCCL collectives are ordered behind the kernels already queued on the current stream.  The gloo backend -- only
    ever combined with device tensors by the one-GPU debug mode of bench.py and by tests -- reads device memory from the
    host side: drain the device first, or it may ship a y the kernels have not finished writing."""
    if tensor.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo":
        torch.cuda.synchronize()


class ShardedSpMV:
    """y = A x with A row-sharded over the ranks of `group`.

    a_local   csr_view of this rank's rows (rebased rowptr, global columns)
    bounds    row boundaries b[0..P] shared by all ranks
    local_spmv(info, a_local, x, y_local): defaults to the HIP path; CPU (gloo) tests
              inject the oracle here to exercise the sharding/gather logic without a GPU.
    gather    how unequal (nnz-balanced) shards are exchanged:
              "p2p"     every rank sends its shard straight into the other ranks' y (one group of P-1 sends and
                        P-1 receives, torch.distributed.batch_isend_irecv = one ncclGroup): each shard crosses each
                        xGMI link once and nothing is padded -- the direct all-gather of SURVEY.md section 8e
              "padded"  one all_gather_into_tensor of max-shard-sized slots, then P contiguous copies (R-MAT shards
                        by nnz prefix differ several-fold in rows, so the padding can exceed the payload)
              "auto"    = "p2p".  Equal shards always use ONE in-place all_gather_into_tensor.
    """

    def __init__(self, a_local, bounds, group=None, local_spmv=None, inspect=True, gather="auto", alg=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        assert len(bounds) == self.world + 1
        self.bounds = list(bounds)
        self.m = bounds[-1]
        self.a_local = a_local
        self.local_spmv = local_spmv or _hip_local_spmv
        counts = [bounds[g + 1] - bounds[g] for g in range(self.world)]
        assert a_local.shape()[0] == counts[self.rank]
        self.counts = counts
        self.equal = len(set(counts)) == 1
        assert gather in ("auto", "p2p", "padded")
        self.gather_mode = "inplace" if self.equal else ("p2p" if gather == "auto" else gather)
        vals = a_local.values()
        self.y_full = torch.zeros(self.m, dtype=vals.dtype, device=vals.device)
        self.pad = None
        if self.gather_mode in ("inplace", "p2p"):
            self.y_local = self.y_full[bounds[self.rank]:bounds[self.rank + 1]]  # computed in place
        else:
            self.maxc = max(counts)
            self.pad = torch.zeros(self.world * self.maxc, dtype=vals.dtype, device=vals.device)
            self.y_local = torch.zeros(self.maxc, dtype=vals.dtype, device=vals.device)
        self.info = api.operation_info_t()
        if inspect and local_spmv is None:
            x_probe = torch.empty(a_local.shape()[1], dtype=vals.dtype, device=vals.device)
            kw = {} if alg is None else {"alg": alg}
            # the operator owns its shard for its lifetime: matrix_opt lets inspect keep a re-tiled copy
            self.info = api.multiply_inspect(api.matrix_opt(a_local), x_probe, self.y_local[:counts[self.rank]], **kw)
        self.infos = [self.info]

    def local(self, x):
        self.local_spmv(self.info, self.a_local, x, self.y_local[:self.counts[self.rank]])

    def gather(self):
        if self.world == 1:
            if self.gather_mode == "padded":
                self.y_full.copy_(self.y_local[:self.m])
            return self.y_full
        _order_before_collective(self.group, self.y_full)
        if self.gather_mode == "inplace":
            dist.all_gather_into_tensor(self.y_full, self.y_local, group=self.group)
        elif self.gather_mode == "p2p":
            ops = []
            for g in range(self.world):
                if g == self.rank:
                    continue
                peer = g if self.group is None else dist.get_global_rank(self.group, g)
                if self.counts[self.rank]:
                    ops.append(dist.P2POp(dist.isend, self.y_local, peer, self.group))
                if self.counts[g]:
                    ops.append(dist.P2POp(dist.irecv, self.y_full[self.bounds[g]:self.bounds[g + 1]], peer, self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        else:
            # one padded all-gather, then P contiguous copies
            dist.all_gather_into_tensor(self.pad, self.y_local, group=self.group)
            for g in range(self.world):
                c = self.counts[g]
                if c:
                    self.y_full[self.bounds[g]:self.bounds[g + 1]].copy_(self.pad[g * self.maxc:g * self.maxc + c])
        return self.y_full

    def step(self, x, events=None):
        """One sharded SpMV: local rows, then all-gather(y).  Returns the full y."""
        if events is not None:
            events[0][0].record()
        self.local(x)
        if events is not None:
            events[0][1].record()
        return self.gather()


def striped_row_ranges(m, world_size, chunks):
    """Row ownership for the pipelined variant: the m rows are cut into `chunks` stripes and every
    stripe is split evenly over the ranks, so rank r owns `chunks` contiguous ranges and the
    all-gather of stripe c fills the CONTIGUOUS slice y[stripe c] in place.
    Returns ranges[c][r] = (lo, hi), or None when m is not divisible by chunks * world_size."""
    if chunks < 1 or m % (chunks * world_size) != 0:
        return None
    per = m // (chunks * world_size)
    return [[((c * world_size + r) * per, (c * world_size + r + 1) * per) for r in range(world_size)]
            for c in range(chunks)]


class PipelinedShardedSpMV:
    """Row-sharded SpMV whose all-gather is overlapped with compute.

    xGMI is point-to-point: at P ranks every rank pushes its shard over one link per peer, so the
    gather of a 40 MB y costs 20 MB / 153 GB/s = 130 us at P = 2 -- as long as the SpMV itself.
    The step is therefore cut into `chunks` stripes: stripe c's local rows are computed on the
    compute stream, then its all-gather is issued asynchronously (RCCL's own stream, ordered after
    the stripe's kernels) while stripe c+1 computes.  Still exactly one collective per stripe and
    no other data-path communication.  Each stripe has its own inspect plan; the slice-split
    reduce (spmv_sliced.hip) keeps the chip full on the small stripes."""

    def __init__(self, a_chunks, ranges, group=None, local_spmv=None, inspect=True, alg=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.ranges = ranges
        self.chunks = len(ranges)
        assert len(a_chunks) == self.chunks and all(len(r) == self.world for r in ranges)
        self.a_chunks = a_chunks
        self.m = ranges[-1][-1][1]
        self.local_spmv = local_spmv or _hip_local_spmv
        vals = a_chunks[0].values()
        self.y_full = torch.zeros(self.m, dtype=vals.dtype, device=vals.device)
        self.y_local, self.y_stripe, self.infos = [], [], []
        for c in range(self.chunks):
            lo, hi = ranges[c][self.rank]
            assert a_chunks[c].shape()[0] == hi - lo
            self.y_local.append(self.y_full[lo:hi])
            self.y_stripe.append(self.y_full[ranges[c][0][0]:ranges[c][-1][1]])
            info = api.operation_info_t()
            if inspect and local_spmv is None:
                x_probe = torch.empty(a_chunks[c].shape()[1], dtype=vals.dtype, device=vals.device)
                kw = {} if alg is None else {"alg": alg}
                info = api.multiply_inspect(api.matrix_opt(a_chunks[c]), x_probe, self.y_local[c], **kw)
            self.infos.append(info)

        self._bound_x, self._bound = None, None

    def _bind(self, x):
        """Bind the per-stripe SpMV calls once per x tensor (api.prepared_multiply): the Python
        host layer is then a single ctypes call per stripe."""
        if self.local_spmv is _hip_local_spmv and self._bound_x is not x:
            self._bound = [api.prepared_multiply(self.infos[c], self.a_chunks[c], x, self.y_local[c])
                           for c in range(self.chunks)]
            self._bound_x = x
        return self._bound if self._bound_x is x else None

    def local(self, x):
        """The local SpMVs of every stripe, no collective (diagnostics)."""
        bound = self._bind(x)
        for c in range(self.chunks):
            if bound is not None:
                bound[c]()
            else:
                self.local_spmv(self.infos[c], self.a_chunks[c], x, self.y_local[c])

    def gather(self):
        """The all-gathers of every stripe alone (diagnostics)."""
        if self.world > 1:
            _order_before_collective(self.group, self.y_full)
            for c in range(self.chunks):
                dist.all_gather_into_tensor(self.y_stripe[c], self.y_local[c], group=self.group)
        return self.y_full

    def step(self, x, events=None):
        works = []
        bound = self._bind(x)
        for c in range(self.chunks):
            if events is not None:
                events[c][0].record()
            if bound is not None:
                bound[c]()
            else:
                self.local_spmv(self.infos[c], self.a_chunks[c], x, self.y_local[c])
            if events is not None:
                events[c][1].record()
            if self.world > 1:
                _order_before_collective(self.group, self.y_full)
                works.append(dist.all_gather_into_tensor(self.y_stripe[c], self.y_local[c], group=self.group,
                                                         async_op=True))
        for w in works:
            w.wait()
        return self.y_full


class OverlappedShardedSpMV:
    """Row-sharded SpMV with ONE local plan whose reduce stage is overlapped with the all-gather.

    Ownership is striped (striped_row_ranges): rank r's local matrix holds its `chunks` stripes back
    to back (local rows [c*L, (c+1)*L) = global rows ranges[c][r]).  A step is
        expand(x)                       every product once, x read once          (compute stream)
        for c: reduce_rows(c*L,(c+1)*L)  finishes stripe c of y                   (compute stream)
               all_gather(stripe c)      async on RCCL's stream, overlaps reduce of stripe c+1
    The plan must be SLICED with row-bins aligned to L (handle option BIN_ROW_ALIGN); the
    constructor raises otherwise and callers fall back to ShardedSpMV.  `stages(x)` may be injected
    (CPU gloo tests): it returns (expand, reduce) with reduce(c, y_stripe_local)."""

    def __init__(self, a_local, ranges, group=None, stages=None, alg=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.ranges, self.chunks = ranges, len(ranges)
        self.L = ranges[0][0][1] - ranges[0][0][0]
        assert a_local.shape()[0] == self.L * self.chunks
        self.m = ranges[-1][-1][1]
        self.a_local = a_local
        vals = a_local.values()
        self.y_full = torch.zeros(self.m, dtype=vals.dtype, device=vals.device)
        self.y_local = [self.y_full[ranges[c][self.rank][0]:ranges[c][self.rank][1]] for c in range(self.chunks)]
        self.y_stripe = [self.y_full[ranges[c][0][0]:ranges[c][-1][1]] for c in range(self.chunks)]
        self._stages, self._bound_x, self._bound = stages, None, None
        self.info = api.operation_info_t()
        if stages is None:
            hd = api._Handle.current(vals.device)
            hd.set_option(api._capi.OPT_BIN_ROW_ALIGN, self.L)
            try:
                x_probe = torch.empty(a_local.shape()[1], dtype=vals.dtype, device=vals.device)
                kw = {} if alg is None else {"alg": alg}
                self.info = api.multiply_inspect(api.matrix_opt(a_local), x_probe, self.y_full[:a_local.shape()[0]],
                                                 **kw)
            finally:
                hd.set_option(api._capi.OPT_BIN_ROW_ALIGN, 0)
            pi = self.info.state_.info()
            if pi["alg"] != api._capi.SPMV_SLICED or not pi["bin_aligned"]:
                raise RuntimeError("OverlappedShardedSpMV needs a SLICED plan with stripe-aligned row bins")

    def _bind(self, x):
        if self._bound_x is not x:
            if self._stages is not None:
                expand, reduce = self._stages(x)
                self._bound = (expand, [lambda c=c: reduce(c, self.y_local[c]) for c in range(self.chunks)])
            else:
                plan, item = self.info.state_, self.y_full.element_size()
                expand, reducers = None, []
                for c in range(self.chunks):
                    # base such that local row r of stripe c lands at y_full[global_lo + (r - c*L)]
                    base = self.y_local[c].data_ptr() - c * self.L * item
                    e, red = plan.bind_stages(x, base, self.y_full.dtype)
                    expand = expand or e
                    reducers.append(lambda c=c, red=red: red(c * self.L, (c + 1) * self.L))
                self._bound = (expand, reducers)
            self._bound_x = x
        return self._bound

    def step(self, x, events=None):
        expand, reducers = self._bind(x)
        if events is not None:
            events[0][0].record()
        expand()
        works = []
        for c in range(self.chunks):
            reducers[c]()
            if events is not None and c == self.chunks - 1:
                events[0][1].record()
            if self.world > 1:
                works.append(dist.all_gather_into_tensor(self.y_stripe[c], self.y_local[c], group=self.group,
                                                         async_op=True))
        for w in works:
            w.wait()
        return self.y_full


# --------------------------------------------------------------------------- fused all-gather (IPC + P2P stores)
class _IpcBuffer:
    """Device buffer from spblas_gfx950_ipc_alloc (plain hipMalloc: exportable with hipIpcGetMemHandle),
    visible to torch through __cuda_array_interface__ without a copy."""

    def __init__(self, nbytes, uncached=False):
        import ctypes
        self._ct = ctypes
        p = ctypes.c_void_p()
        api.check(api._capi.lib().spblas_gfx950_ipc_alloc(max(int(nbytes), 1), int(uncached), ctypes.byref(p)),
                  "ipc_alloc")
        self.ptr, self.nbytes = p.value, int(nbytes)

    def tensor(self, dtype, numel):
        typestr = {torch.float32: "<f4", torch.float64: "<f8", torch.int64: "<i8", torch.int32: "<i4"}[dtype]
        owner = self

        class _View:
            __cuda_array_interface__ = {"shape": (numel,), "typestr": typestr, "data": (self.ptr, False), "version": 2}
            _keep = owner
        return torch.as_tensor(_View(), device="cuda")

    def handle(self):
        buf = (self._ct.c_ubyte * 64)()
        api.check(api._capi.lib().spblas_gfx950_ipc_export(self._ct.c_void_p(self.ptr), buf), "ipc_export")
        return bytes(buf)

    def free(self):
        if self.ptr:
            api._capi.lib().spblas_gfx950_ipc_free(self._ct.c_void_p(self.ptr))
            self.ptr = 0


def _ipc_open(handle_bytes):
    import ctypes
    buf = (ctypes.c_ubyte * 64).from_buffer_copy(handle_bytes)
    p = ctypes.c_void_p()
    api.check(api._capi.lib().spblas_gfx950_ipc_open(buf, ctypes.byref(p)), "ipc_open")
    return p.value


class FusedShardedSpMV:
    """Row-sharded SpMV whose all-gather is FUSED into the reduce kernels (second stage).

    Every rank holds two copies of the full y (double buffer) allocated with ipc_alloc and maps the
    copies of all other ranks (hipIpc).  A step is
        expand(x)                                   products of the local rows
        reduce_rows_bcast(...)                       finished rows are stored into ALL P copies of y by
                                                     the reduce / combine kernels themselves (peer stores
                                                     over xGMI: each shard crosses each link once, like a
                                                     direct all-gather, but with no collective launch and
                                                     spread over the duration of the kernels)
        step_signal / step_wait                      device-side barrier: y is complete on this rank once
                                                     all P ranks have signalled the step
    all on the compute stream; the host never synchronises inside a step.  y of step k lives in buffer
    k & 1, so a fast rank writing step k+1 cannot disturb a slow rank still reading step k (as the next
    x, say); buffer k & 1 is reused by step k+2 only after every rank signalled step k+1, i.e. after it
    finished consuming step k.  Requires equal row shards and a SLICED local plan; raises otherwise so
    that callers fall back to ShardedSpMV (RCCL all-gather)."""

    def __init__(self, a_local, bounds, group=None, alg=None, timeout_ms=20000, info=None, stripes=1, chunks=0,
                 shared_device=False):
        """info: an operation_info_t from multiply_inspect on the SAME a_local may be passed to reuse its plan
        (no second inspect; results bit-identical to the path that plan also serves).
        chunks > 0 prepares step_dependent(): the dependent iteration whose peer rows arrive chunk by chunk behind the next
        expand (spblas_gfx950_spmv_step_bcast_chunked); shared_device: several ranks run on ONE device (tests, bench.py
        --debug-one-gpu) -- their waiting expands are kept to a fraction of the device so that they cannot starve each other."""
        import ctypes
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("FusedShardedSpMV needs an initialised process group")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._ct = ctypes
        self._bufs, self._opened, self.y, self.flags = [], [], [], None
        vals = a_local.values()
        self.dtype, self.device = vals.dtype, vals.device
        # Every phase that can fail locally is followed by an agreement (all-reduce of a flag), so that a
        # rank that cannot take part never leaves the others waiting inside a collective.
        err = None
        try:
            counts = {bounds[g + 1] - bounds[g] for g in range(self.world)}
            if len(counts) != 1:
                raise RuntimeError("FusedShardedSpMV needs equal row shards")
            self.bounds, self.m, self.L = list(bounds), bounds[-1], bounds[1] - bounds[0]
            if a_local.shape()[0] != self.L:
                raise RuntimeError("local matrix does not match the row bounds")
            self.a_local = a_local
            item = vals.element_size()
            # local plan (must be the LDS-sliced one: only its kernels have the broadcast epilogue)
            x_probe = torch.empty(a_local.shape()[1], dtype=self.dtype, device=self.device)
            y_probe = torch.empty(self.L, dtype=self.dtype, device=self.device)
            kw = {} if alg is None else {"alg": alg}
            self.info = info if info is not None else api.multiply_inspect(api.matrix_opt(a_local), x_probe,
                                                                           y_probe, **kw)
            if not isinstance(self.info.state_, api._Plan) or self.info.state_.info()["alg"] != api._capi.SPMV_SLICED:
                raise RuntimeError("FusedShardedSpMV needs a SLICED local plan")
            si_ = self.info.state_.sliced_info()
            if si_.get("refresh_each_call") and not si_.get("value_free"):
                # (a plan made from a plain inspected csr_view must read A's values of every call; the fused entry points are
                # not given them.  A VALUE-FREE plan -- what such an operand gets at this size since round 5 -- holds no copy
                # that could go stale: its reduce reads the caller's array through the pointer registered with the plan, as it
                # is when the step runs (round 6: that kernel has the broadcast epilogue too).  Only the copying form is left out.)
                raise RuntimeError("FusedShardedSpMV needs a plan that owns its values or reads the caller's: inspect "
                                   "matrix_opt(a_local) or a matrix large enough for value-free tiles")
            self.value_free = bool(si_.get("value_free"))
            # buffers: two copies of y, one flag array (slot q = last step signalled by rank q)
            self.chunks = max(0, int(chunks))
            if self.world * max(self.chunks, 1) > 64:
                raise RuntimeError("FusedShardedSpMV: ranks x chunks must not exceed 64")
            self._bufs = [_IpcBuffer(self.m * item), _IpcBuffer(self.m * item), _IpcBuffer(self.world * 8, uncached=True),
                          _IpcBuffer(self.world * max(self.chunks, 1) * 8, uncached=True)]
            self.y = [self._bufs[0].tensor(self.dtype, self.m), self._bufs[1].tensor(self.dtype, self.m)]
            self.flags = self._bufs[2].tensor(torch.int64, self.world)
            self.chunk_flags = self._bufs[3].tensor(torch.int64, self.world * max(self.chunks, 1))
            my_rows = None
            if self.chunks:
                rows = (ctypes.c_int64 * (self.chunks + 1))()
                api.check(api._capi.lib().spblas_gfx950_spmv_chunk_rows(self.info.state_.plan, self.chunks, rows),
                          "spmv_chunk_rows")
                my_rows = [self.bounds[self.rank] + int(r) for r in rows]
            mine = [b.handle() for b in self._bufs] + [my_rows]
        except Exception as e:  # noqa: BLE001 - reported after the agreement
            err, mine = e, None
        self._agree(err, "preparing the local plan and buffers")
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)
        ptrs = [[0] * self.world for _ in range(4)]
        try:
            for q in range(self.world):
                for j in range(4):
                    if q == self.rank:
                        ptrs[j][q] = self._bufs[j].ptr
                    else:
                        ptrs[j][q] = _ipc_open(everyone[q][j])
                        self._opened.append(ptrs[j][q])
            self._tabs = [torch.tensor(ptrs[j], dtype=torch.int64, device=self.device) for j in range(4)]
            self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._chunk_status = torch.zeros(2, dtype=torch.int32, device=self.device)  # [timed out, longest wait in ticks]
            self._chunk_rows = None
            if self.chunks:
                self._chunk_rows = torch.tensor([r for q in range(self.world) for r in everyone[q][4]], dtype=torch.int64,
                                                device=self.device)
            cus = torch.cuda.get_device_properties(self.device).multi_processor_count
            self._max_wgs = max(1, cus // (2 * self.world)) if shared_device else 0
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            err = e
        self._agree(err, "mapping the peers' buffers")  # also: everything is mapped before the first peer store
        self._step, self._timeout, self.stripes = 0, int(timeout_ms), max(1, int(stripes))
        self._alpha = (ctypes.c_float if self.dtype == torch.float32 else ctypes.c_double)(1.0)
        self._plan = self.info.state_
        self._x, self._xp = None, None
        self._pending = False  # a pipelined step whose wait has not been issued yet
        self._chained = False  # the last step was a step_dependent(): its rows arrive behind chunk flags, not a barrier

    def _agree(self, err, what):
        """Collective: raise on every rank if any rank failed."""
        on_gpu = dist.get_backend(self.group) == "nccl"
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=self.device if on_gpu else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()) != 0:
            self._release()
            raise RuntimeError(f"FusedShardedSpMV: a rank failed while {what}" + (f": {err}" if err is not None else ""))

    def step_dependent(self, x=None, alpha=None):
        """One step of the dependent iteration y_{k+1} = alpha * A y_k WITHOUT a step barrier (constructor: chunks > 0).
        x given: the chain starts here (x is complete on every rank: a plain expand).  x None: multiply the y this operator
        produced last -- the expand waits, x slice by x slice, for the chunks of the peers' rows that slice is made of, so
        the link time of step k hides behind the expand of step k + 1; flush_chain() before reading y on the host or
        handing it to anything but the next step_dependent().  Bit-identical to step() on the same vectors."""
        ct, lib = self._ct, api._capi.lib()
        if not self.chunks:
            raise RuntimeError("FusedShardedSpMV: created without chunks")
        if x is not None:
            self.flush()
            self.flush_chain()
        elif not self._step:
            raise RuntimeError("step_dependent(): nothing to continue from")
        elif not self._chained:
            self.flush()  # the last step ended with the barrier: its y is complete here, nothing to wait for
        wait = None
        if x is None:
            x = self.y[self._step & 1]
            if self._chained:
                wait = api._capi.chunk_wait(self.chunk_flags.data_ptr(), self._chunk_rows.data_ptr(), self.world, self.chunks,
                                            self._step, self._timeout, self._chunk_status.data_ptr(), self._max_wgs)
        self._step += 1
        b = self._step & 1
        a = self._alpha if alpha is None else type(self._alpha)(alpha)
        h, plan = api._Handle.current(self.device).h, self._plan.plan
        api.check(lib.spblas_gfx950_spmv_step_bcast_chunked(h, plan, ct.byref(a), ct.c_void_p(x.data_ptr()),
                                                            ct.c_void_p(self._tabs[b].data_ptr()), self.world,
                                                            self.bounds[self.rank], self.chunks,
                                                            ct.c_void_p(self._tabs[3].data_ptr()), self.rank, self._step,
                                                            ct.byref(wait) if wait is not None else None),
                  "spmv_step_bcast_chunked")
        self._chained = True
        self._keep_x = x
        return self.y[b]

    def flush_chain(self):
        """Wait (on the device) until every rank's chunks of the last step_dependent() have arrived in this rank's copy."""
        if self._chained:
            ct, lib = self._ct, api._capi.lib()
            h = api._Handle.current(self.device).h
            api.check(lib.spblas_gfx950_step_wait(h, ct.c_void_p(self.chunk_flags.data_ptr()), self.world * self.chunks,
                                                  self._step, self._timeout, ct.c_void_p(self._chunk_status.data_ptr())),
                      "step_wait")
            self._chained = False
        return self.y[self._step & 1]

    def chunk_wait_us(self, reset=True):
        """Longest time a workgroup of a waiting expand has spent on its chunk flags since the last reset (host sync)."""
        ticks = int(self._chunk_status[1].item())
        if reset:
            self._chunk_status[1] = 0
        khz = self._ct.c_int(0)  # (the device's own rate, as the timeouts use: wall_clock64 ticks at 100 MHz on gfx9)
        api.check(api._capi.lib().spblas_gfx950_wall_clock_khz(api._Handle.current(self.device).h, self._ct.byref(khz)),
                  "wall_clock_khz")
        return ticks * 1e3 / max(1, khz.value)

    def step(self, x, events=None):
        """One sharded SpMV; returns the full y (valid on this rank once the stream reaches this point)."""
        ct, lib = self._ct, api._capi.lib()
        self.flush()  # (a pipelined step before this one: its wait comes first)
        self.flush_chain()
        if self._x is not x:
            self._x, self._xp = x, ct.c_void_p(x.data_ptr())
        self._step += 1
        b = self._step & 1
        h, plan = api._Handle.current(self.device).h, self._plan.plan  # (re)binds torch's current stream
        if events is not None:
            events[0][0].record()
        api.check(lib.spblas_gfx950_spmv_step_bcast(h, plan, ct.byref(self._alpha), self._xp,
                                                    ct.c_void_p(self._tabs[b].data_ptr()), self.world,
                                                    self.bounds[self.rank], self.stripes), "spmv_step_bcast")
        if events is not None:
            events[0][1].record()
        if not self._muted():
            api.check(lib.spblas_gfx950_step_signal(h, ct.c_void_p(self._tabs[2].data_ptr()), self.world, self.rank,
                                                    self._step), "step_signal")
        api.check(lib.spblas_gfx950_step_wait(h, ct.c_void_p(self.flags.data_ptr()), self.world, self._step,
                                              self._timeout, ct.c_void_p(self._status.data_ptr())), "step_wait")
        return self.y[b]

    def _muted(self):
        """Test hook (tests/test_gpu_fused_sharding.py): SPBLAS_GFX950_TEST_MUTE_RANK=r makes rank r stop publishing its step
        flag after step SPBLAS_GFX950_TEST_MUTE_AFTER (default 0: never publishes) -- what a dead link or a lost flag store
        looks like to the peers, whose bounded waits must time out and whose callers must fall back."""
        r = os.environ.get("SPBLAS_GFX950_TEST_MUTE_RANK")
        return r is not None and int(r) == self.rank and self._step > int(os.environ.get("SPBLAS_GFX950_TEST_MUTE_AFTER", "0"))

    def step_pipelined(self, x):
        """Throughput form of step() for INDEPENDENT right-hand sides (back-to-back SpMVs whose x does not depend on the
        previous y): the device-side wait for step k-1 sits right before the kernel of step k that stores to the peers (the
        combine kernel of a K-split reduce, else the reduce), so the peers' stores of step k-1 cross the links while this
        rank already expands and reduces step k -- link time and compute overlap
        ("overlap collectives with compute").  Every step's y is still complete and bit-identical to step()'s; but the
        buffer returned here may only be read after flush() (or after the next call has passed its internal wait and
        before it signals).  A dependent iteration (x_k+1 = f(y_k)) needs step().  Buffer safety: a peer overwrites my
        copy k & 1 in its step k + 2, which it starts after my signal of step k + 1, which I send after my wait of step k."""
        ct, lib = self._ct, api._capi.lib()
        self.flush_chain()
        if self._x is not x:
            self._x, self._xp = x, ct.c_void_p(x.data_ptr())
        self._step += 1
        b = self._step & 1
        h, plan = api._Handle.current(self.device).h, self._plan.plan
        api.check(lib.spblas_gfx950_spmv_expand(h, plan, self._xp), "spmv_expand")
        if self._pending:  # the wait goes right before the kernel that stores to the peers (combine, or the reduce itself)
            api.check(lib.spblas_gfx950_bcast_wait_before(h, ct.c_void_p(self.flags.data_ptr()), self.world,
                                                          self._step - 1, self._timeout,
                                                          ct.c_void_p(self._status.data_ptr())), "bcast_wait_before")
        api.check(lib.spblas_gfx950_spmv_reduce_rows_bcast(h, plan, ct.byref(self._alpha),
                                                           ct.c_void_p(self._tabs[b].data_ptr()), self.world,
                                                           self.bounds[self.rank], 0, self.L), "spmv_reduce_rows_bcast")
        api.check(lib.spblas_gfx950_step_signal(h, ct.c_void_p(self._tabs[2].data_ptr()), self.world, self.rank,
                                                self._step), "step_signal")
        self._pending = True
        return self.y[b]

    def flush(self):
        """Wait (on the device) for the last pipelined step: its y is complete on this rank once the stream gets here."""
        if self._pending:
            ct, lib = self._ct, api._capi.lib()
            h = api._Handle.current(self.device).h
            api.check(lib.spblas_gfx950_step_wait(h, ct.c_void_p(self.flags.data_ptr()), self.world, self._step,
                                                  self._timeout, ct.c_void_p(self._status.data_ptr())), "step_wait")
            self._pending = False
        return self.y[self._step & 1]

    def check_status(self):
        """After a host synchronisation: raise if a step barrier timed out (a peer stopped responding)."""
        if int(self._status.item()) != 0:
            raise RuntimeError("FusedShardedSpMV: step barrier timed out")
        if int(self._chunk_status[0].item()) != 0:
            raise RuntimeError("FusedShardedSpMV: a chunk of a peer's rows did not arrive in time")

    def close(self):
        torch.cuda.synchronize()
        try:
            if dist.is_initialized():
                dist.barrier(group=self.group)  # nobody may still be storing into a buffer about to go away
        except Exception:  # noqa: BLE001
            pass
        self._release()

    def _release(self):
        lib = api._capi.lib()
        for p in self._opened:
            lib.spblas_gfx950_ipc_close(self._ct.c_void_p(p))
        self._opened = []
        self.y, self.flags, self.chunk_flags = [], None, None
        for b in self._bufs:
            b.free()
        self._bufs = []


def try_fused(a_local, bounds, x, reference_step, alg=None, group=None, log=None, info=None, stripes=1, chunks=0,
              shared_device=False):
    """Collective.  Returns a FusedShardedSpMV if EVERY rank could set it up and its full y agrees with
    `reference_step(x_k)` (the RCCL all-gather path) on every rank for FOUR different vectors x_k; otherwise
    None, with everything the attempt allocated released again.  Never raises: any failure means "keep the
    reference path".  The collective sequence is identical on every rank (constructor agreements, the reference
    steps, one final all-reduce), whichever rank fails where.

    The vectors differ from step to step (x, 2x, x + 1, 0.5x - 1) so that a stale or missing peer store cannot
    hide: y of step k lives in buffer k & 1, so both copies are written twice with different contents and a
    row that a peer failed to deliver -- or delivered late, after the step barrier -- still holds the previous
    step's (different) value and the comparison fails.  With the same x on every step both buffers would hold
    identical bits whatever the peers did.

    With `info` (the reference path's own plan) the two paths run the same kernels on the same plan and
    must agree bit for bit; without it each path has its own plan -- inspect orders the entries of a
    run by arrival, so two plans of one matrix differ in summation order -- and agreement is checked to
    1e-4 / 1e-10 relative, which still catches any wiring error (wrong offset, missing rows, stale
    buffer).  reference_step takes the vector to multiply (callables without a parameter are called as
    before and validated with the single x they close over)."""
    import inspect as _inspect
    fused, same = None, 0
    try:
        fused = FusedShardedSpMV(a_local, bounds, group=group, alg=alg, info=info, timeout_ms=3000, stripes=stripes,
                                 chunks=chunks, shared_device=shared_device)
        takes_x = len(_inspect.signature(reference_step).parameters) >= 1
        xs = [x, 2.0 * x, x + 1.0, 0.5 * x - 1.0] if takes_x else [x, x, x]
        # Every rank issues the SAME collective sequence whatever happens locally: all reference steps (RCCL
        # all-gathers) first, then the fused steps, which contain no collective -- a rank whose barrier times out
        # (the missing-peer-store case this validation exists to catch) must not skip an all-gather its peers are
        # still inside.  Local failures only clear `same`; the outcome is agreed in the all-reduce below.
        y_refs = [(reference_step(x_k) if takes_x else reference_step()).clone() for x_k in xs]
        same = 1
        for x_k, y_ref in zip(xs, y_refs):
            try:
                y_fused = fused.step(x_k)
                torch.cuda.synchronize()
                fused.check_status()
                if info is not None:
                    ok = torch.equal(y_ref, y_fused)
                else:
                    tol = 1e-4 if y_ref.dtype == torch.float32 else 1e-10
                    ok = torch.allclose(y_fused, y_ref, rtol=tol, atol=tol * float(y_ref.abs().max()))
            except Exception as e:  # noqa: BLE001 - keep stepping so the peers' barriers are not left short
                ok = False
                if log:
                    log(f"fused step failed: {e}")
            same &= int(ok)
        fused._timeout = 20000
        if not same and log:
            log("fused all-gather disagrees with the reference path")
    except Exception as e:  # noqa: BLE001
        if log:
            log(f"fused all-gather unavailable: {e}")
        same = 0
        if fused is not None and not fused._bufs:
            fused = None  # the constructor already released everything
    on_gpu = dist.get_backend(group) == "nccl"
    flag = torch.tensor([same], dtype=torch.int32, device=x.device if on_gpu else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 1:
        return fused
    if fused is not None:
        fused.close()
    return None


# --------------------------------------------------------------------------- SpMM and SpGEMM over row shards
def _hip_local_spmm(info, a_local, b, c_local):
    api.multiply(info, a_local, b, c_local)


class ShardedSpMM:
    """C = A B with A row-sharded, B (k x n, row-major) replicated, C row-sharded (SURVEY.md section 8e,
    "Partitioning": rows of multiply_impl.hpp:85-91 are independent).  No collective on the data path: every rank
    owns the rows [bounds[r], bounds[r+1]) of C.  gather_c() assembles the full C on every rank when a caller needs it
    (one all-gather of the row blocks, in place for equal shards, direct sends for nnz-balanced ones)."""

    def __init__(self, a_local, bounds, ncols, group=None, local_spmm=None, inspect=True):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        assert len(bounds) == self.world + 1 and a_local.shape()[0] == bounds[self.rank + 1] - bounds[self.rank]
        self.bounds, self.m, self.ncols = list(bounds), bounds[-1], int(ncols)
        self.a_local = a_local
        self.local_spmm = local_spmm or _hip_local_spmm
        vals = a_local.values()
        self.c_full = torch.zeros((self.m, self.ncols), dtype=vals.dtype, device=vals.device)
        self.c_local = self.c_full[bounds[self.rank]:bounds[self.rank + 1]]
        self.info = api.operation_info_t()
        if inspect and local_spmm is None:
            b_probe = torch.empty((a_local.shape()[1], self.ncols), dtype=vals.dtype, device=vals.device)
            self.info = api.multiply_inspect(api.matrix_opt(a_local), b_probe, self.c_local)

    def local(self, b):
        """This rank's rows of C (a view into its full-size buffer)."""
        self.local_spmm(self.info, self.a_local, b, self.c_local)
        return self.c_local

    def gather_c(self):
        if self.world == 1:
            return self.c_full
        counts = [self.bounds[g + 1] - self.bounds[g] for g in range(self.world)]
        if len(set(counts)) == 1:
            dist.all_gather_into_tensor(self.c_full, self.c_local, group=self.group)
            return self.c_full
        ops = []
        for g in range(self.world):
            if g == self.rank:
                continue
            peer = g if self.group is None else dist.get_global_rank(self.group, g)
            if counts[self.rank]:
                ops.append(dist.P2POp(dist.isend, self.c_local, peer, self.group))
            if counts[g]:
                ops.append(dist.P2POp(dist.irecv, self.c_full[self.bounds[g]:self.bounds[g + 1]], peer, self.group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return self.c_full


def _hip_local_spgemm(a_local, b):
    """(rowptr, colind, values) of a_local * b through multiply_compute / multiply_fill."""
    m, n = a_local.shape()[0], b.shape()[1]
    dev = a_local.values().device
    rp = torch.zeros(m + 1, dtype=torch.int32, device=dev)
    c = api.csr_view(None, rp, None, (m, n), 0)
    info = api.multiply_compute(a_local, b, c)
    nnz = info.result_nnz()
    c.update(torch.empty(nnz, dtype=a_local.values().dtype, device=dev), rp,
             torch.empty(nnz, dtype=torch.int32, device=dev), (m, n), nnz)
    api.multiply_fill(info, a_local, b, c)
    return c.rowptr(), c.colind(), c.values()


class ShardedSpGEMM:
    """C = A B with A row-sharded and B replicated: every rank computes its block of rows of C with the single-GPU
    SpGEMM; C's row blocks are disjoint, so the only exchange is the block sizes -- nnz offsets of the global C are an
    exclusive scan over the ranks.  compute() returns this rank's block as
    (rowptr rebased to 0, colind, values) together with (nnz_offset, nnz_total) of the global matrix."""

    def __init__(self, a_local, b, bounds, group=None, local_spgemm=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        assert len(bounds) == self.world + 1 and a_local.shape()[0] == bounds[self.rank + 1] - bounds[self.rank]
        self.a_local, self.b, self.bounds = a_local, b, list(bounds)
        self.local_spgemm = local_spgemm or _hip_local_spgemm

    def compute(self):
        rowptr, colind, values = self.local_spgemm(self.a_local, self.b)
        nnz_local = int(rowptr[-1].item()) if rowptr.numel() else 0
        sizes = [nnz_local]
        if self.world > 1:
            on_gpu = dist.get_backend(self.group) == "nccl"
            mine = torch.tensor([nnz_local], dtype=torch.int64, device=values.device if on_gpu else "cpu")
            every = torch.zeros(self.world, dtype=torch.int64, device=mine.device)
            dist.all_gather_into_tensor(every, mine, group=self.group)
            sizes = [int(v) for v in every.tolist()]
        offset = sum(sizes[:self.rank])
        return (rowptr, colind, values), (offset, sum(sizes))
