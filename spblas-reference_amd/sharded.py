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
import torch
import torch.distributed as dist

from . import api


def partition_rows_by_nnz(rowptr, world_size):
    """Row boundaries b[0..P] with b[g] = first row whose start offset >= g*nnz/P.
    `rowptr` is a 1-D integer tensor (host or device) of m+1 offsets."""
    m = rowptr.numel() - 1
    nnz = int(rowptr[-1].item())
    targets = torch.tensor([(g * nnz) // world_size for g in range(world_size + 1)], dtype=rowptr.dtype,
                           device=rowptr.device)
    b = torch.searchsorted(rowptr.contiguous(), targets, right=False).clamp_(max=m)
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


class ShardedSpMV:
    """y = A x with A row-sharded over the ranks of `group`.

    a_local   csr_view of this rank's rows (rebased rowptr, global columns)
    bounds    row boundaries b[0..P] shared by all ranks
    local_spmv(info, a_local, x, y_local): defaults to the HIP path; CPU (gloo) tests
              inject the oracle here to exercise the sharding/gather logic without a GPU.
    """

    def __init__(self, a_local, bounds, group=None, local_spmv=None, inspect=True):
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
        vals = a_local.values()
        self.y_full = torch.zeros(self.m, dtype=vals.dtype, device=vals.device)
        if self.equal:
            self.y_local = self.y_full[bounds[self.rank]:bounds[self.rank + 1]]  # in-place gather
            self.pad = None
        else:
            self.maxc = max(counts)
            self.pad = torch.zeros(self.world * self.maxc, dtype=vals.dtype, device=vals.device)
            self.y_local = torch.zeros(self.maxc, dtype=vals.dtype, device=vals.device)
        self.info = api.operation_info_t()
        if inspect and local_spmv is None:
            x_probe = torch.empty(a_local.shape()[1], dtype=vals.dtype, device=vals.device)
            self.info = api.multiply_inspect(a_local, x_probe, self.y_local[:counts[self.rank]])

    def local(self, x):
        self.local_spmv(self.info, self.a_local, x, self.y_local[:self.counts[self.rank]])

    def gather(self):
        if self.world == 1:
            if not self.equal:
                self.y_full.copy_(self.y_local[:self.m])
            return self.y_full
        if self.equal:
            dist.all_gather_into_tensor(self.y_full, self.y_local, group=self.group)
        else:
            # unequal (nnz-balanced) shards: one padded all-gather, then P contiguous copies
            dist.all_gather_into_tensor(self.pad, self.y_local, group=self.group)
            for g in range(self.world):
                c = self.counts[g]
                if c:
                    self.y_full[self.bounds[g]:self.bounds[g + 1]].copy_(self.pad[g * self.maxc:g * self.maxc + c])
        return self.y_full

    def step(self, x):
        """One sharded SpMV: local rows, then all-gather(y).  Returns the full y."""
        self.local(x)
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
                info = api.multiply_inspect(a_chunks[c], x_probe, self.y_local[c], **kw)
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
                self.info = api.multiply_inspect(a_local, x_probe, self.y_full[:a_local.shape()[0]], **kw)
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
