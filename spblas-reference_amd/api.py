"""Host-side mirror of the spblas-reference operator interface for the multiply() path.

Same names, argument meaning and error behaviour as the reference (paths relative to
/root/reference/include/spblas/):

  csr_view / csc_view          views/csr_view.hpp:12-77, views/csc_view.hpp
  scaled / scaled_view         algorithms/scaled.hpp, views/scaled_view_impl.hpp:97-219
  conjugated                   views/conjugated_view_impl.hpp (rejected by GPU backends,
                               vendor/rocsparse/detail/spmv_impl.hpp:29-33)
  transposed                   algorithms/transposed.hpp:7-21 (zero-copy CSR<->CSC relabel)
  matrix_opt                   views/matrix_opt_impl.hpp:14-93 (caches the inspect result)
  operation_info_t             detail/operation_info_t.hpp:28-104
  spgemm_state_t               vendor/rocsparse/multiply_spgemm.hpp:28-230
  multiply, multiply_inspect   algorithms/multiply.hpp, vendor/rocsparse/detail/spmv_impl.hpp:18-90,
                               vendor/onemkl_sycl/spmm_impl.hpp:40-198
  multiply_compute/_fill, multiply_symbolic_compute/_fill, multiply_numeric
                               algorithms/multiply.hpp:48-55, vendor/rocsparse/multiply_spgemm.hpp:232-317

The reference is C++; its toolchain dependencies (range-v3, mdspan) are absent from
this image, so the C++ header layer (include/spblas/vendor/gfx950/) cannot be
exercised here and this Python layer is what the tests drive.  Device memory is
held in torch CUDA tensors (plumbing only): a view wraps caller-owned tensors exactly
as csr_view wraps caller-owned device pointers.  All compute goes through the C ABI
(include/spblas_gfx950.h); nothing here computes on the CPU or through torch ops.
"""
import ctypes
import threading

import torch

from . import _capi
from ._capi import check


# --------------------------------------------------------------------------- views
class index(tuple):
    """spblas::index<I> (detail/index.hpp:14-55): a (rows, cols) pair."""

    def __new__(cls, a, b=None):
        if b is None:
            a, b = a
        return super().__new__(cls, (int(a), int(b)))


def _is_tensor(t):
    return isinstance(t, torch.Tensor)


class view_base:
    pass


class csr_view(view_base):
    """Non-owning CSR view over caller-owned device arrays (views/csr_view.hpp:20-26)."""

    def __init__(self, values, rowptr, colind, shape, nnz):
        self._values, self._rowptr, self._colind = values, rowptr, colind
        self._shape, self._nnz = index(shape), int(nnz)

    def update(self, values, rowptr, colind, shape=None, nnz=None):  # csr_view.hpp:36-49
        self._values, self._rowptr, self._colind = values, rowptr, colind
        if shape is not None:
            self._shape, self._nnz = index(shape), int(nnz)

    def values(self):
        return self._values

    def rowptr(self):
        return self._rowptr

    def colind(self):
        return self._colind

    def shape(self):
        return self._shape

    def size(self):
        return self._nnz


class csc_view(view_base):
    """Non-owning CSC view (views/csc_view.hpp)."""

    def __init__(self, values, colptr, rowind, shape, nnz):
        self._values, self._colptr, self._rowind = values, colptr, rowind
        self._shape, self._nnz = index(shape), int(nnz)

    def update(self, values, colptr, rowind, shape=None, nnz=None):  # views/csc_view.hpp
        self._values, self._colptr, self._rowind = values, colptr, rowind
        if shape is not None:
            self._shape, self._nnz = index(shape), int(nnz)

    def values(self):
        return self._values

    def colptr(self):
        return self._colptr

    def rowind(self):
        return self._rowind

    def shape(self):
        return self._shape

    def size(self):
        return self._nnz


class scaled_view(view_base):
    def __init__(self, alpha, base):
        self._alpha, self._base = alpha, base

    def alpha(self):
        return self._alpha

    def base(self):
        return self._base

    def shape(self):
        return _shape_of(self._base)


class conjugated_view(view_base):
    def __init__(self, base):
        self._base = base

    def base(self):
        return self._base

    def shape(self):
        return _shape_of(self._base)


class matrix_opt(view_base):
    """matrix_opt<M> (views/matrix_opt_impl.hpp): owns the cached vendor handle -- here
    the SpMV/SpMM plan built by multiply_inspect."""

    def __init__(self, matrix):
        if not isinstance(matrix, (csr_view, csc_view)):
            raise TypeError("matrix_opt wraps a csr_view or csc_view")
        self.matrix_ = matrix
        self._plan = None

    def base(self):
        return self.matrix_

    def shape(self):
        return self.matrix_.shape()

    def size(self):
        return self.matrix_.size()


def scaled(alpha, t):
    return scaled_view(alpha, t)


def conjugated(t):
    return conjugated_view(t)


def transposed(a):
    """algorithms/transposed.hpp:7-21: re-label CSR<->CSC with zero copy."""
    if isinstance(a, csr_view):
        return csc_view(a.values(), a.rowptr(), a.colind(), (a.shape()[1], a.shape()[0]), a.size())
    if isinstance(a, csc_view):
        return csr_view(a.values(), a.colptr(), a.rowind(), (a.shape()[1], a.shape()[0]), a.size())
    raise TypeError("transposed() expects a csr_view or csc_view")


# ----------------------------------------------------------- detail/view_inspectors.hpp
def get_ultimate_base(t):  # view_inspectors.hpp:104-111
    while isinstance(t, (scaled_view, conjugated_view, matrix_opt)):
        t = t.base()
    return t


def get_scaling_factor(*ts):  # view_inspectors.hpp:22-77: product of all factors, or None
    out = None
    for t in ts:
        while isinstance(t, (scaled_view, conjugated_view, matrix_opt)):
            if isinstance(t, scaled_view):
                out = t.alpha() if out is None else out * t.alpha()
            t = t.base()
    return out


def is_conjugated(t):  # view_inspectors.hpp:81-97: odd number of conjugated_views
    c = False
    while isinstance(t, (scaled_view, conjugated_view, matrix_opt)):
        if isinstance(t, conjugated_view):
            c = not c
        t = t.base()
    return c


def has_matrix_opt(t):  # view_inspectors.hpp:113-122
    while isinstance(t, (scaled_view, conjugated_view, matrix_opt)):
        if isinstance(t, matrix_opt):
            return True
        t = t.base()
    return False


def _get_matrix_opt(t):
    while isinstance(t, (scaled_view, conjugated_view, matrix_opt)):
        if isinstance(t, matrix_opt):
            return t
        t = t.base()
    return None


def _shape_of(t):
    if _is_tensor(t):
        return t.shape[0] if t.dim() == 1 else index(t.shape[0], t.shape[1])
    return t.shape()


# --------------------------------------------------------------------------- state
class _Handle:
    """One backend handle per (host thread, device).  A handle carries the stream the next call launches on and a
    scratch buffer, so -- like the vendor handles it stands in for (vendor/rocsparse/detail/operation_state_t.hpp keeps
    one per operation state) -- it must not be shared by threads that call concurrently: ctypes drops the GIL for the
    duration of a call, and a second thread re-binding the stream between this thread's set_stream and its launch
    would put the launch on the wrong stream."""
    _tls = threading.local()

    def __init__(self, device):
        self.device = device
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            check(_capi.lib().spblas_gfx950_create(ctypes.byref(h), None), "spblas_gfx950_create")
        self.h = h

    def __del__(self):  # a thread's handles go with the thread (plans and states keep theirs alive)
        try:
            if self.h:
                _capi.lib().spblas_gfx950_destroy(self.h)
                self.h = None
        except Exception:
            pass  # interpreter shutdown

    def set_option(self, option, value):
        check(_capi.lib().spblas_gfx950_set_option(self.h, option, value), "spblas_gfx950_set_option")

    @classmethod
    def current(cls, device):
        if device.type != "cuda":
            raise RuntimeError("gfx950 backend: arrays must live in device (HIP) memory; there is no CPU fallback")
        key = device.index if device.index is not None else torch.cuda.current_device()
        by_device = getattr(cls._tls, "by_device", None)
        if by_device is None:
            by_device = cls._tls.by_device = {}
        hd = by_device.get(key)
        if hd is None:
            hd = by_device[key] = cls(key)
        stream = torch.cuda.current_stream(key).cuda_stream
        check(_capi.lib().spblas_gfx950_set_stream(hd.h, ctypes.c_void_p(stream)), "spblas_gfx950_set_stream")
        return hd


# In-place changes to an array made through this module's own raw kernels (invisible to torch's version
# counter) are recorded here: data_ptr -> number of such writes.  Every API entry that writes a value or column
# array through the C ABI -- scale(), multiply_fill / multiply_numeric, add, transpose -- bumps the epoch, so a
# SLICED plan's value snapshot and the "colind provably untouched" test of repeated SpGEMM fills see them.
_value_epochs = {}


def _note_write(*tensors):
    for t in tensors:
        if t is not None:
            _value_epochs[t.data_ptr()] = _value_epochs.get(t.data_ptr(), 0) + 1


def _values_stamp(values):
    """(data_ptr, torch version counter, scale() epoch): changes whenever the array is rebound or modified in
    place through torch or through this module."""
    return values.data_ptr(), values._version, _value_epochs.get(values.data_ptr(), 0)


class _Plan:
    """Owner of a spblas_gfx950_plan_t (released with the handle's stream order).

    Holds references to the matrix's tensors, so their addresses -- the plan's identity in the library -- cannot
    be recycled for another matrix while the plan lives.  A SLICED plan multiplies with a re-tiled COPY of the
    values (include/spblas_gfx950.h, "value snapshot contract"); `stamp` remembers the state of the caller's
    value array the copy was taken from, and _spmv refreshes the copy when it sees another one."""

    def __init__(self, handle, plan, key, tensors=None):
        self.handle, self.plan, self.key = handle, plan, key
        self.tensors = tensors
        self.snapshot = self.info()["alg"] == _capi.SPMV_SLICED
        self.stamp = _values_stamp(tensors[2]) if tensors is not None and tensors[2] is not None else None

    def _calling_handle(self, device=None):
        """The CALLING thread's handle, bound to its current stream: a plan made in thread A and used in thread B (an
        autograd backward thread, a worker pool) must launch its refresh / inspect work on B's stream, in order with B's
        multiply -- not on whatever stream A's handle was last bound to.  The creator's handle (self.handle) is kept
        alive for the plan's destruction only."""
        return _Handle.current(device if device is not None else torch.device("cuda", self.handle.device))

    def refresh_if_stale(self, values):
        """SLICED plans only: re-copy the values when the caller's array was rebound or changed in place through
        torch or scale() since the copy was taken.  (Writes by foreign kernels are invisible here: call
        update_values after those, as the C ABI asks.)"""
        if self.snapshot and values is not None and _values_stamp(values) != self.stamp:
            self.update_values(values)

    def info(self):
        arr = (ctypes.c_int64 * 12)()
        check(_capi.lib().spblas_gfx950_plan_info(self.plan, arr), "spblas_gfx950_plan_info")
        names = ["alg", "window", "n_windows", "n_long_rows", "max_row_len", "device_bytes", "n_slices",
                 "empty_rows", "rows_per_bin", "bin_aligned", "expand_items", "reduce_items"]
        return dict(zip(names, list(arr)))

    def sliced_info(self):
        arr = (ctypes.c_int64 * 12)()
        check(_capi.lib().spblas_gfx950_plan_info_sliced(self.plan, arr), "spblas_gfx950_plan_info_sliced")
        names = ["n_bins", "variable_bins", "expand_blocks", "reduce_blocks", "placed_entries", "hub_rows", "hub_len",
                 "ksplit", "tiled_rows", "auto_trial", "trial_rowblock_ns", "trial_sliced_ns"]
        d = dict(zip(names, list(arr)))
        bits = d["auto_trial"]
        d["row_code_u8"] = (bits >> 1) & 1  # one-byte row codes in the reduce's stream (runs sorted by row)
        d["refresh_each_call"] = (bits >> 6) & 1  # made without the snapshot opt-in: every multiply takes A's values again
        d["value_free"] = (bits >> 7) & 1  # ... and holds no copy of them: the reduce reads the caller's array through LDS (round 5)
        d["nt_product_stores"] = (bits >> 2) & 1  # the expand stores its products with the non-temporal hint
        d["store_trial"] = (bits >> 3) & 1  # THIS plan timed both store flavours (opt-in: OPT_STORE_TRIAL = 2 on the handle or SPBLAS_GFX950_PB_NT=-2)
        d["auto_trial"] = bits & 1
        if (bits >> 4) & 1:  # hot-column split (csrc/spmv_hot.hip): the tile numbers above are those of A_rest
            hot = (ctypes.c_int64 * 8)()
            check(_capi.lib().spblas_gfx950_plan_info_hot(self.plan, hot), "spblas_gfx950_plan_info_hot")
            d["hot_split"] = dict(zip(("hot_columns", "hot_entries", "hot_rows", "hot_long_rows", "tiled_entries",
                                       "tiled_device_bytes", "windows"), list(hot)))
        if d["store_trial"] and not d["auto_trial"]:  # (the two time slots carry AUTO's trial when both ran)
            d["store_trial_ns"] = {"plain": d.pop("trial_rowblock_ns"), "non_temporal": d.pop("trial_sliced_ns")}
            d["trial_rowblock_ns"] = d["trial_sliced_ns"] = 0
        return d

    # two-stage execution of a SLICED plan (include/spblas_gfx950.h: spmv_expand / spmv_reduce_rows)
    def bind_stages(self, x, y_base_ptr, dtype, alpha=1.0, beta=0.0):
        """Returns (expand, reduce_rows) callables with every argument pre-bound; reduce_rows(lo, hi)
        finishes the row-bins starting in [lo, hi).  y_base_ptr is the address of local row 0."""
        lib = _capi.lib()
        ct = _VT[dtype][1]
        a, b = ct(alpha), ct(beta)
        # (the BINDING thread's handle: the callables belong to the thread that bound them)
        hd = _Handle.current(x.device)
        h, p, xp, yp = hd.h, self.plan, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y_base_ptr)
        keep = (a, b, x, hd)
        # the stream that is current NOW: every other API call re-binds the handle to the then-current stream, so
        # the bound callables put the handle back on theirs before they launch
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        set_stream = lib.spblas_gfx950_set_stream

        def expand():
            set_stream(h, stream)
            check(lib.spblas_gfx950_spmv_expand(h, p, xp), "spmv_expand")

        def reduce_rows(lo, hi, _keep=keep):
            set_stream(h, stream)
            check(lib.spblas_gfx950_spmv_reduce_rows(h, p, ctypes.byref(a), ctypes.byref(b), yp, lo, hi),
                  "spmv_reduce_rows")

        return expand, reduce_rows

    def spmm_inspect(self):
        """The SpMM part of multiply_inspect (spblas_gfx950_spmm_inspect): column-locality probe of every block of
        32 rows; qualifying blocks go to the LDS-staged matrix-core kernel, long rows are cut into parts."""
        check(_capi.lib().spblas_gfx950_spmm_inspect(self._calling_handle().h, self.plan), "multiply_inspect")
        return self

    def spmm_info(self):
        arr = (ctypes.c_int64 * 4)()
        check(_capi.lib().spblas_gfx950_spmm_plan_info(self.plan, arr), "spblas_gfx950_spmm_plan_info")
        return dict(zip(("inspected", "panel_blocks", "panel_nnz", "long_rows"), list(arr)))

    def update_values(self, values):
        """Refresh the plan after A's values changed in place (only the SLICED re-tiling keeps
        a copy of them; the other algorithms read the caller's array on every call)."""
        check(_capi.lib().spblas_gfx950_spmv_plan_update_values(self._calling_handle(values.device).h, self.plan,
                                                                _ptr(values)), "update_values")
        self.stamp = _values_stamp(values)
        if self.tensors is not None:
            self.tensors = (self.tensors[0], self.tensors[1], values)

    def __del__(self):
        try:
            if self.plan:
                _capi.lib().spblas_gfx950_plan_destroy(self.handle.h, self.plan)
                self.plan = None
        except Exception:
            pass


class operation_info_t:
    """detail/operation_info_t.hpp:28-104: result_shape(), result_nnz(), backend state_."""

    def __init__(self, result_shape=(0, 0), result_nnz=0, state=None):
        self._result_shape, self._result_nnz = index(result_shape), int(result_nnz)
        self.state_ = state

    def result_shape(self):
        return self._result_shape

    def result_nnz(self):
        return self._result_nnz

    def update_impl_(self, result_shape, result_nnz):
        self._result_shape, self._result_nnz = index(result_shape), int(result_nnz)


class spgemm_state_t:
    """vendor/rocsparse/multiply_spgemm.hpp:28-230."""

    def __init__(self):
        self._handle = None
        self._state = None
        self._result_shape = index(0, 0)
        self._result_nnz = 0

    def result_shape(self):
        return self._result_shape

    def result_nnz(self):
        return self._result_nnz

    def _ensure(self, device):
        hd = _Handle.current(device)  # the CALLING thread's handle on its current stream (the state itself is handle-free)
        if self._state is None:
            self._handle = hd  # kept alive for the state's destruction
            st = ctypes.c_void_p()
            check(_capi.lib().spblas_gfx950_spgemm_create(hd.h, ctypes.byref(st)), "spblas_gfx950_spgemm_create")
            self._state = st
        return hd, self._state

    def info(self):
        """Introspection (spblas_gfx950_spgemm_info): which kernels the numeric passes of this state take."""
        if self._state is None:
            return {}
        raw = (ctypes.c_int64 * 8)()
        check(_capi.lib().spblas_gfx950_spgemm_info(self._state, raw), "spblas_gfx950_spgemm_info")
        return {"nnz_c": raw[0], "wave_per_row_rows": raw[1], "direct_rows": raw[2], "fills_by_rank": bool(raw[3])}

    def __del__(self):
        try:
            if self._state:
                _capi.lib().spblas_gfx950_spgemm_destroy(self._handle.h, self._state)
                self._state = None
        except Exception:
            pass


# --------------------------------------------------------------------------- helpers
_VT = {torch.float32: (_capi.F32, ctypes.c_float), torch.float64: (_capi.F64, ctypes.c_double)}
_OT = {torch.int32: _capi.I32, torch.int64: _capi.I64}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(
        t.data_ptr() if t is not None else 0)


def _vtype(t, what):
    if t.dtype not in _VT:
        raise TypeError(f"{what}: gfx950 backend supports float32/float64 values, got {t.dtype}")
    return _VT[t.dtype]


_NARROWED = {}  # id of an int64 index tensor -> (the tensor, its int32 copy); a handful of entries, oldest dropped first


def _int32_columns(a, what):
    """A csr_view / csc_view whose index array is int64 -- the slot this backend replaces admits them
    (vendor/rocsparse/types.hpp:16-24) -- as the same view over an int32 copy (spblas_gfx950_narrow_indices: ValueError if an
    index does not fit the other dimension).  The copy is made once per index tensor and kept, so that multiply_inspect and
    the multiplies that follow see ONE array (a plan is tied to its structure arrays); the kernels take int32 indices only.
    SpMV and SpMM operands only (what the slot's type list admits and this backend does not)."""
    csr = isinstance(a, csr_view)
    idx = a.colind() if csr else a.rowind()
    if idx is None or idx.dtype != torch.int64:
        return a
    hit = _NARROWED.get(id(idx))
    if hit is None or hit[0] is not idx:
        bound = a.shape()[1] if csr else a.shape()[0]
        dst = torch.empty(idx.numel(), dtype=torch.int32, device=idx.device)
        hd = _Handle.current(idx.device)
        rc = _capi.lib().spblas_gfx950_narrow_indices(hd.h, a.size(), _ptr(idx), _ptr(dst), int(bound))
        if rc == _capi.INVALID_VALUE:
            raise ValueError(f"{what}: a 64-bit index lies outside the matrix (or beyond 2^31 - 1)")
        check(rc, what)
        while len(_NARROWED) >= 8:
            _NARROWED.pop(next(iter(_NARROWED)))
        hit = _NARROWED[id(idx)] = (idx, dst)
    if csr:
        return csr_view(a.values(), a.rowptr(), hit[1], a.shape(), a.size())
    return csc_view(a.values(), a.colptr(), hit[1], a.shape(), a.size())


def _check_csr(a, what):
    if a.colind() is not None and a.colind().dtype != torch.int32:
        raise TypeError(f"{what}: column indices must be int32 (spblas::index_t on GPU backends; SpMV / SpMM also take int64)")
    if a.rowptr().dtype not in _OT:
        raise TypeError(f"{what}: row offsets must be int32 or int64")
    for t in (a.values(), a.rowptr(), a.colind()):
        if t is not None and not t.is_contiguous():
            raise ValueError(f"{what}: arrays must be contiguous")


def _reject_conjugated(*ts):
    if any(is_conjugated(t) for t in ts):
        raise RuntimeError("gfx950 backend does not support conjugated views.")  # spmv_impl.hpp:29-33


def _plan_key(a_base):
    """Identity of a plan = the structure arrays (the library's own check, csrc/spmv.hip); the value array may
    be rebound or change between multiplies like in the reference, where inspect holds no values at all."""
    return (a_base.rowptr().data_ptr(), a_base.colind().data_ptr(), tuple(a_base.shape()), a_base.size(),
            a_base.rowptr().dtype, a_base.values().dtype)


def _build_plan(a_base, alg=_capi.SPMV_AUTO, snapshot=0):
    """snapshot: the value of SPBLAS_GFX950_OPT_VALUE_SNAPSHOT -- 1 / True: AUTO may choose the plan that keeps a re-tiled
    copy of the values (matrix_opt operands); 2: and the plan keeps its source positions from the start."""
    hd = _Handle.current(a_base.rowptr().device)
    vt, _ = _vtype(a_base.values(), "multiply_inspect")
    plan = ctypes.c_void_p()
    m, n = a_base.shape()
    hd.set_option(_capi.OPT_VALUE_SNAPSHOT, int(snapshot))
    try:
        check(_capi.lib().spblas_gfx950_spmv_plan_create(hd.h, ctypes.byref(plan), m, n, a_base.size(),
                                                         _ptr(a_base.rowptr()), _ptr(a_base.colind()),
                                                         _ptr(a_base.values()), _OT[a_base.rowptr().dtype], vt, alg),
              "multiply_inspect")
    finally:
        hd.set_option(_capi.OPT_VALUE_SNAPSHOT, 0)
    return _Plan(hd, plan, _plan_key(a_base), (a_base.rowptr(), a_base.colind(), a_base.values()))


class _CscPlan:
    """multiply_inspect on a csc_view / transposed(csr) operand: the operand is materialised once as
    row-major CSR on the device (spblas_gfx950_csr_transpose: stable counting sort, the device
    counterpart of algorithms/transpose_impl.hpp:14-53) and planned like any CSR matrix, so the
    execute phase runs the regular kernels instead of the atomic scatter.  Holds a snapshot of the
    values (inspect may re-format the matrix, README.md:27-30)."""

    def __init__(self, a_csc, alg):
        self.key = _csc_key(a_csc)
        self.stamp = _values_stamp(a_csc.values())
        m, n = a_csc.shape()          # logical shape of the operand
        nnz = a_csc.size()
        dev, vals = a_csc.values().device, a_csc.values()
        # the device transpose takes 32-bit offsets; 64-bit ones that fit are narrowed once into a plan-owned copy
        # (multiply_inspect leaves operands with more than 2^31 - 1 entries to the plan-free kernels)
        self.colptr32 = a_csc.colptr() if a_csc.colptr().dtype == torch.int32 else a_csc.colptr().to(torch.int32)
        self.rowptr = torch.empty(m + 1, dtype=torch.int32, device=dev)
        self.colind = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        self.values = torch.empty(max(nnz, 1), dtype=vals.dtype, device=dev)
        hd = _Handle.current(dev)
        # the stored arrays are the CSR of the (n x m) transpose
        check(_capi.lib().spblas_gfx950_csr_transpose(hd.h, n, m, nnz, _ptr(self.colptr32), _ptr(a_csc.rowind()),
                                                      _ptr(vals), _ptr(self.rowptr), _ptr(self.colind),
                                                      _ptr(self.values), _vtype(vals, "multiply_inspect")[0]),
              "multiply_inspect")
        self.a_csr = csr_view(self.values[:nnz], self.rowptr, self.colind[:nnz], (m, n), nnz)
        self.plan = _build_plan(self.a_csr, alg, snapshot=True)  # the materialised form is a copy anyway
        # a self-contained plan (tiles with their own values, no hub rows, no hot split) needs the materialised arrays no
        # longer: they are released, and the inspected operand holds the plan alone (cfg2-sized: 1.43 instead of 2.43 x the
        # matrix).  SpMM plans and every other SpMV plan keep them.
        self.detached = False
        self._alg, self._vtype = alg, _vtype(vals, "multiply_inspect")[0]
        if alg != _capi.SPMV_ROWBLOCK and alg != _capi.SPMV_VECTOR and \
                _capi.lib().spblas_gfx950_spmv_plan_detach(hd.h, self.plan.plan) == _capi.SUCCESS:
            self.detached = True
            self.plan.tensors = None
            self.a_csr = self.rowptr = self.colind = self.values = None

    def refresh_if_stale(self, a_csc):
        """The operand's values were rebound or changed in place (torch / scale()) since inspect: transpose
        again into the same arrays (the structure is unchanged, so the plan stays valid) and refresh the plan's
        own copy, so that multiply reads current values like the reference (multiply_impl.hpp:48-52)."""
        if _values_stamp(a_csc.values()) == self.stamp:
            return
        if self.detached:  # (the arrays the plan could be refreshed from are gone: materialise and plan again)
            self.__init__(a_csc, self._alg)
            return
        m, n = a_csc.shape()
        nnz = a_csc.size()
        hd = _Handle.current(a_csc.values().device)
        check(_capi.lib().spblas_gfx950_csr_transpose(hd.h, n, m, nnz, _ptr(self.colptr32), _ptr(a_csc.rowind()),
                                                      _ptr(a_csc.values()), _ptr(self.rowptr), _ptr(self.colind),
                                                      _ptr(self.values), _vtype(a_csc.values(), "multiply")[0]),
              "multiply")
        self.stamp = _values_stamp(a_csc.values())
        self.key = _csc_key(a_csc)
        if self.plan.snapshot:
            self.plan.update_values(self.a_csr.values())

    def info(self):
        return self.plan.info()

    def held_bytes(self):
        """device bytes of the materialised row-major copy this plan holds next to the CSR plan's own (info()['device_bytes'])"""
        t = [self.rowptr, self.colind, self.values]
        return sum(x.numel() * x.element_size() for x in t if x is not None)


def _csc_key(a_csc):
    return (a_csc.colptr().data_ptr(), a_csc.rowind().data_ptr(), tuple(a_csc.shape()), a_csc.size(),
            a_csc.values().dtype)


def transpose_inspect(a, b):
    """transpose_inspect(a, b) (algorithms/transpose_impl.hpp:9-12): nothing to prepare."""
    return operation_info_t()


def transpose(*args):
    """transpose(a, b) / transpose(info, a, b): b = a^T for csr_view operands on the device, entries
    of every output row in source order exactly like the reference's counting sort
    (algorithms/transpose_impl.hpp:14-53).  b's arrays are caller-allocated."""
    a, b = args[-2], args[-1]
    if not (isinstance(a, csr_view) and isinstance(b, csr_view)):
        raise TypeError("transpose: csr_view operands")
    if a.shape()[0] != b.shape()[1] or a.shape()[1] != b.shape()[0]:
        raise ValueError("transpose: matrix dimensions are incompatible.")  # transpose_impl.hpp:17-21
    nnz = a.size()
    if b.values() is None or b.colind() is None or b.values().numel() < nnz or b.colind().numel() < nnz:
        raise RuntimeError("transpose: Transpose ran out of memory.")  # transpose_impl.hpp:22-25
    _check_csr(a, "transpose")
    # the device kernel works on 32-bit offsets (its positions are 32-bit: nnz <= 2^31 - 1, checked there); 64-bit
    # offset arrays -- csr_view is templated on the offset type, views/csr_view.hpp -- are narrowed on the way in and
    # widened on the way out
    if nnz >= 2 ** 31:
        raise ValueError("transpose: more than 2^31 - 1 stored entries are not supported on the device")
    a_rp = a.rowptr() if a.rowptr().dtype == torch.int32 else a.rowptr().to(torch.int32)
    b_rp = b.rowptr() if b.rowptr().dtype == torch.int32 else torch.empty(b.shape()[0] + 1, dtype=torch.int32,
                                                                          device=b.rowptr().device)
    hd = _Handle.current(a.rowptr().device)
    check(_capi.lib().spblas_gfx950_csr_transpose(hd.h, a.shape()[0], a.shape()[1], nnz, _ptr(a_rp),
                                                  _ptr(a.colind()), _ptr(a.values()), _ptr(b_rp),
                                                  _ptr(b.colind()), _ptr(b.values()),
                                                  _vtype(a.values(), "transpose")[0]), "transpose")
    if b_rp is not b.rowptr():
        b.rowptr()[:b.shape()[0] + 1].copy_(b_rp)
    _note_write(b.values(), b.colind())
    b.update(b.values(), b.rowptr(), b.colind(), b.shape(), nnz)  # transpose_impl.hpp:54


def scale(alpha, t):
    """scale(alpha, t) (algorithms/scale_impl.hpp:13-31): multiplies the stored values of a matrix view, or the
    elements of a dense vector, by alpha in place on the device.  An LDS-sliced SpMV plan of the matrix holds
    a snapshot of the values: call update_values on it afterwards."""
    base = get_ultimate_base(t)
    vals = base if _is_tensor(base) else base.values()
    if vals is None or not vals.is_contiguous():
        raise ValueError("scale: the values must be a contiguous device array")
    nvals = vals.numel() if _is_tensor(base) else base.size()
    vt, ct = _vtype(vals, "scale")
    a = ct(alpha)
    hd = _Handle.current(vals.device)
    check(_capi.lib().spblas_gfx950_scale(hd.h, nvals, ctypes.byref(a), _ptr(vals), vt), "scale")
    if not _is_tensor(base):  # plans holding a copy of these values see the change (_Plan.refresh_if_stale)
        _note_write(vals)


def _find_plan(info, a, a_base):
    key = _plan_key(a_base)
    if info is not None and isinstance(info.state_, _Plan) and info.state_.key == key:
        return info.state_
    mo = _get_matrix_opt(a)
    if mo is not None and isinstance(mo._plan, _Plan) and mo._plan.key == key:
        return mo._plan
    return None


# --------------------------------------------------------------------------- SpMV / SpMM
def _spmv(info, a, b, c, prepare_only=False):
    a_base, b_base = get_ultimate_base(a), get_ultimate_base(b)
    a_base = _int32_columns(a_base, "multiply")
    _reject_conjugated(a, b, c)
    if not _is_tensor(c) or c.dim() != 1:
        raise TypeError("multiply: the output vector must be a plain 1-D device tensor")
    op = _capi.OP_N
    csc_plan = None
    if isinstance(a_base, csc_view):
        if info is not None and isinstance(info.state_, _CscPlan) and info.state_.key == _csc_key(a_base):
            csc_plan = info.state_       # inspected: regular kernels on the materialised CSR
            csc_plan.refresh_if_stale(a_base)
            a_csr = csc_plan.a_csr
        else:
            # CSC = CSR of the transpose + TRANSPOSE (vendor/rocsparse/detail/get_transpose.hpp:19-29)
            a_csr = csr_view(a_base.values(), a_base.colptr(), a_base.rowind(),
                             (a_base.shape()[1], a_base.shape()[0]), a_base.size())
            op = _capi.OP_T
    else:
        a_csr = a_base
    detached = csc_plan is not None and csc_plan.detached
    if not detached:
        _check_csr(a_csr, "multiply")
    if a_base.shape()[0] != c.shape[0] or a_base.shape()[1] != b_base.shape[0]:
        # algorithms/multiply_impl.hpp:37-41
        raise ValueError("multiply: matrix and vector dimensions are incompatible.")
    vt, ct = _vtype(a_base.values() if detached else a_csr.values(), "multiply")
    if b_base.dtype != a_base.values().dtype or c.dtype != a_base.values().dtype:
        raise TypeError("multiply: A, x and y must share one value type")
    if not (b_base.is_contiguous() and c.is_contiguous()):
        raise ValueError("multiply: x and y must be contiguous")
    alpha_opt = get_scaling_factor(a, b)
    alpha = ct(1 if alpha_opt is None else alpha_opt)  # spmv_impl.hpp:35-37
    beta = ct(0)
    hd = _Handle.current(c.device)
    plan = csc_plan.plan if csc_plan else (_find_plan(info, a, a_base) if op == _capi.OP_N else None)
    if plan is not None and not csc_plan:
        plan.refresh_if_stale(a_csr.values())
    if detached:  # the plan is all there is of the materialised operand: no matrix arrays in the call
        m, n = a_base.shape()
        args = (hd.h, plan.plan, op, m, n, a_base.size(), ctypes.byref(alpha), None, None, None, _ptr(b_base),
                ctypes.byref(beta), _ptr(c), _capi.I32, vt)
    else:
        m, n = a_csr.shape()
        args = (hd.h, plan.plan if plan else None, op, m, n, a_csr.size(), ctypes.byref(alpha), _ptr(a_csr.rowptr()),
                _ptr(a_csr.colind()), _ptr(a_csr.values()), _ptr(b_base), ctypes.byref(beta), _ptr(c),
                _OT[a_csr.rowptr().dtype], vt)
    if prepare_only:
        return args, (alpha, beta, plan, a, b, c)  # keep the operands alive with the bound call
    check(_capi.lib().spblas_gfx950_spmv(*args), "multiply")


class prepared_multiply:
    """multiply(info, a, x, y) validated and bound ONCE; calling the object re-issues the same
    SpMV on the stream that was current at construction (the handle is put back on that stream before every
    launch: other API calls re-bind it to whatever stream is current when they run).  For solver-style loops
    where the Python host layer (view unwrapping, checks: tens of microseconds) would otherwise cost more than the
    kernel.  The operands must stay the same tensors; their contents may change between calls.  A plan carries
    workspaces and must not run on two streams at once."""

    def __init__(self, info, a, x, y):
        self._args, self._keep = _spmv(info, a, x, y, prepare_only=True)
        self._fn = _capi.lib().spblas_gfx950_spmv
        self._set_stream = _capi.lib().spblas_gfx950_set_stream
        self._stream = ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream)
        a_base = get_ultimate_base(a)
        self._plan = self._keep[2]
        self._values = a_base.values() if isinstance(a_base, csr_view) else None

    def __call__(self):
        if self._plan is not None and self._plan.snapshot and self._values is not None:
            self._plan.refresh_if_stale(self._values)
        self._set_stream(self._args[0], self._stream)
        rc = self._fn(*self._args)
        if rc:
            check(rc, "multiply")


def _spmm(info, a, b, c):
    a_base, b_base = get_ultimate_base(a), get_ultimate_base(b)
    a_base = _int32_columns(a_base, "multiply")
    _reject_conjugated(a, b, c)
    plan = None
    if isinstance(a_base, csc_view):
        # CSC operand (test/gtest/spmm_test.cpp:181): run the CSR kernel on the materialised row-major
        # form -- taken from the inspect result when there is one, otherwise transposed for this call
        if info is not None and isinstance(info.state_, _CscPlan) and info.state_.key == _csc_key(a_base) and \
                not info.state_.detached:  # (a plan inspected for SpMV may have released its row-major arrays)
            info.state_.refresh_if_stale(a_base)
            plan = info.state_.plan
            a_base = info.state_.a_csr
        else:
            a_base = _CscPlan(a_base, _capi.SPMV_VECTOR).a_csr
    elif isinstance(a_base, csr_view):
        plan = _find_plan(info, a, a_base)  # what multiply_inspect(a, B, C) left in info / in the matrix_opt
    if not isinstance(a_base, csr_view):
        raise NotImplementedError("gfx950 SpMM: A must have a csr_view or csc_view base")
    if not _is_tensor(c) or c.dim() != 2:
        raise TypeError("multiply: the output matrix must be a plain 2-D row-major device tensor")
    _check_csr(a_base, "multiply")
    if (a_base.shape()[0] != c.shape[0] or b_base.shape[1] != c.shape[1]
            or a_base.shape()[1] != b_base.shape[0]):
        raise ValueError("multiply: matrix dimensions are incompatible.")  # multiply_impl.hpp:70-76
    vt, ct = _vtype(a_base.values(), "multiply")
    if b_base.dtype != a_base.values().dtype or c.dtype != a_base.values().dtype:
        raise TypeError("multiply: A, B and C must share one value type")
    # Dense operands: layout_right (row-major) or layout_left (column-major, mdspan_col_major of detail/mdspan.hpp:31-36) --
    # the reference's CPU path takes any layout through mdspan's operator() (backend/view_customizations.hpp:230-240).
    # Anything else (overlapping or doubly strided views) is refused like a non-mdspan argument would be.
    m, k = a_base.shape()
    n = c.shape[1]

    def strides(t, rows):
        if t.numel() == 0:
            return max(n, 1), 1
        rs, cs = t.stride(0), t.stride(1)
        if (cs == 1 or n <= 1) and (rows <= 1 or rs >= n):      # layout_right, possibly a column window (rs > n)
            return (rs if rows > 1 else max(n, 1)), 1
        if (rs == 1 or rows <= 1) and (n <= 1 or cs >= rows):   # layout_left, possibly a row window (cs > rows)
            return 1, (cs if n > 1 else max(rows, 1))
        raise ValueError("multiply: dense operands must be row-major (layout_right) or column-major (layout_left)")

    (brs, bcs), (crs, ccs) = strides(b_base, k), strides(c, m)
    alpha_opt = get_scaling_factor(a, b)
    alpha, beta = ct(1 if alpha_opt is None else alpha_opt), ct(0)
    hd = _Handle.current(c.device)
    check(_capi.lib().spblas_gfx950_spmm_strided(hd.h, plan.plan if plan is not None else None, m, k, n, a_base.size(),
                                                 ctypes.byref(alpha),
                                                 _ptr(a_base.rowptr()), _ptr(a_base.colind()), _ptr(a_base.values()),
                                                 _ptr(b_base), brs, bcs, ctypes.byref(beta), _ptr(c), crs, ccs,
                                                 _OT[a_base.rowptr().dtype], vt), "multiply")


def _is_sparse(t):
    return isinstance(get_ultimate_base(t), (csr_view, csc_view))


def _split_info(args):
    if len(args) == 4:
        return args[0], args[1], args[2], args[3]
    if len(args) == 3:
        return None, args[0], args[1], args[2]
    raise TypeError("expected (a, b, c) or (info, a, b, c)")


def multiply(*args):
    """multiply(a, b, c) / multiply(info, a, b, c): c = a * b.
    SpMV when b is a vector, SpMM when b is a dense matrix
    (vendor/rocsparse/detail/spmv_impl.hpp:25,87; vendor/onemkl_sycl/spmm_impl.hpp:96-125)."""
    info, a, b, c = _split_info(args)
    b_base = get_ultimate_base(b)
    if isinstance(info, spgemm_state_t) or _is_sparse(b):
        raise TypeError("SpGEMM is two-phase on device backends: use multiply_compute + multiply_fill")
    if not _is_tensor(b_base):
        raise TypeError("multiply: b must be a device tensor (vector or row-major matrix)")
    if b_base.dim() == 1:
        return _spmv(info, a, b, c)
    return _spmm(info, a, b, c)


def multiply_inspect(*args, alg=_capi.SPMV_AUTO, values_will_change=False):
    """multiply_inspect(a, b, c) -> operation_info_t, or multiply_inspect(info, a, b, c).
    Builds the gfx950 row partition on device and stores it in the info's state_; when A
    is wrapped in matrix_opt it is cached there too (the oneMKL model,
    vendor/onemkl_sycl/spmm_impl.hpp:48-61, views/matrix_opt_impl.hpp:90-92).

    Values: for a plain csr_view the plan holds structure only and every multiply reads a.values() as the
    reference does (algorithms/multiply_impl.hpp:48-52).  Wrapping A in matrix_opt -- "this view owns an
    optimised form of the matrix" -- additionally lets AUTO pick the LDS-sliced plan, which re-tiles A and
    multiplies with its own copy of the values (5x faster when x misses every cache).  The copy is refreshed
    automatically when a.values() is rebound, modified in place through torch, or scaled with scale();
    after writing it with a kernel of your own call info.state_.update_values(values).  alg=SPMV_SLICED asks
    for that plan explicitly, with the same contract.  values_will_change=True (a time-stepping caller) makes such a plan
    keep the source position of every entry from the start (+4 B per entry): the first refresh is then a gather like every
    later one instead of a second inspect (SPBLAS_GFX950_OPT_VALUE_SNAPSHOT = 2)."""
    info, a, b, c = _split_info(args)
    ret = info is None
    if info is None:
        info = operation_info_t()
    if isinstance(info, spgemm_state_t):
        return None  # vendor/rocsparse/multiply_spgemm.hpp:232-235: no-op
    a_base = get_ultimate_base(a)
    _reject_conjugated(a, b, c)
    if isinstance(a_base, (csr_view, csc_view)) and not _is_sparse(b) and a_base.values() is not None:
        a_base = _int32_columns(a_base, "multiply_inspect")
    if isinstance(a_base, csr_view) and not _is_sparse(b) and a_base.values() is not None:
        _check_csr(a_base, "multiply_inspect")
        # the column-sliced plan serves SpMV only; SpMM uses the row partition (spmm_impl.hpp of the C++ layer)
        b_base = get_ultimate_base(b)
        is_spmm = _is_tensor(b_base) and b_base.dim() == 2
        mo = _get_matrix_opt(a)
        plan = _build_plan(a_base, _capi.SPMV_ROWBLOCK if is_spmm and alg == _capi.SPMV_AUTO else alg,
                           snapshot=0 if mo is None and alg != _capi.SPMV_SLICED else (2 if values_will_change else 1))
        if is_spmm:
            plan.spmm_inspect()
        info.state_ = plan
        if mo is not None:
            mo._plan = plan
    elif isinstance(a_base, csc_view) and _is_tensor(get_ultimate_base(b)) and a_base.size() < 2 ** 31:
        is_spmm = get_ultimate_base(b).dim() == 2
        info.state_ = _CscPlan(a_base, _capi.SPMV_ROWBLOCK if is_spmm else alg)
        if is_spmm:
            info.state_.plan.spmm_inspect()
    return info if ret else None


# --------------------------------------------------------------------------- SpGEMM
def _device_transposed(x):
    """csr_view of x^T in freshly allocated device arrays (spblas_gfx950_csr_transpose)."""
    m, n = x.shape()
    nnz = x.size()
    dev = x.rowptr().device
    t = csr_view(torch.empty(nnz, dtype=x.values().dtype, device=dev), torch.empty(n + 1, dtype=torch.int32, device=dev),
                 torch.empty(nnz, dtype=torch.int32, device=dev), (n, m), nnz)
    transpose(x, t)
    return t


def _csr_form(t_base, want_transposed, what):
    """CSR arrays of X (or of X^T) for a csr_view / csc_view X.  A csc_view's arrays ARE the CSR arrays of
    X^T (algorithms/transposed.hpp:7-21), so only one of the two forms needs the device transpose."""
    if isinstance(t_base, csc_view):
        t_base, want_transposed = transposed(t_base), not want_transposed
    if not isinstance(t_base, csr_view):
        raise NotImplementedError("gfx950 SpGEMM operands must have a csr_view or csc_view base")
    _check_csr(t_base, what)
    if t_base.rowptr().dtype != torch.int32:
        raise TypeError("SpGEMM: int32 row offsets only")
    return _device_transposed(t_base) if want_transposed else t_base


def _spgemm_operands(a, b, c, d=None):
    """Operands of the CSR kernel for any mix of csr_view / csc_view A, B, C (the eight combinations of
    algorithms/detail/spgemm/spgemm_{gustavsons,innerproduct,outerproduct}.hpp): with a CSR result the
    kernel runs on CSR(A), CSR(B); with a CSC result it computes C^T = B^T A^T, whose CSR arrays are C's CSC
    arrays.  Operands stored the other way round are transposed on the device for this call.
    Returns (a_eff, b_eff, c_eff, d_eff)."""
    a_base, b_base = get_ultimate_base(a), get_ultimate_base(b)
    if not isinstance(c, (csr_view, csc_view)):
        raise NotImplementedError("gfx950 SpGEMM result must be a csr_view or csc_view")
    _reject_conjugated(a, b)
    if (_shape_of(a_base)[0] != c.shape()[0] or _shape_of(b_base)[1] != c.shape()[1]
            or _shape_of(a_base)[1] != _shape_of(b_base)[0]):
        raise ValueError("multiply: matrix dimensions are incompatible.")  # spgemm_gustavsons.hpp:22-27
    d_base = None
    if d is not None:
        d_base = get_ultimate_base(d)
        _reject_conjugated(d)
        if tuple(_shape_of(d_base)) != tuple(c.shape()):
            raise ValueError("multiply: matrix dimensions are incompatible.")
    if isinstance(c, csr_view):
        return (_csr_form(a_base, False, "multiply_compute"), _csr_form(b_base, False, "multiply_compute"), c,
                None if d_base is None else _csr_form(d_base, False, "multiply_compute"))
    c_eff = csr_view(c.values(), c.colptr(), c.rowind(), (c.shape()[1], c.shape()[0]), c.size())
    return (_csr_form(b_base, True, "multiply_compute"), _csr_form(a_base, True, "multiply_compute"), c_eff,
            None if d_base is None else _csr_form(d_base, True, "multiply_compute"))


def _symbolic(state, a, b, c_user, d=None):
    a_base, b_base, c, d_base = _spgemm_operands(a, b, c_user, d)
    if c.rowptr() is None or c.rowptr().dtype != torch.int32 or c.rowptr().numel() < c.shape()[0] + 1:
        raise ValueError("multiply_compute: c's offsets must hold shape+1 int32 entries")
    hd, st = state._ensure(c.rowptr().device)
    state._keep = (a_base, b_base, d_base)  # device transposes of this call stay alive with the state
    nnz = ctypes.c_int64(0)
    m, k = a_base.shape()
    n = b_base.shape()[1]
    if d is not None:
        check(_capi.lib().spblas_gfx950_spgemm_set_addend(hd.h, st, d_base.size(), _ptr(d_base.rowptr()),
                                                          _ptr(d_base.colind())), "multiply_compute")
    else:
        check(_capi.lib().spblas_gfx950_spgemm_set_addend(hd.h, st, 0, None, None), "multiply_compute")
    state._has_addend = d is not None
    check(_capi.lib().spblas_gfx950_spgemm_symbolic(hd.h, st, m, k, n, a_base.size(), _ptr(a_base.rowptr()),
                                                    _ptr(a_base.colind()), b_base.size(), _ptr(b_base.rowptr()),
                                                    _ptr(b_base.colind()), _ptr(c.rowptr()), ctypes.byref(nnz)),
          "multiply_compute")
    state._result_shape, state._result_nnz = index(c_user.shape()), nnz.value
    state._last_colind = None


def _numeric(state, a, b, c_user, d=None):
    a_base, b_base, c, d_base = _spgemm_operands(a, b, c_user, d)
    if state._state is None:
        raise RuntimeError("multiply_fill: multiply_compute has not been called on this state")
    if (d is not None) != getattr(state, "_has_addend", False):
        raise RuntimeError("multiply_fill: the addend must be passed to both multiply_compute and multiply_fill")
    hd, st = state._ensure(c.rowptr().device)
    state._keep = (a_base, b_base, d_base)
    nnz = state._result_nnz
    cap = 0
    if c.values() is not None and c.colind() is not None:
        cap = min(c.values().numel(), c.colind().numel())
    if cap < nnz:
        raise RuntimeError("multiply: SpGEMM ran out of memory.")  # spgemm_gustavsons.hpp:44-48
    vt, ct = _vtype(a_base.values(), "multiply_fill")
    alpha_opt = get_scaling_factor(a, b)
    alpha = ct(1 if alpha_opt is None else alpha_opt)
    if d is not None:
        if d_base.values().dtype != a_base.values().dtype:
            raise TypeError("multiply_fill: the addend must have A's value type")
        beta_opt = get_scaling_factor(d)
        beta = ct(1 if beta_opt is None else beta_opt)  # multiply_spgemm.hpp:188-189
        check(_capi.lib().spblas_gfx950_spgemm_numeric_addend(
            hd.h, st, ctypes.byref(alpha), _ptr(a_base.rowptr()), _ptr(a_base.colind()), _ptr(a_base.values()),
            _ptr(b_base.rowptr()), _ptr(b_base.colind()), _ptr(b_base.values()), ctypes.byref(beta),
            _ptr(d_base.rowptr()), _ptr(d_base.colind()), _ptr(d_base.values()), _ptr(c.rowptr()),
            _ptr(c.colind()), _ptr(c.values()), cap, vt), "multiply_fill")
        _note_write(c.values(), c.colind())
    else:
        # Repeated fills of ONE result may leave its column indices alone (OPT_SPGEMM_KEEP_COLIND) -- but only when
        # this is provably the array the previous fill wrote and nobody touched it since: the same tensor object
        # (which this state keeps alive, so its address cannot have been recycled) with an unchanged version counter.
        # Writes made through this library's own kernels (another state filling the same arrays, add, transpose)
        # do not move torch's counter; the module-level write epoch of the address covers those.
        ci = c.colind()
        last = getattr(state, "_last_colind", None)
        keep = last is not None and last[0] is ci and last[1] == ci._version and last[2] == ci.data_ptr() and \
            last[3] == _value_epochs.get(ci.data_ptr(), 0)
        if keep:
            hd.set_option(_capi.OPT_SPGEMM_KEEP_COLIND, 1)
        try:
            check(_capi.lib().spblas_gfx950_spgemm_numeric(hd.h, st, ctypes.byref(alpha), _ptr(a_base.rowptr()),
                                                           _ptr(a_base.colind()), _ptr(a_base.values()),
                                                           _ptr(b_base.rowptr()), _ptr(b_base.colind()),
                                                           _ptr(b_base.values()), _ptr(c.rowptr()), _ptr(ci),
                                                           _ptr(c.values()), cap, vt), "multiply_fill")
        finally:
            if keep:
                hd.set_option(_capi.OPT_SPGEMM_KEEP_COLIND, 0)
        _note_write(c.values())
        if not keep:
            _note_write(ci)
        state._last_colind = (ci, ci._version, ci.data_ptr(), _value_epochs.get(ci.data_ptr(), 0))
    if isinstance(c_user, csr_view):
        c_user.update(c.values(), c.rowptr(), c.colind(), state._result_shape, nnz)  # spgemm_gustavsons.hpp:50-51
    else:
        c_user.update(c.values(), c.rowptr(), c.colind(), state._result_shape, nnz)  # the CSR arrays of C^T


def multiply_compute(*args):
    """multiply_compute(a, b, c) -> operation_info_t; multiply_compute(info, a, b, c);
    multiply_compute(spgemm_state, a, b, c)   (algorithms/multiply.hpp:48-52,
    vendor/rocsparse/multiply_spgemm.hpp:72-118,277-283).  Writes c.rowptr, reports nnz(C)."""
    if len(args) == 5:  # (spgemm_state, a, b, c, d): C = alpha*A*B + beta*D, multiply_spgemm.hpp:237-243
        if not isinstance(args[0], spgemm_state_t):
            raise TypeError("multiply_compute(state, a, b, c, d): the five-argument form takes a spgemm_state_t")
        return _symbolic(*args)
    info, a, b, c = _split_info(args)
    if isinstance(info, spgemm_state_t):
        return _symbolic(info, a, b, c)
    ret = info is None
    if info is None:
        info = operation_info_t()
    if not isinstance(info.state_, spgemm_state_t):
        info.state_ = spgemm_state_t()
    _symbolic(info.state_, a, b, c)
    info.update_impl_(info.state_.result_shape(), info.state_.result_nnz())
    return info if ret else None


def multiply_fill(info, a, b, c, d=None):
    """multiply_fill(info | spgemm_state, a, b, c[, d]) (algorithms/multiply.hpp:54-55,
    vendor/rocsparse/multiply_spgemm.hpp:123-145,245-250)."""
    state = info if isinstance(info, spgemm_state_t) else info.state_
    if not isinstance(state, spgemm_state_t):
        raise RuntimeError("multiply_fill: info does not come from multiply_compute")
    return _numeric(state, a, b, c, d)


# symbolic/numeric reuse family (vendor/rocsparse/multiply_spgemm.hpp:252-274,293-317)
def multiply_symbolic_compute(state, a, b, c, d=None):
    return _symbolic(state, a, b, c, d)


def multiply_symbolic_fill(state, a, b, c, d=None):
    """Leaves the STRUCTURE of C (rowptr and colind) in the caller's arrays, as the reference's symbolic stage does
    (vendor/rocsparse/multiply_spgemm.hpp:147-176; test/gtest/device/spgemm_reuse_test.cpp:325-400 copies those arrays
    afterwards and hands the copies to multiply_numeric).  The column indices come out of the same pass as the values
    here, so this is one numeric pass."""
    if state._state is None:
        raise RuntimeError("multiply_symbolic_fill: multiply_symbolic_compute has not been called")
    return _numeric(state, a, b, c, d)


def multiply_numeric(state, a, b, c, d=None):
    return _numeric(state, a, b, c, d)


# --------------------------------------------------------------------------- add
def _add_operands(a, b, c):
    a_base, b_base = get_ultimate_base(a), get_ultimate_base(b)
    if not (isinstance(a_base, csr_view) and isinstance(b_base, csr_view) and isinstance(c, csr_view)):
        raise NotImplementedError("gfx950 add supports CSR + CSR -> CSR")
    _reject_conjugated(a, b)
    for t in (a_base, b_base):
        _check_csr(t, "add")
        if t.rowptr().dtype != torch.int32:
            raise TypeError("add: int32 row offsets only")
    if tuple(a_base.shape()) != tuple(b_base.shape()) or tuple(b_base.shape()) != tuple(c.shape()):
        raise ValueError("add: matrix dimensions are incompatible.")  # add_impl.hpp:44-47,83-86
    if a_base.values().dtype != b_base.values().dtype:
        raise TypeError("add: a and b must have the same value type")
    return a_base, b_base


def _add_symbolic(state, a, b, c):
    a_base, b_base = _add_operands(a, b, c)
    if c.rowptr() is None or c.rowptr().dtype != torch.int32 or c.rowptr().numel() < c.shape()[0] + 1:
        raise ValueError("add_inspect: c.rowptr must hold shape[0]+1 int32 entries")
    hd, st = state._ensure(c.rowptr().device)
    nnz = ctypes.c_int64(0)
    m, n = a_base.shape()
    check(_capi.lib().spblas_gfx950_csr_add_symbolic(hd.h, st, m, n, a_base.size(), _ptr(a_base.rowptr()),
                                                     _ptr(a_base.colind()), b_base.size(), _ptr(b_base.rowptr()),
                                                     _ptr(b_base.colind()), _ptr(c.rowptr()), ctypes.byref(nnz)),
          "add_inspect")
    state._result_shape, state._result_nnz = index(m, n), nnz.value
    state._has_addend = True


def _add_numeric(state, a, b, c):
    a_base, b_base = _add_operands(a, b, c)
    hd, st = state._ensure(c.rowptr().device)
    nnz = state._result_nnz
    cap = 0
    if c.values() is not None and c.colind() is not None:
        cap = min(c.values().numel(), c.colind().numel())
    if cap < nnz:  # add_impl.hpp:67-72
        raise RuntimeError("add: ran out of memory.  CSR output view has insufficient memory.")
    vt, ct = _vtype(a_base.values(), "add")
    sa, sb = get_scaling_factor(a), get_scaling_factor(b)
    alpha, beta = ct(1 if sa is None else sa), ct(1 if sb is None else sb)
    check(_capi.lib().spblas_gfx950_csr_add_numeric(hd.h, st, ctypes.byref(alpha), _ptr(a_base.rowptr()),
                                                    _ptr(a_base.colind()), _ptr(a_base.values()), ctypes.byref(beta),
                                                    _ptr(b_base.rowptr()), _ptr(b_base.colind()),
                                                    _ptr(b_base.values()), _ptr(c.rowptr()), _ptr(c.colind()),
                                                    _ptr(c.values()), cap, vt), "add")
    _note_write(c.values(), c.colind())
    c.update(c.values(), c.rowptr(), c.colind(), state._result_shape, nnz)  # add_impl.hpp:75-76


def add_inspect(*args):
    """add_inspect(a, b, c) -> operation_info_t / add_inspect(info, a, b, c)
    (algorithms/add.hpp:15-19, add_impl.hpp:79-108): structural nnz of A + B; also writes c.rowptr."""
    info, a, b, c = _split_info(args)
    ret = info is None
    if info is None:
        info = operation_info_t()
    if not isinstance(info.state_, spgemm_state_t):
        info.state_ = spgemm_state_t()
    _add_symbolic(info.state_, a, b, c)
    info.update_impl_(info.state_.result_shape(), info.state_.result_nnz())
    return info if ret else None


def add_compute(info, a, b, c):
    """add_compute(info, a, b, c) (add_impl.hpp:110-113): fill c's colind / values."""
    if not isinstance(info.state_, spgemm_state_t) or info.state_._state is None:
        raise RuntimeError("add_compute: info does not come from add_inspect")
    return _add_numeric(info.state_, a, b, c)


def add(a, b, c):
    """add(a, b, c): c = a + b for CSR operands (add_impl.hpp:40-77); c must already own enough
    room for the result (csr_builder semantics)."""
    state = spgemm_state_t()
    _add_symbolic(state, a, b, c)
    _add_numeric(state, a, b, c)


# --------------------------------------------------------------------------- SpTRSV
class upper_triangle_t:      # detail/triangular_types.hpp:5-8
    pass


class lower_triangle_t:      # detail/triangular_types.hpp:10-13
    pass


class implicit_unit_diagonal_t:  # detail/triangular_types.hpp:15-18
    pass


class explicit_diagonal_t:       # detail/triangular_types.hpp:20-23
    pass


upper_triangle, lower_triangle = upper_triangle_t(), lower_triangle_t()
implicit_unit_diagonal, explicit_diagonal = implicit_unit_diagonal_t(), explicit_diagonal_t()


class _TrsvPlan:
    """triangular_solve_inspect state: level sets of the dependency graph (spblas_gfx950_sptrsv_create)."""

    def __init__(self, hd, plan, key):
        self.hd, self.plan, self.key = hd, plan, key

    def info(self):
        arr = (ctypes.c_int64 * 4)()
        check(_capi.lib().spblas_gfx950_sptrsv_info(self.plan, arr), "spblas_gfx950_sptrsv_info")
        return dict(zip(("levels", "max_level_width", "launches_per_solve", "lanes_per_row"), list(arr)))

    def check_status(self):
        """Synchronises the stream; raises if a device-side wait of the last solve gave up (spblas_gfx950_sptrsv_status)."""
        st = ctypes.c_int(0)
        hd = _Handle.current(torch.device("cuda", self.hd.device))
        check(_capi.lib().spblas_gfx950_sptrsv_status(hd.h, self.plan, ctypes.byref(st)), "spblas_gfx950_sptrsv_status")
        if st.value != 0:
            raise RuntimeError("triangular_solve: a device-side wait ran into its bound; the result is not valid")

    def __del__(self):
        try:
            if self.plan:
                _capi.lib().spblas_gfx950_sptrsv_destroy(self.hd.h, self.plan)
                self.plan = None
        except Exception:
            pass


def _trsv_operands(a, uplo, diag, b, x):
    a_base = get_ultimate_base(a)
    if not isinstance(a_base, csr_view):
        raise NotImplementedError("gfx950 triangular_solve supports csr_view operands")
    _reject_conjugated(a, b, x)
    _check_csr(a_base, "triangular_solve")
    if a_base.rowptr().dtype != torch.int32:
        raise TypeError("triangular_solve: int32 row offsets only")
    if not isinstance(uplo, (upper_triangle_t, lower_triangle_t)):
        raise TypeError("triangular_solve: uplo must be upper_triangle_t or lower_triangle_t")  # :48-49 static_assert
    if not isinstance(diag, (implicit_unit_diagonal_t, explicit_diagonal_t)):
        raise TypeError("triangular_solve: diag must be implicit_unit_diagonal_t or explicit_diagonal_t")
    if get_scaling_factor(x) is not None:
        raise NotImplementedError("gfx950 triangular_solve: x must be a plain vector")
    b = get_ultimate_base(b)  # scaled(alpha, b) is allowed (examples/simple_sptrsv.cpp:49-53): applied to x afterwards
    m, n = a_base.shape()
    # the reference asserts squareness and matching vector lengths (triangular_solve_impl.hpp:50-53)
    if m != n or not _is_tensor(b) or not _is_tensor(x) or b.dim() != 1 or x.dim() != 1 or x.numel() != n or \
            b.numel() != m:
        raise ValueError("triangular_solve: matrix and vector dimensions are incompatible.")
    if b.dtype != a_base.values().dtype or x.dtype != a_base.values().dtype:
        raise TypeError("triangular_solve: b and x must have A's value type")
    if not (b.is_contiguous() and x.is_contiguous()):
        raise ValueError("triangular_solve: vectors must be contiguous")
    return a_base


def _trsv_key(a_base, uplo, diag):
    return (a_base.rowptr().data_ptr(), a_base.colind().data_ptr(), tuple(a_base.shape()), a_base.size(),
            type(uplo), type(diag))


def _trsv_plan(a_base, uplo, diag):
    hd = _Handle.current(a_base.rowptr().device)
    plan = ctypes.c_void_p()
    check(_capi.lib().spblas_gfx950_sptrsv_create(
        hd.h, ctypes.byref(plan), a_base.shape()[0], a_base.size(), _ptr(a_base.rowptr()), _ptr(a_base.colind()),
        _capi.UPPER if isinstance(uplo, upper_triangle_t) else _capi.LOWER,
        _capi.DIAG_UNIT if isinstance(diag, implicit_unit_diagonal_t) else _capi.DIAG_EXPLICIT),
        "triangular_solve_inspect")
    return _TrsvPlan(hd, plan, _trsv_key(a_base, uplo, diag))


def triangular_solve_inspect(*args):
    """triangular_solve_inspect(a, uplo, diag, b, x) -> operation_info_t, or with a leading info
    (algorithms/triangular_solve.hpp:8-15).  The reference's CPU inspect is empty
    (triangular_solve_impl.hpp:13-38); here it builds the level sets the device solve needs."""
    if len(args) == 6:
        info, a, uplo, diag, b, x = args
        ret = False
    elif len(args) == 5:
        a, uplo, diag, b, x = args
        info, ret = operation_info_t(), True
    else:
        raise TypeError("expected (a, uplo, diag, b, x) or (info, a, uplo, diag, b, x)")
    a_base = _trsv_operands(a, uplo, diag, b, x)
    info.state_ = _trsv_plan(a_base, uplo, diag)
    return info if ret else None


def triangular_solve(*args):
    """triangular_solve(a, uplo, diag, b, x) / triangular_solve(info, a, uplo, diag, b, x):
    x = inv(A) b using only the named triangle of A (triangular_solve_impl.hpp:41-107)."""
    if len(args) == 6:
        info, a, uplo, diag, b, x = args
    elif len(args) == 5:
        a, uplo, diag, b, x = args
        info = None
    else:
        raise TypeError("expected (a, uplo, diag, b, x) or (info, a, uplo, diag, b, x)")
    a_base = _trsv_operands(a, uplo, diag, b, x)
    plan = info.state_ if info is not None and isinstance(info.state_, _TrsvPlan) else None
    if plan is None or plan.key != _trsv_key(a_base, uplo, diag):
        plan = _trsv_plan(a_base, uplo, diag)  # no usable inspect result: analyse now
        if info is not None:
            info.state_ = plan
    hd = _Handle.current(a_base.rowptr().device)
    vt, ct = _vtype(a_base.values(), "triangular_solve")
    sa = get_scaling_factor(a)
    alpha = ct(1 if sa is None else sa)
    sb = get_scaling_factor(b)
    check(_capi.lib().spblas_gfx950_sptrsv_solve(hd.h, plan.plan, a_base.shape()[0], a_base.size(),
                                                 ctypes.byref(alpha), _ptr(a_base.rowptr()), _ptr(a_base.colind()),
                                                 _ptr(a_base.values()), _ptr(get_ultimate_base(b)), _ptr(x), vt),
          "triangular_solve")
    if sb is not None:  # the solve is linear in b: x = inv(A) (s b) = s inv(A) b
        beta = ct(sb)
        check(_capi.lib().spblas_gfx950_scale(hd.h, x.numel(), ctypes.byref(beta), _ptr(x), vt), "triangular_solve")
