"""-m gpu: SpMV on shapes far from the benchmark's (square, 10 entries per row) through every plan.

The reference's loops (/root/reference/include/spblas/algorithms/multiply_impl.hpp:33-53) do not care about shape; a
re-tiled plan does: its slice count follows the column count, its bin count the row count, its run lengths the
density.  Each case below pushes one of those to an end: almost no entries in millions of rows, one column, one row,
a hundred million columns (thousands of slices), the same column repeated inside a row, a block-diagonal band (every
slice sees few bins), entries only in the last rows and columns.
"""
import zlib

import numpy as np
import pytest
import torch

import gpu_util as G
import spblas_reference_amd as sp
import util
from oracle import oracle
from spblas_reference_amd import _capi

pytestmark = pytest.mark.gpu

ALGS = {"noplan": None, "auto": _capi.SPMV_AUTO, "vector": _capi.SPMV_VECTOR, "rowblock": _capi.SPMV_ROWBLOCK,
        "sliced": _capi.SPMV_SLICED}


def _csr(m, rows, cols, rng):
    order = np.argsort(rows, kind="stable")
    rows, cols = rows[order], cols[order]
    rowptr = np.zeros(m + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    np.cumsum(rowptr, out=rowptr)
    return rowptr.astype(np.int32), cols.astype(np.int32), (rng.random(rows.size) - 0.5)


def make(case):
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    if case == "few_entries_in_5M_rows":
        m, n = 5_000_000, 3_000_000
        rows, cols = rng.integers(0, m, 1000), rng.integers(0, n, 1000)
    elif case == "one_column":
        m, n = 1_000_000, 1
        rows, cols = np.arange(m), np.zeros(m, np.int64)
    elif case == "one_row":
        m, n = 1, 2_000_000
        rows, cols = np.zeros(1_500_000, np.int64), rng.integers(0, n, 1_500_000)
    elif case == "100M_columns":
        m, n = 200_000, 100_000_000
        rows, cols = rng.integers(0, m, 2_000_000), rng.integers(0, n, 2_000_000)
    elif case == "column_repeated_in_row":
        m, n = 50_000, 60_000
        rows = np.repeat(np.arange(m), 40)
        cols = np.repeat(rng.integers(0, n, m), 40)  # 40 copies of one column per row
    elif case == "block_band":
        m, n = 400_000, 400_000
        rows = np.repeat(np.arange(m), 8)
        cols = np.clip(rows + rng.integers(-200, 200, rows.size), 0, n - 1)
    elif case == "last_rows_and_columns_only":
        m, n = 1_000_000, 1_000_000
        rows, cols = rng.integers(m - 3000, m, 500_000), rng.integers(n - 100, n, 500_000)
    elif case == "tall_and_dense_rows":
        m, n = 300, 3_000_000
        rows, cols = rng.integers(0, m, 3_000_000), rng.integers(0, n, 3_000_000)
    else:
        raise KeyError(case)
    rowptr, colind, values = _csr(m, rows, cols, rng)
    return (m, n), rowptr, colind, values


CASES = ["few_entries_in_5M_rows", "one_column", "one_row", "100M_columns", "column_repeated_in_row", "block_band",
         "last_rows_and_columns_only", "tall_and_dense_rows"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("case", CASES)
def test_spmv_shapes_far_from_the_benchmark(gpu, case, dtype):
    shape, rowptr, colind, values = make(case)
    values = values.astype(dtype)
    m, n = shape
    x_h = (np.random.default_rng(9).random(n) - 0.5).astype(dtype)
    y_ref = oracle.spmv(shape, rowptr, colind, values, x_h)
    absrow = oracle.spmv_absrow(rowptr, colind, values, x_h)
    lens = np.diff(rowptr)
    a = sp.csr_view(G.dev(values), G.dev(rowptr), G.dev(colind), shape, int(rowptr[-1]))
    x = G.dev(x_h)
    for name, alg in ALGS.items():
        y = torch.full((m,), float("nan"), dtype=x.dtype, device="cuda")
        if alg is None:
            sp.multiply(a, x, y)
        else:
            info = sp.multiply_inspect(a, x, y, alg=alg)
            if name != "auto":  # every forced plan is built on every one of these shapes (no silent fall-back)
                assert info.state_.info()["alg"] == alg, f"{case}: forced {name} ended up as {info.state_.info()['alg']}"
            sp.multiply(info, a, x, y)
            del info
        torch.cuda.synchronize()
        util.assert_parity(G.host(y), y_ref, absrow, dtype, row_len=lens, what=f"{case} / {name}")
