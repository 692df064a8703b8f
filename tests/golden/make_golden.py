"""Generates the known-answer fixtures in this directory.

The reference cannot be run in this image and stores no vectors of its own, so these
fixtures are computed HERE with exact Python integer arithmetic -- independently of both
oracle/ and the HIP kernels -- on small-integer data.  Every intermediate is an integer
below 2**24, so float32/float64 evaluation in ANY summation order reproduces the expected
arrays bit for bit: they pin the oracle and the GPU kernels exactly, including the edge
cases the reference's generator produces (unsorted columns, backend/generate.hpp:112-117)
or merely permits (duplicate columns, empty rows, one very long row, numeric cancellation
in SpGEMM that must still count structurally, spgemm_gustavsons.hpp:57-89).

Run:  python tests/golden/make_golden.py     (deterministic; fixtures are committed)
"""
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def rand_csr(rng, m, n, row_lens, vmax=8, allow_dups=True):
    rowptr, colind, values = [0], [], []
    for L in row_lens:
        cols = [rng.randrange(n) for _ in range(L)] if allow_dups else rng.sample(range(n), L)
        colind += cols
        values += [rng.choice([v for v in range(-vmax, vmax + 1) if v != 0]) for _ in range(L)]
        rowptr.append(len(colind))
    return rowptr, colind, values


def exact_spmv(rowptr, colind, values, x, sa=1, sx=1):
    return [sum(sa * values[p] * sx * x[colind[p]] for p in range(rowptr[i], rowptr[i + 1]))
            for i in range(len(rowptr) - 1)]


def save(name, **kw):
    np.savez(os.path.join(HERE, name + ".npz"), **kw)


def spmv_case(name, m, n, row_lens, dtype, seed, sa=None, sx=None):
    rng = random.Random(seed)
    rowptr, colind, values = rand_csr(rng, m, n, row_lens)
    x = [rng.randrange(-4, 5) for _ in range(n)]
    y = exact_spmv(rowptr, colind, values, x, sa or 1, sx or 1)
    assert max(abs(v) for v in y + [0]) < 2 ** 24
    kw = dict(kind="spmv", shape=np.array([m, n]), rowptr=np.array(rowptr, np.int32),
              colind=np.array(colind, np.int32), values=np.array(values, dtype), x=np.array(x, dtype),
              y=np.array(y, dtype))
    if sa is not None:
        kw["scale_a"] = np.array(sa, dtype)
    if sx is not None:
        kw["scale_x"] = np.array(sx, dtype)
    save(name, **kw)


def main():
    rng = random.Random(0)
    # examples/simple_spmv.cpp:12-14 shape: 100 x 100 with 10 nonzeros (most rows empty)
    lens = [0] * 100
    for _ in range(10):
        lens[rng.randrange(100)] += 1
    spmv_case("spmv_example_100x100_nnz10", 100, 100, lens, np.float32, 1)
    # ragged: empty rows, duplicates, unsorted columns, one 3000-entry row, trailing empties
    lens = [rng.choice([0, 0, 1, 2, 3, 7, 10, 33, 64, 65]) for _ in range(400)]
    lens[137] = 3000
    lens[-5:] = [0] * 5
    spmv_case("spmv_ragged_f32", 400, 513, lens, np.float32, 2)
    spmv_case("spmv_ragged_f64", 400, 513, lens, np.float64, 3)
    spmv_case("spmv_scaled_a_m10", 120, 90, [rng.randrange(0, 20) for _ in range(120)], np.float32, 4, sa=-10)
    spmv_case("spmv_scaled_x_5", 120, 90, [rng.randrange(0, 20) for _ in range(120)], np.float32, 5, sx=5)
    # rows much longer than one nnz window (2048 fp32 / 1024 fp64): the split-row path
    lens = [5] * 64
    lens[3] = 9000
    lens[40] = 2500
    lens[41] = 2049
    spmv_case("spmv_long_rows_f32", 64, 2000, lens, np.float32, 6)
    spmv_case("spmv_long_rows_f64", 64, 2000, lens, np.float64, 7)

    # SpMM: n = 8 (vector width 4) and n = 3 (scalar path)
    for n, seed in ((8, 8), (3, 9), (130, 10)):
        r = random.Random(seed)
        m, k = 60, 50
        rowptr, colind, values = rand_csr(r, m, k, [r.randrange(0, 12) for _ in range(m)])
        B = [[r.randrange(-4, 5) for _ in range(n)] for _ in range(k)]
        C = [[sum(values[p] * B[colind[p]][j] for p in range(rowptr[i], rowptr[i + 1])) for j in range(n)]
             for i in range(m)]
        save(f"spmm_n{n}", kind="spmm", shape=np.array([m, k]), rowptr=np.array(rowptr, np.int32),
             colind=np.array(colind, np.int32), values=np.array(values, np.float32),
             B=np.array(B, np.float32), C=np.array(C, np.float32))

    # SpGEMM: duplicates inside A rows, and products that cancel numerically (must still be
    # counted and stored, with value 0)
    r = random.Random(11)
    m, k, n = 70, 40, 55
    ar, ac, av = rand_csr(r, m, k, [r.randrange(0, 9) for _ in range(m)])
    br, bc, bv = rand_csr(r, k, n, [r.randrange(0, 9) for _ in range(k)], allow_dups=False)
    # force a cancellation: row 0 of A = {(k0, +1), (k0, -1)}
    k0 = next(i for i in range(k) if br[i + 1] > br[i])
    la = ar[1] - ar[0]
    ac[ar[0]:ar[1]] = []
    av[ar[0]:ar[1]] = []
    ac[0:0] = [k0, k0]
    av[0:0] = [1, -1]
    ar = [0] + [p - la + 2 for p in ar[1:]]
    c_rowptr, c_colind, c_values = [0], [], []
    for i in range(m):
        acc = {}
        for p in range(ar[i], ar[i + 1]):
            for q in range(br[ac[p]], br[ac[p] + 1]):
                acc[bc[q]] = acc.get(bc[q], 0) + av[p] * bv[q]
        for j in sorted(acc):
            c_colind.append(j)
            c_values.append(acc[j])
        c_rowptr.append(len(c_colind))
    assert c_rowptr[1] > 0 and all(v == 0 for v in c_values[:c_rowptr[1]])
    save("spgemm_dups_cancel", kind="spgemm", a_shape=np.array([m, k]), b_shape=np.array([k, n]),
         a_rowptr=np.array(ar, np.int32), a_colind=np.array(ac, np.int32), a_values=np.array(av, np.float32),
         b_rowptr=np.array(br, np.int32), b_colind=np.array(bc, np.int32), b_values=np.array(bv, np.float32),
         c_rowptr=np.array(c_rowptr, np.int32), c_colind=np.array(c_colind, np.int32),
         c_values=np.array(c_values, np.float32), c_nnz=np.array(len(c_colind)))


def union_rows(m, parts):
    """parts: list of (rowptr, colind, values, scale).  Exact row-wise sum, columns ascending."""
    c_rowptr, c_colind, c_values = [0], [], []
    for i in range(m):
        acc = {}
        for rp, ci, v, sc in parts:
            for p in range(rp[i], rp[i + 1]):
                acc[ci[p]] = acc.get(ci[p], 0) + sc * v[p]
        for j in sorted(acc):
            c_colind.append(j)
            c_values.append(acc[j])
        c_rowptr.append(len(c_colind))
    return c_rowptr, c_colind, c_values


def main_8f():
    """Fixtures for the SURVEY 8f rows: add, four-argument SpGEMM, triangular solve."""
    from fractions import Fraction
    i32 = lambda a: np.array(a, np.int32)  # noqa: E731
    # add: duplicates inside rows, entries that cancel (kept, value 0), empty rows, scaled views
    r = random.Random(21)
    m, n = 90, 64
    ar, ac, av = rand_csr(r, m, n, [r.choice([0, 0, 1, 3, 8, 20]) for _ in range(m)])
    br, bc, bv = rand_csr(r, m, n, [r.choice([0, 2, 5, 40]) for _ in range(m)])
    # row 1: B = -A exactly (structural entries with value 0)
    la, lb = ar[2] - ar[1], br[2] - br[1]
    cols = ac[ar[1]:ar[2]]
    bc[br[1]:br[2]] = cols
    bv[br[1]:br[2]] = [-v for v in av[ar[1]:ar[2]]]
    br = br[:2] + [p - lb + la for p in br[2:]]
    for sa, sb, tag in ((1, 1, "plain"), (2, -3, "scaled")):
        cr, cc, cv = union_rows(m, [(ar, ac, av, sa), (br, bc, bv, sb)])
        assert max(abs(v) for v in cv + [0]) < 2 ** 24
        save(f"add_{tag}", kind="add", shape=np.array([m, n]), a_rowptr=i32(ar), a_colind=i32(ac),
             a_values=np.array(av, np.float32), b_rowptr=i32(br), b_colind=i32(bc), b_values=np.array(bv, np.float32),
             scale_a=np.array(sa, np.float32), scale_b=np.array(sb, np.float32), c_rowptr=i32(cr), c_colind=i32(cc),
             c_values=np.array(cv, np.float32), c_nnz=np.array(len(cc)))

    # C = alpha*A*B + beta*D
    r = random.Random(22)
    m, k, n = 60, 45, 50
    ar, ac, av = rand_csr(r, m, k, [r.randrange(0, 7) for _ in range(m)])
    br, bc, bv = rand_csr(r, k, n, [r.randrange(0, 7) for _ in range(k)], allow_dups=False)
    dr, dc, dv = rand_csr(r, m, n, [r.choice([0, 1, 4, 30]) for _ in range(m)])
    alpha, beta = 2, -3
    c_rowptr, c_colind, c_values = [0], [], []
    for i in range(m):
        acc = {}
        for p in range(ar[i], ar[i + 1]):
            for q in range(br[ac[p]], br[ac[p] + 1]):
                acc[bc[q]] = acc.get(bc[q], 0) + alpha * av[p] * bv[q]
        for p in range(dr[i], dr[i + 1]):
            acc[dc[p]] = acc.get(dc[p], 0) + beta * dv[p]
        for j in sorted(acc):
            c_colind.append(j)
            c_values.append(acc[j])
        c_rowptr.append(len(c_colind))
    assert max(abs(v) for v in c_values) < 2 ** 24
    save("spgemm4_scaled", kind="spgemm4", a_shape=np.array([m, k]), b_shape=np.array([k, n]),
         d_shape=np.array([m, n]), a_rowptr=i32(ar), a_colind=i32(ac), a_values=np.array(av, np.float32),
         b_rowptr=i32(br), b_colind=i32(bc), b_values=np.array(bv, np.float32), d_rowptr=i32(dr), d_colind=i32(dc),
         d_values=np.array(dv, np.float32), alpha=np.array(alpha, np.float32), beta=np.array(beta, np.float32),
         c_rowptr=i32(c_rowptr), c_colind=i32(c_colind), c_values=np.array(c_values, np.float32),
         c_nnz=np.array(len(c_colind)))

    # triangular solves on a GENERAL matrix (both triangles stored, the other one must be ignored),
    # entries in {-1, 1}, diagonals in {1, -1, 2, -2, 4}: every x_i is a dyadic rational, exact in
    # float32 for any summation order as long as the partial sums stay below 2**24 ulps
    r = random.Random(23)
    n = 120
    rowptr, colind, values = [0], [], []
    for i in range(n):
        cols = sorted(set(r.randrange(n) for _ in range(r.choice([0, 1, 2, 3]))) - {i})
        ent = [(c, r.choice([-1, 1])) for c in cols] + [(i, r.choice([1, -1, 2, -2, 4]))]
        r.shuffle(ent)
        colind += [c for c, _ in ent]
        values += [v for _, v in ent]
        rowptr.append(len(colind))
    b = [r.randrange(-3, 4) for _ in range(n)]
    out = {}
    for upper in (False, True):
        for unit in (False, True):
            x = [Fraction(0)] * n
            for t in range(n):
                i = n - 1 - t if upper else t
                dot, diag = Fraction(0), None
                for p in range(rowptr[i], rowptr[i + 1]):
                    c = colind[p]
                    if (c > i) if upper else (c < i):
                        dot += values[p] * x[c]
                    elif c == i:
                        diag = values[p]
                x[i] = (b[i] - dot) if unit else (b[i] - dot) / diag
            for v in x:  # exactly representable, with head-room for the intermediate sums
                assert v.denominator & (v.denominator - 1) == 0 and abs(v.numerator) * 64 < 2 ** 24, v
            out[f"x_{'upper' if upper else 'lower'}_{'unit' if unit else 'explicit'}"] = np.array(
                [float(v) for v in x], np.float32)
    save("trsv_general_dyadic", kind="trsv", shape=np.array([n, n]), rowptr=i32(rowptr), colind=i32(colind),
         values=np.array(values, np.float32), b=np.array(b, np.float32), **out)


if __name__ == "__main__":
    main()
    main_8f()
