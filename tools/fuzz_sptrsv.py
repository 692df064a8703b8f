"""Randomised stress of the device triangular solve (run on the GPU box):
    python tools/fuzz_sptrsv.py [iterations] [first_seed]
Random sizes and structures -- random triangles of several densities, banded ones (few, wide levels), chains (one row per
level: the narrow-run path), a long dense row or column, empty rows, repeated diagonal entries (the last one wins), entries
of the other triangle (ignored) -- lower / upper, explicit / unit diagonal, fp32 / fp64, with and without inspect, the
launch-per-level form and small cooperative grids next to the default cooperative kernel.  Checked like tests/test_gpu_sptrsv.py: row-wise
backward error at the parity tolerance and forward error against the CPU oracle."""
import os, sys
import numpy as np
import scipy.sparse as sps
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_sptrsv as T  # noqa: E402  (device_solve / check of the test-suite)

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
KNOBS = ["SPBLAS_GFX950_TRSV_COOP", "SPBLAS_GFX950_TRSV_NARROW", "SPBLAS_GFX950_TRSV_KAHN", "SPBLAS_GFX950_TRSV_COOP_GRID"]
bad = 0
for it in range(iters):
    rng = np.random.default_rng(seed0 + it)
    for k in KNOBS:
        os.environ.pop(k, None)
    n = int(rng.choice([1, 2, 17, 300, 4000, 60000, 400000]))
    upper, unit = bool(rng.integers(2)), bool(rng.integers(2))
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    kind = rng.choice(["random", "dense_small", "banded", "chain", "arrow", "emptyish", "dupdiag"])
    if kind == "random":
        M = T.tri_system(n, min(1.0, float(rng.choice([2, 8, 30])) / max(n, 1)), upper, rng)
    elif kind == "dense_small":
        n = min(n, 300)
        M = T.tri_system(n, 0.5, upper, rng)
    elif kind == "banded":
        bw = max(0, min(int(rng.choice([1, 3, 20])), n - 1))
        diags = [rng.random(n) * 0.1 for _ in range(bw)]
        offs = [(j + 1) if upper else -(j + 1) for j in range(bw)]
        M = ((sps.diags(diags, offs, shape=(n, n)) if bw else sps.csr_matrix((n, n))) + sps.diags(1.5 + rng.random(n))).tocsr()
    elif kind == "chain":
        M = ((sps.diags([0.3 * rng.random(n)], [1 if upper else -1], shape=(n, n)) if n > 1 else sps.csr_matrix((n, n))) +
             sps.diags(1.0 + rng.random(n))).tocsr()
    elif kind == "arrow":  # one long row (lower) / column-like structure + diagonal
        r = np.full(max(n - 1, 0), (0 if upper else n - 1)); c = np.arange(1, n) if upper else np.arange(0, n - 1)
        A = sps.coo_matrix((0.01 * rng.random(len(c)), (r, c)), shape=(n, n))
        M = (A + sps.diags(2.0 + rng.random(n))).tocsr()
    elif kind == "emptyish":  # most rows hold only their diagonal
        A = sps.random(n, n, density=min(1.0, 0.5 / max(n, 1)), format="csr", random_state=rng)
        S = sps.triu(A, 1) if upper else sps.tril(A, -1)
        M = (S + sps.diags(1.0 + rng.random(n))).tocsr()
    else:  # entries of the OTHER triangle (ignored) and a full matrix's diagonal
        A = sps.random(n, n, density=min(1.0, 6.0 / max(n, 1)), format="csr", random_state=rng)
        rowsum = np.asarray(abs(A).sum(axis=1)).ravel()
        M = (A + sps.diags(rowsum + 1.0)).tocsr()
    M.sum_duplicates()
    if unit:
        # an implicit unit diagonal: keep the system diagonally dominant (rows of the strict triangle scaled to sum 0.5 at
        # most), or the forward error against the oracle measures the conditioning of the matrix, not the solve (five such
        # cases in the first 600: backward error fine, forward error 1.2e-4 ... 3.1e-4 against the 1e-4 bound)
        S = (sps.triu(M, 1) if upper else sps.tril(M, -1)).tocsr()
        r = np.asarray(abs(S).sum(axis=1)).ravel()
        scale = np.where(r > 0.5, 0.5 / np.maximum(r, 1e-300), 1.0)
        other = (sps.tril(M, -1) if upper else sps.triu(M, 1))
        M = (sps.diags(scale) @ S + other + sps.diags(M.diagonal())).tocsr()
    b = rng.random(n) - 0.5
    mode = rng.choice(["coop", "coop", "levels", "smallgrid", "noinspect"])
    # (reproduction aids: FUZZ_MODE / FUZZ_DTYPE override the draw)
    mode = os.environ.get("FUZZ_MODE", mode)
    if os.environ.get("FUZZ_DTYPE"):
        dtype = np.float32 if os.environ["FUZZ_DTYPE"] == "f32" else np.float64
    if mode == "levels":
        os.environ["SPBLAS_GFX950_TRSV_COOP"] = "0"
    elif mode == "smallgrid":
        os.environ["SPBLAS_GFX950_TRSV_COOP_GRID"] = str(int(rng.choice([1, 3, 40])))
    if rng.random() < 0.3:
        os.environ["SPBLAS_GFX950_TRSV_NARROW"] = str(int(rng.choice([1, 16, 100000])))
    if rng.random() < 0.15:
        os.environ["SPBLAS_GFX950_TRSV_KAHN"] = "1"
    scale_a = float(rng.choice([2.0, -0.5])) if rng.random() < 0.2 else None
    try:
        T.check(M, b, upper, unit, dtype, scale_a=scale_a, inspect=(mode != "noinspect"))
    except AssertionError as e:
        bad += 1
        print(f"FAIL seed {seed0 + it}: n {n} {kind} upper {upper} unit {unit} {dtype.__name__} {mode} "
              f"{ {k: os.environ[k] for k in KNOBS if k in os.environ} }: {str(e)[:300]}", flush=True)
print(f"fuzz_sptrsv: {iters} cases, {bad} failures")
sys.exit(1 if bad else 0)
