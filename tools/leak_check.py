"""Create / use / destroy every kind of plan and state many times; the free device memory must come back."""
import gc, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp
from spblas_reference_amd import generate, _capi
dev = torch.device("cuda:0")
def free_mb():
    torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0] / 2**20
v, rp, ci, shape, nnz = generate.uniform_csr_device(1_000_000, 1_000_000, 10, seed=0, device=dev)
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(shape[1], device=dev); y = torch.empty(shape[0], device=dev)
rv, rrp, rci, rshape, rnnz = generate.rmat_csr_device(20, 16, dtype=torch.float64, device=dev)
ra = sp.csr_view(rv, rrp, rci, rshape, rnnz)
rx = torch.rand(rshape[1], dtype=torch.float64, device=dev); ry = torch.empty(rshape[0], dtype=torch.float64, device=dev)
sv, srp, sci, sshape, snnz = generate.uniform_csr_device(100_000, 100_000, 8, seed=1, device=dev)
sa = sp.csr_view(sv, srp, sci, sshape, snnz)
B = torch.rand(shape[1] // 10, 32, device=dev)
def cycle():
    for alg in (_capi.SPMV_SLICED, _capi.SPMV_ROWBLOCK, _capi.SPMV_AUTO):
        info = sp.multiply_inspect(sp.matrix_opt(a), x, y, alg=alg); sp.multiply(info, a, x, y); del info
        info = sp.multiply_inspect(sp.matrix_opt(ra), rx, ry, alg=alg); sp.multiply(info, ra, rx, ry); del info
    crp = torch.zeros(sshape[0] + 1, dtype=torch.int32, device=dev)
    c = sp.csr_view(None, crp, None, sshape, 0)
    st = sp.spgemm_state_t(); sp.multiply_compute(st, sa, sa, c); n = st.result_nnz()
    c.update(torch.empty(n, device=dev), crp, torch.empty(n, dtype=torch.int32, device=dev), sshape, n)
    for _ in range(3): sp.multiply_fill(st, sa, sa, c)
    del st, c
    t = sp.csr_view(torch.empty(snnz, device=dev), torch.empty(sshape[1] + 1, dtype=torch.int32, device=dev),
                    torch.empty(snnz, dtype=torch.int32, device=dev), (sshape[1], sshape[0]), snnz)
    sp.transpose(sa, t)
    bb = torch.rand(sshape[0], device=dev); xx = torch.empty(sshape[0], device=dev)
    info = sp.triangular_solve_inspect(sa, sp.lower_triangle, sp.implicit_unit_diagonal, bb, xx)
    sp.triangular_solve(info, sa, sp.lower_triangle, sp.implicit_unit_diagonal, bb, xx); del info
cycle(); base = free_mb()
for i in range(30):
    cycle()
end = free_mb()
print(f"free device memory: {base:.0f} MiB after the first cycle, {end:.0f} MiB after 30 more (delta {base - end:+.0f} MiB)")
sys.exit(0 if base - end < 64 else 1)
