"""CPU tests of the host-side mirror: view inspection (detail/view_inspectors.hpp) and the
error behaviour that must hold before any device work is issued."""
import numpy as np
import pytest
import torch

import spblas_reference_amd as sp


def _csr(m=4, n=5, device="cpu"):
    rowptr = torch.tensor([0, 2, 2, 3, 5], dtype=torch.int32, device=device)
    colind = torch.tensor([4, 1, 0, 3, 3], dtype=torch.int32, device=device)
    values = torch.tensor([1., 2., 3., 4., 5.], dtype=torch.float32, device=device)
    return sp.csr_view(values, rowptr, colind, (m, n), 5)


def test_view_accessors_and_update():
    a = _csr()
    assert a.shape() == (4, 5) and a.size() == 5
    assert a.values().numel() == 5 and a.rowptr().numel() == 5 and a.colind().numel() == 5
    a.update(a.values(), a.rowptr(), a.colind(), (4, 5), 5)
    assert a.shape()[0] == 4


def test_scaling_factor_is_product_of_all_factors():
    # detail/view_inspectors.hpp:22-77
    a = _csr()
    x = torch.ones(5)
    assert sp.get_scaling_factor(a) is None
    assert sp.get_scaling_factor(sp.scaled(2.0, a)) == 2.0
    assert sp.get_scaling_factor(sp.scaled(3.0, sp.scaled(2.0, a))) == 6.0
    assert sp.get_scaling_factor(sp.scaled(2.0, a), sp.scaled(-5, x)) == -10.0
    assert sp.get_scaling_factor(a, sp.scaled(4, x)) == 4
    assert sp.get_ultimate_base(sp.scaled(2.0, sp.matrix_opt(a))) is a
    assert sp.has_matrix_opt(sp.scaled(2.0, sp.matrix_opt(a))) and not sp.has_matrix_opt(sp.scaled(2.0, a))


def test_conjugation_parity():
    # view_inspectors.hpp:81-97: odd number of conjugated views
    a = _csr()
    assert not sp.is_conjugated(a)
    assert sp.is_conjugated(sp.conjugated(a))
    assert not sp.is_conjugated(sp.conjugated(sp.conjugated(a)))
    assert sp.is_conjugated(sp.scaled(2.0, sp.conjugated(a)))


def test_transposed_relabels_without_copy():
    a = _csr()
    t = sp.transposed(a)
    assert isinstance(t, sp.csc_view) and t.shape() == (5, 4)
    assert t.values().data_ptr() == a.values().data_ptr()
    assert isinstance(sp.transposed(t), sp.csr_view)


def test_shape_mismatch_raises_invalid_argument_equivalent():
    # algorithms/multiply_impl.hpp:37-41 (checked before any device work)
    a = _csr()
    with pytest.raises(ValueError):
        sp.multiply(a, torch.ones(4), torch.zeros(4))
    with pytest.raises(ValueError):
        sp.multiply(a, torch.ones(5), torch.zeros(3))
    with pytest.raises(ValueError):
        sp.multiply(a, torch.ones(6, 2), torch.zeros(4, 2))


def test_conjugated_views_are_rejected():
    # vendor/rocsparse/detail/spmv_impl.hpp:29-33
    a = _csr()
    with pytest.raises(RuntimeError, match="conjugated"):
        sp.multiply(sp.conjugated(a), torch.ones(5), torch.zeros(4))


def test_host_memory_fails_loudly_no_cpu_fallback():
    a = _csr()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sp.multiply(a, torch.ones(5), torch.zeros(4))


def test_operation_info_and_state_objects():
    info = sp.operation_info_t((3, 4), 7)
    assert info.result_shape() == (3, 4) and info.result_nnz() == 7
    info.update_impl_((5, 6), 9)
    assert info.result_shape() == (5, 6) and info.result_nnz() == 9
    st = sp.spgemm_state_t()
    assert st.result_nnz() == 0 and st.result_shape() == (0, 0)


def test_generate_csr_has_unsorted_unique_columns():
    from spblas_reference_amd import generate
    values, rowptr, colind, shape, nnz = generate.generate_csr(100, 1000, 10000)
    assert rowptr[-1] == nnz == 10000 and shape == (100, 1000)
    unsorted = 0
    for r in range(100):
        cols = colind[rowptr[r]:rowptr[r + 1]]
        assert len(set(cols.tolist())) == len(cols)
        unsorted += int(np.any(np.diff(cols) < 0))
    assert unsorted > 50


def test_scale_and_mixed_format_spgemm_host_checks():
    """Host-side behaviour of the widenings (no device needed): scale() on host memory fails loudly like every
    other entry point; SpGEMM accepts csc_view operands/results, checks the shapes of the LOGICAL matrices
    before any device work and rejects other operand types."""
    a = _csr()                                   # 4 x 5
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sp.scale(2.0, a)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sp.scale(2.0, torch.ones(7))
    a_csc = sp.transposed(sp.csr_view(a.values(), a.rowptr(), a.colind(), (4, 5), a.size()))  # 5 x 4 as CSC
    c_csc = sp.csc_view(None, torch.zeros(5, dtype=torch.int32), None, (4, 4), 0)
    with pytest.raises(ValueError):              # (4x5) * (4x5): inner dimensions differ
        sp.multiply_compute(a, a, c_csc)
    with pytest.raises(ValueError):              # (4x5) * (5x4) -> 4x4, but C says 4x3
        sp.multiply_compute(a, a_csc, sp.csr_view(None, torch.zeros(5, dtype=torch.int32), None, (4, 3), 0))
    with pytest.raises(NotImplementedError):
        sp.multiply_compute(a, a_csc, torch.zeros(4, 4))
    c_csc.update(torch.ones(3), torch.zeros(5, dtype=torch.int32), torch.zeros(3, dtype=torch.int32), (4, 4), 3)
    assert c_csc.size() == 3 and c_csc.values().numel() == 3
