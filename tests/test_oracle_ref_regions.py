"""Pins the oracle's region geometry to the reference's OWN RectangularRegions.h, compiled from
/root/reference into oracle/_ref/libref_regions.so (recipe: oracle/Makefile).  The .so travels
to the GPU box with the repo snapshot; /root/reference itself is never read at test time."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

SO = os.path.join(O.ORACLE_DIR, "_ref", "libref_regions.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref not built (needs /root/reference)")

i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(SO)
    L.ref_input_region_offset.restype = C.c_int
    L.ref_input_region_offset.argtypes = [i64p, i64p, i64p, i64p, i64p]
    L.ref_output_region_points.restype = C.c_long
    L.ref_output_region_points.argtypes = [i64p, i64p, i64p, i64p, i64p, C.c_long]
    L.ref_region_points.restype = C.c_long
    L.ref_region_points.argtypes = [i64p, i64p, i64p, i32p, C.c_long]
    return L


GEOMS = [((2, 2, 2), (2, 2, 2)), ((3, 3, 3), (2, 2, 2)), ((1, 1, 4), (1, 1, 1)), ((3, 3, 3), (1, 1, 1)),
         ((2, 3, 4), (2, 1, 3)), ((4, 4, 4), (2, 2, 2))]


@pytest.mark.parametrize("size,stride", GEOMS)
def test_output_region_and_offsets(ref, size, stride):
    rng = np.random.default_rng(0)
    size = np.array(size, np.int64)
    stride = np.array(stride, np.int64)
    out_sp = np.array([7, 6, 5], np.int64)
    L = O.lib()
    for _ in range(200):
        p = rng.integers(0, 16, 3).astype(np.int64)
        rp = np.zeros((64, 3), np.int64)
        n_ref = ref.ref_output_region_points(p, size.copy(), stride.copy(), out_sp.copy(), rp, 64)
        lb, ub = np.zeros(3, np.int64), np.zeros(3, np.int64)
        L.oracle_output_region(p, size, stride, out_sp, lb, ub)
        op = np.zeros((64, 3), np.int64)
        n_or = L.oracle_region_points(lb, ub, op)
        assert n_or == n_ref
        np.testing.assert_array_equal(op[:n_or], rp[:n_ref])
        for j in rp[:n_ref]:
            lbub = np.zeros(6, np.int64)
            off_ref = ref.ref_input_region_offset(np.ascontiguousarray(j), p, size.copy(), stride.copy(), lbub)
            ilb, iub = np.zeros(3, np.int64), np.zeros(3, np.int64)
            L.oracle_input_region(np.ascontiguousarray(j), size, stride, ilb, iub)
            np.testing.assert_array_equal(np.concatenate([ilb, iub]), lbub)
            assert L.oracle_region_offset(p, ilb, iub) == off_ref
            assert 0 <= off_ref < size.prod()


def test_region_iteration_order_is_last_dim_fastest(ref):
    lb = np.array([-1, 2, 5], np.int64)
    ub = np.array([1, 3, 7], np.int64)
    rp = np.zeros((32, 3), np.int64)
    ro = np.zeros(32, np.int32)
    n = ref.ref_region_points(lb, ub, rp, ro, 32)
    assert n == 18
    assert (ro[:n] == np.arange(n)).all()  # offset() enumerates in iteration order
    op = np.zeros((32, 3), np.int64)
    assert O.lib().oracle_region_points(lb, ub, op) == n
    np.testing.assert_array_equal(op[:n], rp[:n])
    # submanifold region = [p - size/2, p + size - 1 - size/2]
    slb, sub = np.zeros(3, np.int64), np.zeros(3, np.int64)
    O.lib().oracle_submanifold_region(np.array([0, 3, 6], np.int64), np.array([3, 2, 3], np.int64), slb, sub)
    np.testing.assert_array_equal(slb, [-1, 2, 5])
    np.testing.assert_array_equal(sub, [1, 3, 7])
