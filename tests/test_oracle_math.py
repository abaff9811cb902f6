"""Oracle pinning, part 1: src/math.rs and src/elements/{angular,angular_int}.rs.

The reference holds no golden values for these (its tests draw from an unseeded RNG), so the
C oracle is pinned by (a) the reference's own property tests, restated here with their
file:line, and (b) a bit-for-bit diff against the independent Python restatement
oracle/pyref.py.
"""
import numpy as np
import pytest

from oracle import pyref
from tests.conftest import random_floats

EPS = float(np.finfo(np.float32).eps)
DIST_EPSILON = 10.0 * EPS  # src/elements/angular.rs:97


def test_dot_product_property(oracle):
    """src/math.rs:183-196 (math::tests::dot_product)."""
    rng = np.random.default_rng(1)
    for n in range(1, 101):
        x, y = random_floats(rng, n), random_floats(rng, n)
        expected = np.float32(0.0)
        for i in range(n):
            expected = np.float32(expected + np.float32(x[i] * y[i]))
        assert abs(float(expected) - oracle.dot_f32(x, y)) < 1e-6


@pytest.mark.parametrize("n", [0, 1, 3, 31, 32, 33, 63, 64, 65, 96, 100, 128, 200, 300])
def test_dot_f32_bit_exact_vs_pyref(oracle, n):
    rng = np.random.default_rng(100 + n)
    for _ in range(20):
        x, y = random_floats(rng, n), random_floats(rng, n)
        a = np.float32(oracle.dot_f32(x, y))
        b = pyref.dot_product_f32(x, y)
        assert a.tobytes() == np.float32(b).tobytes(), (n, a, b)


def test_dot_f32_association_is_32_lane_then_ordered_sum(oracle):
    """A hand-computed case where (chunked, ordered) differs from a plain left-to-right sum:
    1e8 in lane 0 of chunk 0, -1e8 in lane 0 of chunk 1 cancel inside accumulator 0 first."""
    x = np.zeros(64, np.float32)
    y = np.ones(64, np.float32)
    x[0], x[32], x[1] = 1e8, -1e8, 1.0
    assert oracle.dot_f32(x, y) == 1.0          # lane 0: fma(-1e8,1,1e8)=0 ; then 0 + 0 + 1
    plain = np.float32(0.0)
    for i in range(64):
        plain = np.float32(plain + x[i] * y[i])
    assert plain == 0.0                          # left-to-right loses the 1.0


def test_reference_dist(oracle):
    """src/elements/angular.rs:99-107."""
    rng = np.random.default_rng(2)
    for _ in range(100):
        x = oracle.normalize_f32(random_floats(rng, 100))
        y = oracle.normalize_f32(random_floats(rng, 100))
        assert abs(oracle.dist(x, y) - oracle.reference_dist_f32(x, y)) < DIST_EPSILON


def test_dist_between_same_vector(oracle):
    """src/elements/angular.rs:109-116."""
    rng = np.random.default_rng(3)
    for _ in range(100):
        x = oracle.normalize_f32(random_floats(rng, 100))
        assert oracle.dist(x, x) < DIST_EPSILON


def test_dist_between_opposite_vector(oracle):
    """src/elements/angular.rs:118-126."""
    rng = np.random.default_rng(4)
    for _ in range(100):
        x = oracle.normalize_f32(random_floats(rng, 100))
        y = oracle.normalize_f32(-x)
        assert oracle.dist(x, y) > 2.0 - DIST_EPSILON


def test_small_and_large_arrays(oracle):
    """src/elements/angular.rs:128-142 (test_array, test_large_arrays): must not panic."""
    a = oracle.normalize_f32(np.array([0, 1, 2], np.float32))
    assert oracle.dist(a, a) < DIST_EPSILON
    b = oracle.normalize_f32(np.ones(100, np.float32))
    assert oracle.dist(b, b) < DIST_EPSILON


def test_normalize_bit_exact_vs_pyref_and_zero(oracle):
    rng = np.random.default_rng(5)
    for n in [1, 3, 25, 100, 200]:
        x = random_floats(rng, n)
        assert oracle.normalize_f32(x).tobytes() == pyref.normalize_f32(x).tobytes()
    z = np.zeros(100, np.float32)
    assert (oracle.normalize_f32(z) == 0).all()  # norm == 0: untouched (src/math.rs:134)


def test_dist_f32_bit_exact_vs_pyref(oracle):
    rng = np.random.default_rng(6)
    for n in [3, 28, 100, 200]:
        for _ in range(20):
            x = oracle.normalize_f32(random_floats(rng, n))
            y = oracle.normalize_f32(random_floats(rng, n))
            assert np.float32(oracle.dist(x, y)).tobytes() == np.float32(pyref.dist_f32(x, y)).tobytes()


def test_dist_is_clamped_to_zero(oracle):
    x = np.full(100, 0.1, np.float32)  # deliberately NOT normalised to norm 1: dot > 1
    x = (x * np.float32(1.01)).astype(np.float32)
    assert oracle.dot_f32(x, x) > 1.0
    assert oracle.dist(x, x) == 0.0


def test_quantize_matches_pyref_and_edges(oracle):
    rng = np.random.default_rng(7)
    for n in [1, 32, 100]:
        for _ in range(20):
            x = random_floats(rng, n)
            assert (oracle.quantize(x) == pyref.quantize(x)).all()
    # truncation toward zero and +-127 at the max-abs component (angular_int.rs:37-41)
    q = oracle.quantize(np.array([1.0, -1.0, 0.999, -0.999, 0.5, 0.004], np.float32))
    assert q.tolist() == [127, -127, 126, -126, 63, 0]
    # all-zero input: 0*127/0 = NaN -> `as i8` gives 0
    assert oracle.quantize(np.zeros(8, np.float32)).tolist() == [0] * 8


def test_dist_i8_matches_pyref_and_zero_vector(oracle):
    rng = np.random.default_rng(8)
    for n in [32, 100, 200]:
        for _ in range(50):
            x = oracle.quantize(random_floats(rng, n))
            y = oracle.quantize(random_floats(rng, n))
            assert np.float32(oracle.dist(x, y)).tobytes() == np.float32(pyref.dist_i8(x, y)).tobytes()
            r, dx, dy = oracle.dot_i8(x, y)
            assert r == int(np.dot(x.astype(np.int64), y.astype(np.int64)))
            assert dx == int(np.dot(x.astype(np.int64), x.astype(np.int64)))
            assert dy == int(np.dot(y.astype(np.int64), y.astype(np.int64)))
    z = np.zeros(100, np.int8)
    y = oracle.quantize(random_floats(rng, 100))
    assert oracle.dist(z, y) == 1.0  # 0/0 -> NaN -> 0 (angular_int.rs:55)
    assert oracle.dist(y, y) <= 10 * EPS


def test_int8_extremes_are_exact(oracle):
    x = np.full(200, -128, np.int8)
    y = np.full(200, 127, np.int8)
    r, dx, dy = oracle.dot_i8(x, y)
    assert (r, dx, dy) == (-128 * 127 * 200, 128 * 128 * 200, 127 * 127 * 200)
    assert abs(oracle.dist(x, y) - 2.0) < 1e-6
