"""The 18 Vector3 unit tests of the reference (src/math.rs:246-418), restated
against the oracle's Vec3.  These are the only known-answer tests the reference
holds for this path; they pin the oracle's arithmetic layer."""
import numpy as np
import pytest

# orc.vec3 ops: 0 add 1 sub 2 mul_s 3 mul_v 4 div_s 5 neg 6 dot 7 cross 8 length 9 normalize
# 10 normal_from_triangle 11 reflect 12 refract 13 face_forward 14 max 15 luminance 16 div_v 17 length_squared
P64, P32 = 64, 32


@pytest.mark.parametrize("prec", [P64, P32])
def test_vector3_creation_add_sub(orc, prec):                     # math.rs:251,259,267
    _, v = orc.vec3(0, [1, 2, 3], [0, 0, 0], precision=prec)
    assert list(v) == [1.0, 2.0, 3.0]
    _, v = orc.vec3(0, [1, 2, 3], [4, 5, 6], precision=prec)
    assert list(v) == [5.0, 7.0, 9.0]
    _, v = orc.vec3(1, [4, 5, 6], [1, 2, 3], precision=prec)
    assert list(v) == [3.0, 3.0, 3.0]


@pytest.mark.parametrize("prec", [P64, P32])
def test_vector3_mul_div_neg(orc, prec):                          # math.rs:275,282,289,297,304
    _, v = orc.vec3(2, [1, 2, 3], s=2.0, precision=prec)
    assert list(v) == [2.0, 4.0, 6.0]
    _, v = orc.vec3(3, [1, 2, 3], [2, 3, 4], precision=prec)
    assert list(v) == [2.0, 6.0, 12.0]
    _, v = orc.vec3(4, [2, 4, 6], s=2.0, precision=prec)
    assert list(v) == [1.0, 2.0, 3.0]
    _, v = orc.vec3(5, [1, -2, 3], precision=prec)
    assert list(v) == [-1.0, 2.0, -3.0]
    _, v = orc.vec3(16, [2, 6, 12], [2, 3, 4], precision=prec)
    assert list(v) == [1.0, 2.0, 3.0]


@pytest.mark.parametrize("prec", [P64, P32])
def test_dot_cross_length(orc, prec):                             # math.rs:311,319,327
    r, _ = orc.vec3(6, [1, 2, 3], [4, 5, 6], precision=prec)
    assert r == 32.0
    _, v = orc.vec3(7, [1, 0, 0], [0, 1, 0], precision=prec)
    assert list(v) == [0.0, 0.0, 1.0]
    r, _ = orc.vec3(8, [3, 4, 0], precision=prec)
    assert r == 5.0
    r, _ = orc.vec3(17, [3, 4, 0], precision=prec)
    assert r == 25.0


def test_normalize_f64_exact(orc):                                # math.rs:333,341 (assert_eq on 0.6, 0.8)
    _, v = orc.vec3(9, [3, 4, 0], precision=P64)
    assert list(v) == [0.6, 0.8, 0.0]
    assert abs(np.linalg.norm(v) - 1.0) < 1e-10


def test_normalize_f32(orc):
    # f32 mode multiplies by the IEEE reciprocal (documented deviation of the f32 arithmetic mode)
    _, v = orc.vec3(9, [3, 4, 0], precision=P32)
    assert np.allclose(v, [0.6, 0.8, 0.0], rtol=0, atol=2e-7)


def test_normalize_zero_returns_self(orc):                        # math.rs:48-51
    _, v = orc.vec3(9, [0, 0, 0], precision=P64)
    assert list(v) == [0.0, 0.0, 0.0]


@pytest.mark.parametrize("prec,tol", [(P64, 1e-10), (P32, 1e-6)])
def test_normal_from_triangle(orc, prec, tol):                    # math.rs:349
    _, n = orc.vec3(10, [0, 0, 0], [1, 0, 0], [0, 1, 0], precision=prec)
    assert abs(n[0]) < tol and abs(n[1]) < tol and abs(n[2] - 1.0) < tol
    assert abs(np.linalg.norm(n) - 1.0) < tol


@pytest.mark.parametrize("prec,tol", [(P64, 1e-10), (P32, 1e-6)])
def test_reflect(orc, prec, tol):                                 # math.rs:364
    inc = np.array([1.0, -1.0, 0.0]) / np.sqrt(2.0)
    _, r = orc.vec3(11, inc, [0, 1, 0], precision=prec)
    exp = np.array([1.0, 1.0, 0.0]) / np.sqrt(2.0)
    assert np.all(np.abs(r - exp) < tol)


@pytest.mark.parametrize("prec", [P64, P32])
def test_refract_normal_incidence(orc, prec):                     # math.rs:378
    ok, r = orc.vec3(12, [0, -1, 0], [0, 1, 0], s=1.0 / 1.5, precision=prec)
    assert ok == 1.0
    assert abs(r[0]) < 1e-10 and r[1] < 0.0


@pytest.mark.parametrize("prec", [P64, P32])
def test_refract_total_internal_reflection(orc, prec):            # math.rs:393
    inc = np.array([0.8, -0.6, 0.0])
    inc /= np.linalg.norm(inc)
    ok, _ = orc.vec3(12, inc, [0, 1, 0], s=1.5, precision=prec)
    assert ok == 0.0


@pytest.mark.parametrize("prec", [P64, P32])
def test_face_forward(orc, prec):                                 # math.rs:405
    _, f = orc.vec3(13, [0, 1, 0], [0, -1, 0], precision=prec)
    assert list(f) == [0.0, 1.0, 0.0]
    _, f = orc.vec3(13, [0, 1, 0], [0, 1, 0], precision=prec)
    assert list(f) == [-0.0, -1.0, -0.0] or list(f) == [0.0, -1.0, 0.0]


def test_max_and_luminance(orc):                                  # math.rs:128,133
    r, _ = orc.vec3(14, [0.2, 0.9, 0.4])
    assert r == 0.9
    r, _ = orc.vec3(15, [1.0, 1.0, 1.0])
    assert abs(r - 1.0) < 1e-15
    r, _ = orc.vec3(15, [0.5, 0.25, 2.0])
    assert r == 0.2126 * 0.5 + 0.7152 * 0.25 + 0.0722 * 2.0
