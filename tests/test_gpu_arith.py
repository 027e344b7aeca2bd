"""The kernels replace the generic IEEE division expansion by rcp + fma corrections (Markstein) wherever the operands
are ordinary normal numbers (mdvt_device.h).  The decree asks for the correctly rounded result,
so the substitution must give the same bits: checked on the device against the compiler's IEEE expansion for EVERY
f32 divisor in [2^-32, 2^32) (reciprocal) and for 16 numerators per divisor (division)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from metric_depth_video_toolbox_amd import _lib
    c = _lib.Context(0, 64, 48)
    yield c
    c.close()


def test_reciprocal_is_correctly_rounded_for_every_operand(ctx):
    n = C.c_uint64(123)
    ctx.check(ctx._L.mdvt_selftest(ctx.handle, 0, 0, C.byref(n)))
    assert n.value == 0, f"{n.value} of 2^29 reciprocals differ from the IEEE expansion"


@pytest.mark.parametrize("seed", [1, 20260928])
def test_division_is_correctly_rounded(ctx, seed):
    n = C.c_uint64(123)
    ctx.check(ctx._L.mdvt_selftest(ctx.handle, 1, seed, C.byref(n)))
    assert n.value == 0, f"{n.value} of 16 x 2^29 quotients differ from the IEEE expansion"


def test_u8_conversion_is_the_decrees_rint_and_clamp_for_every_float(ctx):
    n = C.c_uint64(123)
    ctx.check(ctx._L.mdvt_selftest(ctx.handle, 2, 0, C.byref(n)))
    assert n.value == 0, f"v_cvt_pk_u8_f32 differs from rint/clamp on {n.value} inputs"


def test_selftest_rejects_bad_arguments(ctx):
    n = C.c_uint64(0)
    assert ctx._L.mdvt_selftest(ctx.handle, 7, 0, C.byref(n)) < 0
    assert ctx._L.mdvt_selftest(ctx.handle, 0, 0, None) < 0
