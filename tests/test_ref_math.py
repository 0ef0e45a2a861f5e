"""The two float32 transcendentals of the step, restated in oracle/ref_math.h (CPU) and hlx_device.h (GPU), pinned
against the real thing: glibc's powf (what `np.float32 ** python_float` calls: physics_models.py:100, :324) and numpy's
own float32 exp kernel (what `np.exp(np.float32)` runs: physics_models.py:78,105,113, environment.py:1174-1180,1222).

Exhaustive over the argument ranges the step path can reach (and well beyond): T/T0 in (0.7, 1] with the ISA exponent,
altitude/10 in (1, 100] with 0.143, exp arguments in [-3400, 0)."""
import ctypes as C

import numpy as np
import pytest

ISA_EXPONENT = np.float32(9.80665 / (287.05 * 0.0065))


def _bits(x):
    return int(np.float32(x).view(np.uint32))


def test_restated_powf_equals_this_hosts_libm_on_the_reachable_ranges():
    import oracle.oracle as orc
    L = orc.lib()
    L.orc_check_powf.restype = C.c_long
    L.orc_check_powf.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.POINTER(C.c_uint32)]
    bad = C.c_uint32(0)
    # every float32 in [0.5, 1]: temperature ratios (the ISA troposphere gives (0.75, 1], domain randomisation a bit more)
    assert L.orc_check_powf(_bits(0.5), _bits(1.0), float(ISA_EXPONENT), C.byref(bad)) == 0, hex(bad.value)
    # every float32 in [1, 128]: altitude / 10 m below the boundary-layer height (default 1000 m)
    assert L.orc_check_powf(_bits(1.0), _bits(128.0), float(np.float32(0.143)), C.byref(bad)) == 0, hex(bad.value)


def test_numpy_scalar_power_is_libm_powf():
    """What pins the oracle's use of libm powf: numpy's float32 scalar power goes to the C library."""
    libm = C.CDLL("libm.so.6")
    libm.powf.restype, libm.powf.argtypes = C.c_float, [C.c_float, C.c_float]
    rng = np.random.default_rng(0)
    e = 9.80665 / (287.05 * 0.0065)
    for x in (0.7 + 0.3 * rng.random(20000)).astype(np.float32):
        assert np.float32(x ** e) == np.float32(libm.powf(float(x), float(np.float32(e))))


def test_restated_numpy_exp_equals_numpy_on_every_float_in_range():
    import oracle.oracle as orc
    L = orc.lib()
    L.orc_np_expf_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lo, hi = _bits(-1e-6), _bits(-3400.0)       # negative floats: bit patterns grow with magnitude; denormal results and the
    # underflow to exact zero included (a 10 km miss distance / 3 is the largest argument of environment.py:1180)
    bad = 0
    for start in range(lo, hi + 1, 1 << 25):
        x = np.arange(start, min(start + (1 << 25), hi + 1), dtype=np.uint32).view(np.float32)
        y = np.empty_like(x)
        L.orc_np_expf_batch(x.ctypes.data, y.ctypes.data, len(x))
        with np.errstate(all="ignore"):
            bad += int((np.exp(x).view(np.uint32) != y.view(np.uint32)).sum())
    assert bad == 0
    # scalars take the same kernel as arrays
    xs = (-np.random.default_rng(1).random(5000) * 30).astype(np.float32)
    assert all(np.exp(x) == np.exp(xs[i:i + 1])[0] for i, x in enumerate(xs))


@pytest.mark.gpu
def test_device_copies_equal_the_oracle_copies():
    """hlx_device.h pow_ref / exp_np (through hlx_selftest_math) against oracle/ref_math.h, bit for bit."""
    import torch
    import oracle.oracle as orc
    from hlynr_intercept_amd import _lib
    lib, L = _lib.load(), orc.lib()
    L.orc_np_expf_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.orc_ref_powf_batch.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_int64]
    dev = torch.device("cuda", 0)

    def gpu(kind, x, y=0.0):
        xd = torch.from_numpy(x).to(dev)
        out = torch.empty_like(xd)
        _lib.check(lib.hlx_selftest_math(kind, xd.data_ptr(), float(y), out.data_ptr(), xd.numel(), None))
        torch.cuda.synchronize()
        return out.cpu().numpy()

    for lo, hi, y in ((0.5, 1.0, ISA_EXPONENT), (1.0, 128.0, np.float32(0.143))):
        for start in range(_bits(lo), _bits(hi) + 1, 1 << 24):
            x = np.arange(start, min(start + (1 << 24), _bits(hi) + 1), dtype=np.uint32).view(np.float32)
            ref = np.empty_like(x)
            L.orc_ref_powf_batch(x.ctypes.data, float(y), ref.ctypes.data, len(x))
            assert np.array_equal(gpu(0, x, y).view(np.uint32), ref.view(np.uint32)), (lo, hi, start)
    lo, hi = _bits(-1e-6), _bits(-3400.0)       # through the denormal results down to exact zero
    for start in range(lo, hi + 1, 1 << 24):
        x = np.arange(start, min(start + (1 << 24), hi + 1), dtype=np.uint32).view(np.float32)
        ref = np.empty_like(x)
        L.orc_np_expf_batch(x.ctypes.data, ref.ctypes.data, len(x))
        assert np.array_equal(gpu(1, x).view(np.uint32), ref.view(np.uint32)), start


@pytest.mark.gpu
def test_short_square_roots_equal_the_compilers_own():
    """hlx_device.h sqrt_rn (float32) on EVERY float from 2^-100 up, on 0 and on +inf, and dsqrt_pos (float64) on 2^27 sums of
    products spread over the magnitudes the step forms, against the compiler's IEEE sqrtf / sqrt on the device, bit for bit
    (hlx_selftest_math kinds 2 and 3 answer 1.0 where the bits agree)."""
    import torch
    from hlynr_intercept_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)

    def agree(kind, xd, y=0.0):
        out = torch.empty_like(xd)
        _lib.check(lib.hlx_selftest_math(kind, xd.data_ptr(), float(y), out.data_ptr(), xd.numel(), None))
        return bool((out == 1.0).all().item())

    lo, hi = _bits(2.0 ** -100), 0x7F800000          # ... +inf included
    step = 1 << 26
    for start in range(lo, hi + 1, step):
        x = torch.arange(start, min(start + step, hi + 1), dtype=torch.int64, device=dev).to(torch.int32).view(torch.float32)
        assert agree(2, x), hex(start)
    assert agree(2, torch.zeros(64, device=dev))
    g = torch.Generator(device=dev).manual_seed(5)
    for scale in (1e-3, 1.0, 30.0, 3e2, 3e3, 1e5):
        x = (torch.rand(1 << 24, generator=g, device=dev) * 2 - 1) * scale
        for y in (0.0, 1e-3, 7.5, -0.3 * scale, 1e4):
            assert agree(3, x, y), (scale, y)
    assert agree(3, torch.zeros(64, device=dev), 0.0)
