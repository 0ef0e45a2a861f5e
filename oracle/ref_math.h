/* ref_math.h -- restatements of the two float32 transcendentals the reference's step really executes.
 *
 * TEST INFRASTRUCTURE (oracle/): the product kernel carries its own copy of these formulas in
 * hlynr_intercept_amd/csrc/hlx_device.h; tests/test_ref_math.py pins BOTH against the real thing.
 *
 * (1) `np.float32 ** python_float` (physics_models.py:100 barometric formula, :324 wind profile) is numpy's scalar
 *     power = the host libm's powf.  glibc >= 2.27 (this image: 2.35) ships the ARM optimized-routines algorithm
 *     (sysdeps/ieee754/flt-32/e_powf.c, e_powf_log2_data.c, e_exp2f_data.c): log2 through a 16-entry table and a
 *     degree-5 polynomial, exp2 through a 32-entry table and a degree-3 polynomial, all in double.  Restated below
 *     for finite positive normal x and results inside the normal range (every call on the step path: x in (0.7, 1]
 *     with the ISA exponent, x in (1, 100] with 0.143).  The double result is rounded to float once; whether the
 *     multiply-adds are fused or not (glibc's x86-64 ifunc picks an FMA build on FMA hardware) changes the double in
 *     its last bits only: tests/test_ref_math.py finds ZERO differences from this container's powf over all
 *     8 388 609 floats in [0.5, 1] and all 67 108 866 floats in [1, 128] with either choice.
 * (2) `np.exp(np.float32)` (physics_models.py:78,105,113 ISA layers above 11 km; environment.py:1174-1180,1222
 *     precision-mode reward) is NOT libm: numpy >= 1.17 evaluates float32 exp with its own SIMD kernel
 *     (numpy/_core/src/umath/loops_exponent_log.dispatch.c.src: Cody-Waite reduction by ln2 in two float32 pieces,
 *     a degree-5 / degree-2 rational in float32 FMA arithmetic, IEEE division, scale by 2^k) for arrays and scalars
 *     alike; it differs from glibc's expf in 39 % of arguments (by one ulp).  Restated below; identical to np.exp of
 *     numpy 2.2.6 on all 219 793 476 floats in [-80, -1e-6].
 */
#ifndef HLX_REF_MATH_H
#define HLX_REF_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static const double REF_LOG2_TAB[16][2] = {   /* __powf_log2_data.tab: {1/c, log2(c)} */
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
static const double REF_LOG2_POLY[5] = {0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2,
                                        -0x1.7154748bef6c8p-1, 0x1.71547652ab82bp0};
static const uint64_t REF_EXP2_TAB[32] = {   /* __exp2f_data.tab: bits(2^(i/32)) - (i << 47) */
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
static const double REF_EXP2_POLY[3] = {0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1};

static inline float ref_powf(float x, float y) {   /* glibc powf, x > 0 finite normal, no overflow / underflow */
    uint32_t ix;
    memcpy(&ix, &x, 4);
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) % 16u);
    const uint32_t top = tmp & 0xff800000u, iz = ix - top;
    const int k = (int32_t)top >> 23;
    float zf;
    memcpy(&zf, &iz, 4);
    const double r = (double)zf * REF_LOG2_TAB[i][0] - 1.0, y0 = REF_LOG2_TAB[i][1] + (double)k;
    const double r2 = r * r;
    double l = REF_LOG2_POLY[0] * r + REF_LOG2_POLY[1];
    const double p = REF_LOG2_POLY[2] * r + REF_LOG2_POLY[3];
    const double r4 = r2 * r2;
    double q = REF_LOG2_POLY[4] * r + y0;
    q = p * r2 + q;
    l = l * r4 + q;                                   /* log2(x) */
    const double ylogx = (double)y * l;
    const double SHIFT = 0x1.8p+52 / 32.0;
    double kd = ylogx + SHIFT;
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    const uint64_t t = REF_EXP2_TAB[ki % 32u] + (ki << 47);
    double s;
    memcpy(&s, &t, 8);
    const double z = REF_EXP2_POLY[0] * rr + REF_EXP2_POLY[1];
    double e = REF_EXP2_POLY[2] * rr + 1.0;
    e = z * (rr * rr) + e;
    return (float)(e * s);
}

static inline float ref_np_expf(float x) {   /* numpy's float32 exp kernel, -103.97 < x < 88.72 */
    const float log2e = 1.44269504088896340736f, c1 = -6.93145752e-1f, c2 = -1.42860677e-6f, magic = 0x1.8p23f;
    const float p0 = 9.999999999980870924916e-01f, p1 = 7.257664613233124478488e-01f, p2 = 2.473615434895520810817e-01f,
                p3 = 5.114512081637298353406e-02f, p4 = 6.757896990527504603057e-03f, p5 = 5.082762527590693718096e-04f;
    const float q0 = 1.0f, q1 = -2.742335390411667452936e-01f, q2 = 2.159509375685829852307e-02f;
    volatile float qv = x * log2e;       /* round to nearest integer with the 1.5 * 2^23 trick (must not be folded) */
    qv = qv + magic;
    const float quad = qv - magic;
    x = fmaf(quad, c1, x);
    x = fmaf(quad, c2, x);
    float num = fmaf(p5, x, p4);
    num = fmaf(num, x, p3);
    num = fmaf(num, x, p2);
    num = fmaf(num, x, p1);
    num = fmaf(num, x, p0);
    float den = fmaf(q2, x, q1);
    den = fmaf(den, x, q0);
    return ldexpf(num / den, (int)quad);
}
#endif
