// hlx_device.h -- device-side building blocks of the fused intercept-environment step (gfx950).
//
// One lane = one environment, everything register resident.  No MFMA: the path is per-environment
// physics, not a contraction.
//
// ARITHMETIC DISCIPLINE (this translation unit is compiled with -ffp-contract=off):
// the reference (RomanSlack/Hlynr_Intercept, numpy >= 2) evaluates the step in a fixed mix of
// float32 and float64.  Per-step rewards are differences of two ~km distances, so a 1-ulp
// difference in a position shows up as a 1e-4 relative error in the reward; to stay inside the
// 1e-5 bar the integrator chain is restated operation by operation:
//   * float32 where the reference is float32: plain IEEE +,-,*,/ and sqrt, NO fused multiply-add;
//   * np.dot / np.linalg.norm of float32 3-vectors = float32 products accumulated in float64 and
//     rounded once (OpenBLAS sdot) -> sdot3()/snorm3();
//   * float64 where the reference is float64 (simple-wind state, missile acceleration, Kalman
//     state after a ground-radar measurement): MI355X runs fp64 vector math at half the fp32 rate,
//     so this costs a few dozen instructions, not a redesign;
//   * x / c for a constant c: (float)((double)x * (1.0 / c)) -- equals the correctly rounded
//     float32 quotient except with probability ~1e-8 per operation, at a third of the cost of an
//     IEEE float32 division.
// Pure outputs (the 26-D observation formulas) use ordinary fast float32.
// oracle/hlx_oracle.c is the CPU restatement these functions are tested against; citations are to
// the reference's rl_system/ files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEV __device__ __forceinline__

namespace hlx {

struct V3 {
    float x, y, z;
};
struct D3 {
    double x, y, z;
};
DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
DEV D3 d3(double x, double y, double z) { return D3{x, y, z}; }
DEV D3 to_d3(V3 a) { return D3{(double)a.x, (double)a.y, (double)a.z}; }
DEV V3 to_v3(D3 a) { return V3{(float)a.x, (float)a.y, (float)a.z}; }
DEV D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
DEV double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV double dsqrt_pos(double x);
DEV double dnorm(D3 a) { return dsqrt_pos(ddot(a, a)); }

// Correctly rounded float32 sqrt and division WITHOUT the range handling of the compiler's expansions.
// sqrt_rn: exact for x = 0, +inf and every x >= 2^-104 (measured: the last mismatch with sqrtf is 0x0b6e9372 = 4.6e-32).  The raw v_sqrt_f32 (1 ulp) plus the two neighbour-residual tests of
//   the compiler's own sqrtf sequence, minus its denormal pre/post-scaling (5 of 16 instructions).  Every argument on the
//   step path is a sum of squares of metre / second scale float32 values: exactly 0, or far above that.
// div_rn: exact whenever neither operand nor the quotient leaves the normal range and b != 0: the compiler's sequence
//   (reciprocal, one Newton step, quotient, two residual corrections) minus v_div_scale x2 / v_div_fixup and the VCC hazard
//   of v_div_fmas (3 of 11 instructions + wait states).  Used only where the operands are bounded.
// tools/micro/rn_check.hip compares both with sqrtf / operator/ (all 2^31 non-negative floats; 2^34 random pairs).
DEV float sqrt_rn(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
    const float rl = __builtin_fmaf(-lo, s, x), rh = __builtin_fmaf(-hi, s, x);
    float r = (0.f >= rl) ? lo : s;
    r = (0.f < rh) ? hi : r;
    return r;   // x = 0 and x = +inf need no special case: their neighbour residuals are NaN, both tests fail, r = s = x
}
// float64 square root of a sum of squares (0, or in the normal range): the compiler's own sequence (v_rsq_f64, one coupled
// Goldschmidt step, two residual corrections) without the scaling of arguments below 2^-767 -- 12 instead of 17
// instructions, bit-identical for every argument that needs no scaling (hlx_selftest_math kind 3).
DEV double dsqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return (x == 0.0 || x == (double)__builtin_inff()) ? x : g;
}
DEV float div_rn(float a, float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float y = __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
    const float q0 = a * y;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
    return __builtin_fmaf(__builtin_fmaf(-b, q1, a), y, q1);
}

#ifndef HLX_SHORT_SQRT
#define HLX_SHORT_SQRT 1
#endif
#if HLX_SHORT_SQRT
#define HLX_SQRTF(x) sqrt_rn(x)
#else
#define HLX_SQRTF(x) __builtin_sqrtf(x)
#endif

#ifndef HLX_SHORT_DIV
#define HLX_SHORT_DIV 1
#endif
#if HLX_SHORT_DIV
#define HLX_DIVF(a, b) div_rn((a), (b))
#else
#define HLX_DIVF(a, b) ((a) / (b))
#endif
// vector / norm-like scalar: every caller divides by a guarded norm (> 1e-6) or an O(1) physical scale
DEV V3 operator/(V3 a, float s) { return V3{HLX_DIVF(a.x, s), HLX_DIVF(a.y, s), HLX_DIVF(a.z, s)}; }

// numpy float32 dot / norm (OpenBLAS sdot): float32 products, float64 accumulation, one rounding
DEV float sdot3(V3 a, V3 b) { return (float)(((double)(a.x * b.x) + (double)(a.y * b.y)) + (double)(a.z * b.z)); }
DEV float snorm3(V3 a) { return HLX_SQRTF(sdot3(a, a)); }
DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
// x / c with c a constant whose float64 reciprocal is `inv`
DEV float divc(float x, double inv) { return (float)((double)x * inv); }
DEV V3 divc(V3 a, double inv) { return V3{divc(a.x, inv), divc(a.y, inv), divc(a.z, inv)}; }
// sin / cos of a small angle (|x| <= 0.35 rad: half of the largest per-step rotation, 20*sqrt(3) rad/s * dt)
// to ~1e-15, i.e. the float32 rounding of the result equals that of a correctly rounded libm call except
// with probability ~1e-7.  Taylor series in float64; contraction is fine here (only the value matters).
DEV double sin_small(double x) {
    const double x2 = x * x;
    double p = __builtin_fma(x2, -1.0 / 1307674368000.0, 1.0 / 6227020800.0);
    p = __builtin_fma(x2, p, -1.0 / 39916800.0);
    p = __builtin_fma(x2, p, 1.0 / 362880.0);
    p = __builtin_fma(x2, p, -1.0 / 5040.0);
    p = __builtin_fma(x2, p, 1.0 / 120.0);
    p = __builtin_fma(x2, p, -1.0 / 6.0);
    return __builtin_fma(x * x2, p, x);
}
DEV double cos_small(double x) {
    const double x2 = x * x;
    double p = __builtin_fma(x2, 1.0 / 20922789888000.0, -1.0 / 87178291200.0);
    p = __builtin_fma(x2, p, 1.0 / 479001600.0);
    p = __builtin_fma(x2, p, -1.0 / 3628800.0);
    p = __builtin_fma(x2, p, 1.0 / 40320.0);
    p = __builtin_fma(x2, p, -1.0 / 720.0);
    p = __builtin_fma(x2, p, 1.0 / 24.0);
    p = __builtin_fma(x2, p, -0.5);
    return __builtin_fma(x2, p, 1.0);
}
// fast (non-mirrored) helpers for pure outputs and for detection decisions (a threshold compare tolerates an
// ulp: it can only matter for an input that sits exactly on the boundary)
DEV float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
// atan on [-1,1], Abramowitz & Stegun 4.4.49 (|err| <= 2e-8 in exact arithmetic, ~1e-7 in float32)
DEV float atan_unit(float a) {
    const float s = a * a;
    float p = __builtin_fmaf(s, 0.0028662257f, -0.0161657367f);
    p = __builtin_fmaf(s, p, 0.0429096138f);
    p = __builtin_fmaf(s, p, -0.0752896400f);
    p = __builtin_fmaf(s, p, 0.1065626393f);
    p = __builtin_fmaf(s, p, -0.1420889944f);
    p = __builtin_fmaf(s, p, 0.1999355085f);
    p = __builtin_fmaf(s, p, -0.3333314528f);
    return __builtin_fmaf(a * s, p, a);
}
DEV float fast_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float r = (mx > 0.f) ? atan_unit(mn * __builtin_amdgcn_rcpf(mx)) : 0.f;
    r = (ay > ax) ? 1.5707963267948966f - r : r;
    r = (x < 0.f) ? 3.141592653589793f - r : r;
    return copysignf(r, y);
}
// asin on [-1,1], Abramowitz & Stegun 4.4.46 (|err| <= 2e-8 exact, ~3e-7 float32)
DEV float fast_asin(float x) {
    const float ax = fminf(fabsf(x), 1.0f);
    float p = __builtin_fmaf(ax, -0.0012624911f, 0.0066700901f);
    p = __builtin_fmaf(ax, p, -0.0170881256f);
    p = __builtin_fmaf(ax, p, 0.0308918810f);
    p = __builtin_fmaf(ax, p, -0.0501743046f);
    p = __builtin_fmaf(ax, p, 0.0889789874f);
    p = __builtin_fmaf(ax, p, -0.2145988016f);
    p = __builtin_fmaf(ax, p, 1.5707963050f);
    return copysignf(1.5707963267948966f - HLX_SQRTF(1.0f - ax) * p, x);
}
DEV float fdot(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
DEV float fnorm(V3 a) { return HLX_SQRTF(fdot(a, a)); }
// for pure outputs (observation entries, data-link / fusion scores): the raw 1-ulp v_sqrt_f32
DEV float fnorm_out(V3 a) { return __builtin_amdgcn_sqrtf(fdot(a, a)); }

// ---------------------------------------------------------------------------------------------
// Counter-based RNG: Philox4x32-10 keyed by the env-set seed; counter = (global env id, vec-step,
// stream).  Results therefore do not depend on sharding or launch geometry.
// ---------------------------------------------------------------------------------------------
DEV uint4 philox4x32_10(uint4 c, uint2 k) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) per word instead of separate mul_hi / mul_lo
        const uint64_t p0 = (uint64_t)M0 * (uint64_t)c.x, p1 = (uint64_t)M1 * (uint64_t)c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = uint4{hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0};
        k.x += W0;
        k.y += W1;
    }
    return c;
}
DEV float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }             // [0,1)
DEV float u01_open(uint32_t x) { return (float)((x >> 8) + 1u) * 5.9604644775390625e-08f; }  // (0,1]
// Box-Muller on two 32-bit words; v_sin/v_cos take revolutions, so 2*pi*u needs no range reduction.
// The radius uses the raw v_log_f32 (log2) and v_sqrt_f32: the argument is in [2^-24, 1], so none of the denormal / range
// handling of logf() and the correctly rounded sqrtf() (a dozen compares, selects and ldexps per call) can trigger, and a
// random variate needs no last-bit guarantee -- only determinism, which holds (the same function feeds hlx_fill_noise).
DEV void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01_open(a)));   // sqrt(-2 ln u)
    float u = u01(b);
    z0 = r * __builtin_amdgcn_cosf(u);
    z1 = r * __builtin_amdgcn_sinf(u);
}

struct Rng {
    uint2 key;
    uint32_t id_lo, id_hi, t_lo, t_hi;
    DEV uint4 raw(uint32_t stream) const { return philox4x32_10(uint4{id_lo, id_hi, t_lo, (t_hi << 8) | stream}, key); }
    DEV void normals4(uint32_t stream, float& a, float& b, float& c, float& d) const {
        uint4 x = raw(stream);
        box_muller(x.x, x.y, a, b);
        box_muller(x.z, x.w, c, d);
    }
};
// stream ids (RS_GUST uses two consecutive streams: 5 and 6)
enum : uint32_t {
    RS_STEP_U = 0, RS_STEP_N0 = 1, RS_STEP_N1 = 2, RS_STEP_N2 = 3, RS_GUST = 5,
    // step normals: N0 = evasion xyz + wind x, N1 = wind y z + ground pos x y, N2 = ground pos z + ground vel xyz
    RS_RESET_U0 = 8, RS_RESET_U1 = 9, RS_RESET_U2 = 10, RS_RESET_OBS_U = 11, RS_RESET_GPOS = 12, RS_RESET_GVEL = 13,
    RS_DR0 = 14, RS_DR1 = 15, RS_DR2 = 16, RS_DR3 = 17,
    // volley missile k = 1..3: evasion normals (stream RS_STEP_V1 + k - 1), spawn uniforms (RS_RESET_V1 + k - 1)
    RS_STEP_V1 = 18, RS_RESET_V1 = 21
};
// physics_models.py:382-384 gust direction N(0,1)^3 and magnitude Exp(1)
DEV void gust_draws(const Rng& rng, V3& g, float& e) {
    uint4 x = rng.raw(RS_GUST), y = rng.raw(RS_GUST + 1);
    float w_;
    box_muller(x.x, x.y, g.x, g.y);
    box_muller(x.z, x.w, g.z, w_);
    e = -0.6931471805599453f * __builtin_amdgcn_logf(u01_open(y.x));   // Exp(1) = -ln u
}

// ---------------------------------------------------------------------------------------------
// The two float32 transcendentals the reference's step really executes, restated (CPU twin: oracle/ref_math.h;
// tests/test_ref_math.py pins both, and hlx_selftest_math lets the GPU tests compare this copy bit for bit):
//   * `np.float32 ** python_float` (physics_models.py:100, :324) = the host libm's powf = glibc's table-driven
//     algorithm (e_powf.c: log2 by a 16-entry table + degree-5 polynomial, exp2 by a 32-entry table + degree-3
//     polynomial, all in double, one final rounding).  Fused or unfused multiply-adds give the same float on every
//     argument of the step path (ref_math.h), so the fused form is used here.
//   * `np.exp(np.float32)` (physics_models.py:78,105,113; environment.py:1174-1180,1222) = numpy's own float32 SIMD
//     kernel (Cody-Waite reduction, degree-5 / degree-2 rational in float32 FMA arithmetic, IEEE division).
// The tables (512 B) are staged in LDS by the step kernel: a per-lane table lookup from LDS costs ~100 cycles, from
// global memory ~700, and a wave that is alone on its SIMD eats every one of them.
// ---------------------------------------------------------------------------------------------
struct PowTab {
    double lt[16][2];                // __powf_log2_data.tab: {1/c, log2(c)}
    unsigned long long et[32];       // __exp2f_data.tab: bits(2^(i/32)) - (i << 47)
};
__device__ const PowTab HLX_POW_TAB = {
    {{0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
     {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
     {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
     {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
     {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
     {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
     {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
     {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}},
    {0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
     0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
     0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
     0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
     0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
     0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
     0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
     0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull}};
// `tab` = the table staged in LDS (or HLX_POW_TAB itself where latency does not matter)
DEV float pow_ref(float x, float y, const PowTab* tab) {   // x > 0 finite normal; result inside the normal range
    const uint32_t ix = __float_as_uint(x);
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const int k = (int32_t)top >> 23;
    const double2 e = *reinterpret_cast<const double2*>(tab->lt[i]);
    const double r = __builtin_fma((double)__uint_as_float(ix - top), e.x, -1.0), y0 = e.y + (double)k;
    const double r2 = r * r;
    double l = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    double q = __builtin_fma(0x1.71547652ab82bp0, r, y0);
    q = __builtin_fma(p, r2, q);
    l = __builtin_fma(l, r2 * r2, q);                                   // log2(x)
    const double ylogx = (double)y * l;
    const double SHIFT = 0x1.8p+52 / 32.0;
    double kd = ylogx + SHIFT;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    const double s = __longlong_as_double((long long)(tab->et[ki & 31ull] + (ki << 47)));
    const double z = __builtin_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    double ev = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    ev = __builtin_fma(z, rr * rr, ev);
    return (float)(ev * s);
}
DEV float exp_np(float x) {   // numpy's float32 exp kernel, -103.97 < x < 88.72
    const float magic = 0x1.8p23f;
    float qd = x * 1.44269504088896340736f;
    qd = qd + magic;
    asm volatile("" : "+v"(qd));              // the round-to-integer trick must survive the optimiser
    qd = qd - magic;
    x = __builtin_fmaf(qd, -6.93145752e-1f, x);
    x = __builtin_fmaf(qd, -1.42860677e-6f, x);
    float num = __builtin_fmaf(5.082762527590693718096e-04f, x, 6.757896990527504603057e-03f);
    num = __builtin_fmaf(num, x, 5.114512081637298353406e-02f);
    num = __builtin_fmaf(num, x, 2.473615434895520810817e-01f);
    num = __builtin_fmaf(num, x, 7.257664613233124478488e-01f);
    num = __builtin_fmaf(num, x, 9.999999999980870924916e-01f);
    float den = __builtin_fmaf(2.159509375685829852307e-02f, x, -2.742335390411667452936e-01f);
    den = __builtin_fmaf(den, x, 1.0f);
    return ldexpf(num / den, (int)qd);
}

// ---------------------------------------------------------------------------------------------
// physics_models.py:154-177  ISA atmosphere (float32 altitude) -> density, speed of sound
// ---------------------------------------------------------------------------------------------
DEV void atmosphere(float alt, float T0, float& rho, float& sos, const PowTab* tab) {
    constexpr float R = 287.05f, G = 9.80665f, L = 0.0065f;
    constexpr float EXPO = (float)(9.80665 / (287.05 * 0.0065));
    constexpr float GAMMA_R = (float)(1.4 * 287.05);
    // troposphere for every lane; the upper layers (not reachable in the shipped scenarios) override it out of line
    float T = T0 - L * alt;                       // :70-71
    float P = 101325.0f * pow_ref(HLX_DIVF(T, T0), EXPO, tab);   // :95-100
    if (__builtin_expect(alt > 11000.0f, 0)) {
        if (alt <= 20000.0f) {                    // :72-74,102-108
            T = 216.65f;
            P = 22632.0f * exp_np((-G * (alt - 11000.0f)) / (float)(287.05 * 216.65));
        } else {                                  // :76-78,110-113
            float ex = alt - 20000.0f;
            T = 216.65f * exp_np(-ex / 10000.0f);
            const float Pb = 5474.790039909648f;  // 22632 * exp(-g*9000/(R*216.65))  (:112)
            P = Pb * exp_np(-ex / 6000.0f);
        }
    }
    rho = HLX_DIVF(P, R * T);                     // :166
    sos = HLX_SQRTF(GAMMA_R * T);                     // :167-169
}

struct DragParams {
    // the reference holds base_cd / peak multiplier as Python floats: each use rounds a float64 expression ONCE to float32
    float subsonic, supersonic, mach_span, peak_m1 /* F(peak - 1.0) */, base_cd /* F(base_cd) */, cd_super /* F(base_cd * supersonic_multiplier) */;
};
// physics_models.py:236-264  drag force vector (float32 velocity); `area` = reference_area
DEV V3 mach_drag_force(V3 v, float rho, float sos, float area, const DragParams& p) {
    float vm = snorm3(v);
    float mach = HLX_DIVF(vm, sos);                                          // :233-234
    float cd = p.base_cd;                                                   // :207-209 subsonic; the other regimes out of line
    if (__builtin_expect(!(mach < p.subsonic), 0)) {
        if (mach < p.supersonic) {
            float frac = HLX_DIVF(mach - p.subsonic, p.mach_span);          // :213-214
            cd = p.base_cd * (1.0f + p.peak_m1 * frac);                     // :215-216 ((peak - 1.0) is a Python-float expression)
        } else cd = p.cd_super;                                             // :220 (python-float product)
    }
    float a = (((0.5f * rho) * (vm * vm)) * cd) * area;                     // :258
    V3 f = V3{HLX_DIVF(-v.x, vm) * a, HLX_DIVF(-v.y, vm) * a, HLX_DIVF(-v.z, vm) * a};   // :262-264
    if (__builtin_expect(vm < 1e-6f, 0)) f = v3(0.f, 0.f, 0.f);             // :238-239 (an override, not an early return: out of line)
    return f;
}
// the same with a float64 air-relative velocity (Mach model on, simple float64 wind): generic kernel only
DEV D3 mach_drag_force64(D3 v, float rho, float sos, double area, const DragParams& p) {
    double vm = dnorm(v);
    double mach = vm / (double)sos;
    double cd = p.base_cd;
    if (__builtin_expect(!(mach < (double)p.subsonic), 0)) {
        if (mach < (double)p.supersonic)
            cd = (double)p.base_cd * (1.0 + (double)p.peak_m1 * ((mach - (double)p.subsonic) / ((double)p.supersonic - (double)p.subsonic)));
        else cd = (double)p.cd_super;
    }
    double a = ((((double)(0.5f * rho)) * (vm * vm)) * cd) * area;
    D3 f = D3{(-v.x / vm) * a, (-v.y / vm) * a, (-v.z / vm) * a};
    if (__builtin_expect(vm < 1e-6, 0)) f = d3(0., 0., 0.);
    return f;
}
// environment.py:927-930 / 1111-1113 nan_to_num(nan=0, +-inf=+-lim), applied when any component is non-finite
DEV bool nan_hit(D3 a) {   // |x| + |y| + |z| is finite exactly when all three are (no cancellation between non-negative terms)
    return !isfinite((fabs(a.x) + fabs(a.y)) + fabs(a.z));
}
DEV D3 nan_guard(D3 a, double lim) {
    auto fix = [lim](double x) { return isnan(x) ? 0.0 : (isinf(x) ? (x > 0. ? lim : -lim) : x); };
    return D3{fix(a.x), fix(a.y), fix(a.z)};
}

// ---------------------------------------------------------------------------------------------
// quaternion helpers  core.py:1103-1205 ([w, x, y, z])
// ---------------------------------------------------------------------------------------------
struct Quat {
    float w, x, y, z;
};
DEV V3 forward_vec(Quat q) {   // core.py:1143-1152 (pure outputs and the fast half of the beam test: fast float32)
    V3 f = v3(2.f * (q.x * q.z + q.w * q.y), 2.f * (q.y * q.z - q.w * q.x), 1.f - 2.f * (q.x * q.x + q.y * q.y));
    return f * __builtin_amdgcn_rcpf(fnorm_out(f) + 1e-6f);
}
DEV V3 forward_vec_exact(Quat q) {   // the same, operation by operation as the reference: decides beam-edge cases (core.py:546-553)
    V3 f = v3(2.f * (q.x * q.z + q.w * q.y), 2.f * (q.y * q.z - q.w * q.x), 1.f - 2.f * (q.x * q.x + q.y * q.y));
    return f / (snorm3(f) + 1e-6f);
}
DEV V3 right_vec(Quat q) {     // core.py:1155-1164
    V3 f = v3(1.f - 2.f * (q.y * q.y + q.z * q.z), 2.f * (q.x * q.y + q.w * q.z), 2.f * (q.x * q.z - q.w * q.y));
    return f * __builtin_amdgcn_rcpf(fnorm_out(f) + 1e-6f);
}
DEV V3 up_vec(Quat q) {        // core.py:1167-1176
    V3 f = v3(2.f * (q.x * q.y - q.w * q.z), 1.f - 2.f * (q.x * q.x + q.z * q.z), 2.f * (q.y * q.z + q.w * q.x));
    return f * __builtin_amdgcn_rcpf(fnorm_out(f) + 1e-6f);
}
DEV Quat quat_mul(Quat a, Quat b) {  // environment.py:1322-1331 (left-to-right float32 sums)
    return Quat{((a.w * b.w - a.x * b.x) - a.y * b.y) - a.z * b.z, ((a.w * b.x + a.x * b.w) + a.y * b.z) - a.z * b.y,
                ((a.w * b.y - a.x * b.z) + a.y * b.w) + a.z * b.x, ((a.w * b.z + a.x * b.y) - a.y * b.x) + a.z * b.w};
}
// LOS orthonormal basis: environment.py:1006-1020 == core.py:824-834,939-945
DEV void los_basis(V3 lu, V3& h, V3& v) {
    V3 right = cross(lu, v3(0.f, 0.f, 1.f));
    float n = snorm3(right);
    h = (n > 1e-6f) ? right / n : v3(1.f, 0.f, 0.f);
    v = cross(lu, h);
}

// core.py:802-868 (los_frame: LOS rates obs[2:4], lead-angle cosine obs[5]) REPLAYED operation by operation, in the dtype the
// reference computes in: float64 once the Kalman state is (oracle/hlx_oracle.c observe(), `f64`), float32 before.  The fast
// float32 formulas of the observation section are good to ~1e-7 of their largest intermediate -- a closing speed of
// ~1000 m/s -- and the LOS rates divide that by the filtered range: under a few hundred metres (the end of every successful
// pursuit; under a metre when the filter was initialised from an empty delay-line sample) the quotient no longer holds
// 1e-5.  Used only there (out of line, los_frame configurations only).
struct LosExact { float rate_h, rate_v, lead_cos; };
DEV LosExact los_exact64(D3 kxp, D3 kxv, V3 ipos, V3 ivel) {
    const D3 frp = kxp - to_d3(ipos), frv = kxv - to_d3(ivel);                          // :758-759
    const double rng = sqrt(ddot(frp, frp));
    const double closing = -ddot(frp, frv) / (rng + 1e-6);                              // :786
    const D3 lu = (rng > 1e-6) ? D3{frp.x / rng, frp.y / rng, frp.z / rng} : D3{1.0, 0.0, 0.0};
    const double den = rng + 1e-6;
    const D3 rate = D3{(frv.x - closing * lu.x) / den, (frv.y - closing * lu.y) / den, (frv.z - closing * lu.z) / den};   // :810-811
    // np.cross(los_unit, [0, 0, 1]) and np.cross(los_unit, h): cp = a1*b2 - a2*b1, ... (:824-834)
    const D3 right = D3{lu.y * 1.0 - lu.z * 0.0, lu.z * 0.0 - lu.x * 1.0, lu.x * 0.0 - lu.y * 0.0};
    const double n = sqrt(ddot(right, right));
    const D3 h = (n > 1e-6) ? D3{right.x / n, right.y / n, right.z / n} : D3{1.0, 0.0, 0.0};
    const D3 v = D3{lu.y * h.z - lu.z * h.y, lu.z * h.x - lu.x * h.z, lu.x * h.y - lu.y * h.x};
    LosExact o;
    o.rate_h = (float)fmin(fmax(ddot(rate, h) / 0.5, -1.0), 1.0);                       // :844-845
    o.rate_v = (float)fmin(fmax(ddot(rate, v) / 0.5, -1.0), 1.0);                       // :848-849
    const D3 tv = frv + to_d3(ivel);                                                    // :861
    const double tvm = sqrt(ddot(tv, tv));
    o.lead_cos = (tvm > 1e-6) ? (float)ddot(D3{tv.x / tvm, tv.y / tvm, tv.z / tvm}, D3{-lu.x, -lu.y, -lu.z}) : 0.f;   // :862-868
    return o;
}
DEV LosExact los_exact32(V3 frp, V3 frv, V3 ivel) {      // IEEE float32 divisions (the compiler's), numpy's float32 dot / norm
    const float rng = snorm3(frp);
    const float den = rng + 1e-6f;
    const float closing = -(sdot3(frp, frv) / den);
    const V3 lu = (rng > 1e-6f) ? V3{frp.x / rng, frp.y / rng, frp.z / rng} : v3(1.f, 0.f, 0.f);
    const V3 rate = V3{(frv.x - closing * lu.x) / den, (frv.y - closing * lu.y) / den, (frv.z - closing * lu.z) / den};
    const V3 right = cross(lu, v3(0.f, 0.f, 1.f));
    const float n = snorm3(right);
    const V3 h = (n > 1e-6f) ? V3{right.x / n, right.y / n, right.z / n} : v3(1.f, 0.f, 0.f);
    const V3 v = cross(lu, h);
    LosExact o;
    o.rate_h = clampf(sdot3(rate, h) / 0.5f, -1.f, 1.f);
    o.rate_v = clampf(sdot3(rate, v) / 0.5f, -1.f, 1.f);
    const V3 tv = v3(frv.x + ivel.x, frv.y + ivel.y, frv.z + ivel.z);
    const float tvm = snorm3(tv);
    o.lead_cos = (tvm > 1e-6f) ? sdot3(V3{tv.x / tvm, tv.y / tvm, tv.z / tvm}, v3(-lu.x, -lu.y, -lu.z)) : 0.f;
    return o;
}

}  // namespace hlx
