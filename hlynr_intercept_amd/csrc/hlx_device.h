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
DEV double dnorm(D3 a) { return sqrt(ddot(a, a)); }

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
    return (x == 0.f || x == __builtin_inff()) ? x : r;
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
// physics_models.py:154-177  ISA atmosphere (float32 altitude) -> density, speed of sound
// ---------------------------------------------------------------------------------------------
DEV void atmosphere(float alt, float T0, float& rho, float& sos) {
    constexpr float R = 287.05f, G = 9.80665f, L = 0.0065f;
    constexpr float EXPO = (float)(9.80665 / (287.05 * 0.0065));
    constexpr float GAMMA_R = (float)(1.4 * 287.05);
    float T, P;
    if (alt <= 11000.0f) {
        T = T0 - L * alt;                         // :70-71
        P = 101325.0f * powf(T / T0, EXPO);       // :95-100
    } else if (alt <= 20000.0f) {                 // :72-74,102-108 (not reachable in shipped scenarios)
        T = 216.65f;
        P = 22632.0f * expf((-G * (alt - 11000.0f)) / (float)(287.05 * 216.65));
    } else {                                      // :76-78,110-113
        float ex = alt - 20000.0f;
        T = 216.65f * expf(-ex / 10000.0f);
        const float Pb = 5474.790039909648f;      // 22632 * exp(-g*9000/(R*216.65))  (:112)
        P = Pb * expf(-ex / 6000.0f);
    }
    rho = HLX_DIVF(P, R * T);                     // :166
    sos = HLX_SQRTF(GAMMA_R * T);                     // :167-169
}

struct DragParams {
    float subsonic, supersonic, mach_span, peak, base_cd, cd_super;   // cd_super = F(base_cd * supersonic_multiplier)
};
// physics_models.py:236-264  drag force vector (float32 velocity); `area` = reference_area
DEV V3 mach_drag_force(V3 v, float rho, float sos, float area, const DragParams& p) {
    float vm = snorm3(v);
    if (vm < 1e-6f) return v3(0.f, 0.f, 0.f);
    float mach = HLX_DIVF(vm, sos);                                          // :233-234
    float cd;
    if (mach < p.subsonic) cd = p.base_cd;                                  // :207-209
    else if (mach < p.supersonic) {
        float frac = HLX_DIVF(mach - p.subsonic, p.mach_span);              // :213-214
        cd = p.base_cd * (1.0f + (p.peak - 1.0f) * frac);                   // :215-216
    } else cd = p.cd_super;                                                 // :220 (python-float product)
    float a = (((0.5f * rho) * (vm * vm)) * cd) * area;                     // :258
    return V3{HLX_DIVF(-v.x, vm) * a, HLX_DIVF(-v.y, vm) * a, HLX_DIVF(-v.z, vm) * a};   // :262-264
}
// the same with a float64 air-relative velocity (Mach model on, simple float64 wind): generic kernel only
DEV D3 mach_drag_force64(D3 v, float rho, float sos, double area, const DragParams& p) {
    double vm = dnorm(v);
    if (vm < 1e-6) return d3(0., 0., 0.);
    double mach = vm / (double)sos;
    double cd;
    if (mach < (double)p.subsonic) cd = p.base_cd;
    else if (mach < (double)p.supersonic)
        cd = (double)p.base_cd * (1.0 + ((double)p.peak - 1.0) * ((mach - (double)p.subsonic) / ((double)p.supersonic - (double)p.subsonic)));
    else cd = (double)p.cd_super;
    double a = ((((double)(0.5f * rho)) * (vm * vm)) * cd) * area;
    return D3{(-v.x / vm) * a, (-v.y / vm) * a, (-v.z / vm) * a};
}
// environment.py:927-930 / 1111-1113 nan_to_num(nan=0, +-inf=+-lim), applied when any component is non-finite
DEV D3 nan_guard(D3 a, double lim) {
    bool ok = isfinite(a.x) && isfinite(a.y) && isfinite(a.z);
    if (ok) return a;
    auto fix = [lim](double x) { return isnan(x) ? 0.0 : (isinf(x) ? (x > 0. ? lim : -lim) : x); };
    return D3{fix(a.x), fix(a.y), fix(a.z)};
}

// ---------------------------------------------------------------------------------------------
// quaternion helpers  core.py:1103-1205 ([w, x, y, z])
// ---------------------------------------------------------------------------------------------
struct Quat {
    float w, x, y, z;
};
DEV V3 forward_vec(Quat q) {   // core.py:1143-1152 (feeds a threshold compare and pure outputs: fast float32)
    V3 f = v3(2.f * (q.x * q.z + q.w * q.y), 2.f * (q.y * q.z - q.w * q.x), 1.f - 2.f * (q.x * q.x + q.y * q.y));
    return f * __builtin_amdgcn_rcpf(fnorm_out(f) + 1e-6f);
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

}  // namespace hlx
