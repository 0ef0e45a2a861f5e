/*
 * hlx_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see hlx_oracle.h).
 *
 * CPU restatement of RomanSlack/Hlynr_Intercept's per-environment step, written to follow the
 * reference's arithmetic *including its mixed precision* under numpy >= 2 (NEP 50): values the
 * reference holds as float32 are rounded to float32 after every operation, values it holds as
 * float64 (simple-wind state, ground-radar measurements, the Kalman state once a float64
 * measurement has been absorbed, missile acceleration) stay double.  Python-float constants that
 * meet a float32 operand are "weak": they are rounded to float32 first.
 *
 * Conventions: every quantity lives in a `double` variable; F(x) rounds to float32; rr(x, is64)
 * rounds unless the reference value is float64 at that point.  float32 add/sub/mul/div/sqrt done
 * in double and rounded once are correctly rounded (53 >= 2*24+2), so this reproduces float32
 * arithmetic exactly.  np.dot/np.linalg.norm on float32 3-vectors = float32 products accumulated
 * in double, rounded once (OpenBLAS sdot as shipped with numpy 2.2.6; measured).
 * float32 matrix products (the Kalman predict's F @ P @ F.T) = OpenBLAS sgemm: one fused multiply-add
 * chain per element, k ascending (measured here, kf_predict below); matrix-vector products do not fuse.
 *
 * All file:line citations are to /root/reference/rl_system/.
 */
#define _GNU_SOURCE
#include "hlx_oracle.h"
#include "ref_math.h"

#include <math.h>
#include <string.h>

static inline double F(double x) { return (double)(float)x; }
static inline double rr(double x, int is64) { return is64 ? x : F(x); }
static inline double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

static double dot3(const double *a, const double *b, int is64) {
    if (is64) return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    return F(F(a[0] * b[0]) + F(a[1] * b[1]) + F(a[2] * b[2]));
}
static double norm3(const double *a, int is64) { return rr(sqrt(dot3(a, a, is64)), is64); }
static void cross3(const double *a, const double *b, double *o, int is64) {
    /* numpy.cross: cp0 = a1*b2; cp0 -= a2*b1 ... */
    o[0] = rr(rr(a[1] * b[2], is64) - rr(a[2] * b[1], is64), is64);
    o[1] = rr(rr(a[2] * b[0], is64) - rr(a[0] * b[2], is64), is64);
    o[2] = rr(rr(a[0] * b[1], is64) - rr(a[1] * b[0], is64), is64);
}

/* ------------------------------------------------------------------------------------------
 * physics_models.py:56-177  AtmosphericModel.get_atmospheric_properties (float32 altitude)
 * ---------------------------------------------------------------------------------------- */
static void atmosphere(double alt /*f32, >=0*/, double T0, double *rho, double *sos) {
    const double R = 287.05, G = 9.80665, L = 0.0065;
    double T, P;
    if (alt <= F(11000.0)) {
        T = F(F(T0) - F(F(L) * alt));                         /* :70-71 */
        double ratio = F(T / F(T0));                          /* :95 */
        double expo = G / (R * L);                            /* :99 (python float) */
        P = F(F(101325.0) * (double)powf((float)ratio, (float)expo)); /* :100 */
        *rho = F(P / F(F(R) * T));                            /* :166 */
        *sos = F(sqrt(F(F(1.4 * R) * T)));                    /* :167-169 */
    } else if (alt <= F(20000.0)) {
        /* :72-74,102-108 -- unreachable in the shipped scenarios (spawn altitude <= 4 km) */
        T = 216.65;
        double ex = F(alt - F(11000.0));
        double a = F(F(F(-G) * ex) / F(R * 216.65));
        P = F(F(22632.0) * (double)ref_np_expf((float)a));
        *rho = F(P / F(R * T));
        *sos = F(sqrt(1.4 * R * T));
    } else {
        double ex = F(alt - F(20000.0));
        T = F(F(216.65) * (double)ref_np_expf((float)F(F(-ex) / F(10000.0))));       /* :76-78 */
        double Pb = 22632.0 * exp(-G * (20000.0 - 11000.0) / (R * 216.65));    /* :112 (python floats) */
        P = F(F(Pb) * (double)ref_np_expf((float)F(F(-ex) / F(6000.0))));             /* :113 */
        *rho = F(P / F(F(R) * T));
        *sos = F(sqrt(F(F(1.4 * R) * T)));
    }
}

/* physics_models.py:197-264  MachDragModel.get_drag_force -> force vector */
static void mach_drag_force(const orc_config *c, const orc_state *s, const double *v, int is64, double rho,
                            double sos, double area, double *force) {
    double vm = norm3(v, is64);
    if (vm < rr(1e-6, is64)) { force[0] = force[1] = force[2] = 0.0; return; }
    double mach = rr(vm / sos, is64);                                           /* :233-234 */
    double cd;
    if (mach < rr(c->subsonic_mach, is64)) {
        cd = s->base_cd;                                                        /* :207-209 */
    } else if (mach < rr(c->supersonic_mach, is64)) {
        double frac = rr(rr(mach - rr(c->subsonic_mach, is64), is64) /
                         rr(c->supersonic_mach - c->subsonic_mach, is64), is64); /* :213-214 */
        double mult = rr(rr(1.0, is64) + rr(rr(s->transonic_peak - 1.0, is64) * frac, is64), is64);
        cd = rr(rr(s->base_cd, is64) * mult, is64);                             /* :215-216 */
    } else {
        cd = s->base_cd * c->supersonic_multiplier;                             /* :220 */
    }
    double a = F(F(0.5) * rho);                                                 /* :258 */
    a = rr(a * rr(vm * vm, is64), is64);
    a = rr(a * rr(cd, is64), is64);
    a = rr(a * rr(area, is64), is64);
    for (int i = 0; i < 3; ++i) {
        double d = rr(-v[i] / vm, is64);                                        /* :262 */
        force[i] = rr(d * a, is64);                                             /* :264 */
    }
}

/* environment.py:920-921 / 1099-1100 simple drag acceleration */
static void simple_drag_accel(const orc_config *c, const double *v, int is64, double rho, double mass,
                              double *acc) {
    double vm = norm3(v, is64);
    double c1;
    if (c->atmosphere) c1 = F(F(-0.5 * 0.3) * rho);   /* python * np.float32 -> float32 */
    else c1 = -0.5 * 0.3 * 1.225;                     /* python floats (environment.py:48-49,905) */
    double c2 = is64 ? c1 * vm : F(F(c1) * vm);
    for (int i = 0; i < 3; ++i) acc[i] = is64 ? (c2 * v[i]) / mass : F(F(c2 * v[i]) / F(mass));
}

static void nan_guard(double *a, double lim) { /* environment.py:927-930, 1111-1113 */
    if (isfinite(a[0]) && isfinite(a[1]) && isfinite(a[2])) return;
    for (int i = 0; i < 3; ++i) {
        if (isnan(a[i])) a[i] = 0.0;
        else if (isinf(a[i])) a[i] = a[i] > 0 ? lim : -lim;
    }
}

/* ------------------------------------------------------------------------------------------
 * quaternion helpers  core.py:1103-1205
 * ---------------------------------------------------------------------------------------- */
static void forward_vec(const float *q, double *o) { /* core.py:1143-1152 */
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double f[3] = {F(2.0 * F(F(x * z) + F(w * y))), F(2.0 * F(F(y * z) - F(w * x))),
                   F(1.0 - F(2.0 * F(F(x * x) + F(y * y))))};
    double n = norm3(f, 0);
    double d = F(n + F(1e-6));
    for (int i = 0; i < 3; ++i) o[i] = F(f[i] / d);
}
static void right_vec(const float *q, double *o) { /* core.py:1155-1164 */
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double f[3] = {F(1.0 - F(2.0 * F(F(y * y) + F(z * z)))), F(2.0 * F(F(x * y) + F(w * z))),
                   F(2.0 * F(F(x * z) - F(w * y)))};
    double n = norm3(f, 0);
    double d = F(n + F(1e-6));
    for (int i = 0; i < 3; ++i) o[i] = F(f[i] / d);
}
static void up_vec(const float *q, double *o) { /* core.py:1167-1176 */
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double f[3] = {F(2.0 * F(F(x * y) - F(w * z))), F(1.0 - F(2.0 * F(F(x * x) + F(z * z)))),
                   F(2.0 * F(F(y * z) + F(w * x)))};
    double n = norm3(f, 0);
    double d = F(n + F(1e-6));
    for (int i = 0; i < 3; ++i) o[i] = F(f[i] / d);
}
static void world_to_body(const double *v, int is64, const float *q, double *o) { /* core.py:1179-1205 */
    double f[3], r[3], u[3];
    forward_vec(q, f); right_vec(q, r); up_vec(q, u);
    o[0] = F(dot3(v, f, is64)); o[1] = F(dot3(v, r, is64)); o[2] = F(dot3(v, u, is64));
}
static void quat_to_euler(const float *q, double *e) { /* core.py:1103-1121 */
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double sinr = F(2.0 * F(F(w * x) + F(y * z)));
    double cosr = F(1.0 - F(2.0 * F(F(x * x) + F(y * y))));
    e[0] = (double)atan2f((float)sinr, (float)cosr);
    double sinp = F(2.0 * F(F(w * y) - F(z * x)));
    e[1] = (double)asinf((float)clipd(sinp, -1.0, 1.0));
    double siny = F(2.0 * F(F(w * z) + F(x * y)));
    double cosy = F(1.0 - F(2.0 * F(F(y * y) + F(z * z))));
    e[2] = (double)atan2f((float)siny, (float)cosy);
}

/* LOS orthonormal basis shared by action transform and observation
 * (environment.py:965-1021, core.py:802-834,932-945) */
static void los_basis(const double *los_unit, int is64, double *h, double *v) {
    const double up[3] = {0.0, 0.0, 1.0};
    double right[3];
    cross3(los_unit, up, right, is64);
    double n = norm3(right, is64);
    if (n > rr(1e-6, is64)) { for (int i = 0; i < 3; ++i) h[i] = rr(right[i] / n, is64); }
    else { h[0] = 1.0; h[1] = 0.0; h[2] = 0.0; }
    cross3(los_unit, h, v, is64);
}

/* ------------------------------------------------------------------------------------------
 * Kalman filter  core.py:12-133 (full 6x6 float32 covariance, as the reference holds it)
 * ---------------------------------------------------------------------------------------- */
static void kf_reset(orc_state *s) { /* core.py:65-69 */
    s->kf_init = 0; s->kf_x_is64 = 0;
    memset(s->kf_x, 0, sizeof s->kf_x);
    for (int i = 0; i < 36; ++i) s->kf_P[i] = 0.0f;
    for (int i = 0; i < 6; ++i) s->kf_P[i * 6 + i] = 1000.0f;
}
static void kf_update(orc_state *s, const double *z, int z64) { /* core.py:91-116 */
    if (!s->kf_init) { /* :93-96, initialize :71-78 -- assignment into the float32 state array */
        for (int i = 0; i < 3; ++i) { s->kf_x[i] = F(z[i]); s->kf_x[i + 3] = 0.0; }
        s->kf_init = 1;
        return;
    }
    float *P = s->kf_P;
    int y64 = z64 || s->kf_x_is64;
    double y[3], Sinv[3], K[6][3];
    for (int i = 0; i < 3; ++i) y[i] = rr(z[i] - s->kf_x[i], y64);             /* :99 */
    for (int i = 0; i < 3; ++i) {
        double S = F((double)P[i * 6 + i] + 400.0);                               /* :102, R = 20^2 I */
        Sinv[i] = F(1.0 / S);                                                     /* :106 np.linalg.inv */
        for (int j = 0; j < 3; ++j) if (j != i && P[i * 6 + j] != 0.0f) s->structure_violations++;
    }
    for (int r = 0; r < 6; ++r) for (int j = 0; j < 3; ++j) K[r][j] = F((double)P[r * 6 + j] * Sinv[j]);
    int x64 = s->kf_x_is64 || y64;
    for (int r = 0; r < 6; ++r) {                                                 /* :112 */
        double acc = 0.0;
        for (int j = 0; j < 3; ++j) acc = rr(acc + rr(K[r][j] * y[j], y64), y64);
        s->kf_x[r] = rr(s->kf_x[r] + acc, x64);
    }
    s->kf_x_is64 = x64;
    float Pn[36];                                                                 /* :115-116 */
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
        double acc = 0.0;
        for (int k = 0; k < 6; ++k) {
            double ikh = (k < 3) ? F((r == k ? 1.0 : 0.0) - K[r][k]) : (r == k ? 1.0 : 0.0);
            acc = F(acc + F(ikh * (double)P[k * 6 + c]));
        }
        Pn[r * 6 + c] = (float)acc;
    }
    memcpy(P, Pn, sizeof Pn);
}
static void kf_predict(orc_state *s, double dt) { /* core.py:80-89 */
    if (!s->kf_init) return;
    int x64 = s->kf_x_is64;
    double dtf = F(dt);
    for (int i = 0; i < 3; ++i) s->kf_x[i] = rr(s->kf_x[i] + rr(dtf * s->kf_x[i + 3], x64), x64);
    float *P = s->kf_P;
    /* `self.F @ self.P @ self.F.T` are two float32 6x6 matrix products = OpenBLAS sgemm, whose micro-kernel accumulates each
     * element as ONE fused multiply-add chain over k in ascending order from 0 (measured against numpy 2.2.6 / OpenBLAS 0.3.29
     * here: 100 % bit match over 3000 random products of every shape the filter forms, dense and block-structured; an unfused
     * order matches 0-14 %).  With F = [I dt*I; 0 I] a chain has at most two non-zero terms, 1 * a (exact) and dt * b:
     * element = fmaf(dt, b, a), one rounding.  (`self.F @ self.state` is a matrix-VECTOR product = sgemv / dgemv, which does
     * not fuse: x + round(dt * v), 100 % over 20 000 draws; the update's products have one non-zero term per element.) */
    double FP[36], FPF[36];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c)
        FP[r * 6 + c] = (r < 3) ? (double)fmaf((float)dtf, P[(r + 3) * 6 + c], P[r * 6 + c]) : (double)P[r * 6 + c];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c)
        FPF[r * 6 + c] = (c < 3) ? (double)fmaf((float)FP[r * 6 + c + 3], (float)dtf, (float)FP[r * 6 + c]) : FP[r * 6 + c];
    double q = 5.0 * 5.0;                                                         /* core.py:333, :34 */
    double q11 = F(q * pow(dt, 4) / 4.0), q12 = F(q * pow(dt, 3) / 2.0), q22 = F(q * dt * dt); /* :35-42 */
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
        double Q = 0.0;
        if (r == c) Q = (r < 3) ? q11 : q22;
        else if (r % 3 == c % 3) Q = q12;
        P[r * 6 + c] = (float)F(FPF[r * 6 + c] + Q);
    }
}

/* ------------------------------------------------------------------------------------------
 * observation: core.py:511-691 compute_radar_detection + :693-1032 compute
 * noise: [0] onboard U, [1] ground U, [2..4] ground pos N, [5..7] ground vel N, [8] datalink U
 * ---------------------------------------------------------------------------------------- */
static void observe(const orc_config *c, orc_state *s, const double *nz, float *obs) {
    double ip[3], iv[3], mp[3], mv[3];
    for (int i = 0; i < 3; ++i) { ip[i] = s->int_pos[i]; iv[i] = s->int_vel[i]; mp[i] = s->mis_pos[i]; mv[i] = s->mis_vel[i]; }
    const float *q = s->int_quat;

    /* ---- onboard radar (core.py:531-593) ---- */
    double rel[3];
    for (int i = 0; i < 3; ++i) rel[i] = F(mp[i] - ip[i]);
    double range = norm3(rel, 0);
    int on_det = !(range > F(c->radar_range));                                   /* :539 */
    double fwd[3], tom[3];
    forward_vec(q, fwd);
    for (int i = 0; i < 3; ++i) tom[i] = F(rel[i] / F(range + F(1e-6)));       /* :546 */
    double beam_angle = (double)acosf((float)clipd(dot3(fwd, tom, 0), -1.0, 1.0)); /* :547 */
    double half_beam = (c->beam_width_deg / 2.0) * (M_PI / 180.0);              /* :548 np.radians -> float64 */
    if (on_det && beam_angle > half_beam) on_det = 0;                            /* :553 */
    if (on_det) {                                                                /* :559-566 */
        double rf = F(1.0 - F(F(range / F(c->radar_range)) * F(0.5)));
        double aq = F(F(F(c->radar_quality) * rf) * F(c->onboard_reliability));
        if (F(nz[0]) > aq) on_det = 0;
    }
    double d_on[3] = {0, 0, 0};
    int d_on_det = 0;
    if (s->on_delay > 0) {                                                       /* :576-588, core.py:175-208 */
        int cap = s->on_delay + 1;
        if (s->on_len == cap) {
            for (int k = 1; k < cap; ++k) { memcpy(s->on_ring[k - 1], s->on_ring[k], sizeof s->on_ring[0]); s->on_det[k - 1] = s->on_det[k]; }
            s->on_len--;
        }
        memcpy(s->on_ring[s->on_len], rel, sizeof rel);
        s->on_det[s->on_len] = on_det;
        s->on_len++;
        s->on_count++;
        if (s->on_count >= s->on_delay && s->on_len > s->on_delay) {
            memcpy(d_on, s->on_ring[0], sizeof d_on);
            d_on_det = s->on_det[0];
        }
    } else {
        memcpy(d_on, rel, sizeof rel);
        d_on_det = on_det;
    }

    /* ---- ground radar (core.py:368-438) ---- */
    int g_det = 0;
    double g_meas[7] = {0, 0, 0, 0, 0, 0, 0};
    double gp[3] = {F(c->ground_pos[0]), F(c->ground_pos[1]), F(c->ground_pos[2])};
    if (c->ground_enabled) {
        double g2m[3];
        for (int i = 0; i < 3; ++i) g2m[i] = F(mp[i] - gp[i]);
        double grange = norm3(g2m, 0);
        g_det = 1;
        if (grange > F(c->ground_max_range)) g_det = 0;                          /* :396 */
        if (g_det && grange > F(1e-6)) {                                         /* :401-406 */
            double el = (double)asinf((float)clipd(F(g2m[2] / grange), -1.0, 1.0));
            if (el < c->ground_min_elev || el > c->ground_max_elev) g_det = 0;
        }
        if (g_det && mp[2] < F(50.0)) g_det = 0;                                 /* :409 */
        if (g_det) {                                                             /* :413-418 */
            double dp = F(F(c->ground_base_quality) * F(1.0 - F(F(grange / F(c->ground_max_range)) * F(0.4))));
            dp = F(dp * F(c->weather_factor));
            dp = F(dp * F(c->ground_reliability));
            if (F(nz[1]) > dp) g_det = 0;
            else {
                for (int i = 0; i < 3; ++i) {                                    /* :422-429 */
                    g_meas[i] = rel[i] + c->ground_range_accuracy * nz[2 + i];
                    g_meas[3 + i] = F(mv[i] - iv[i]) + c->ground_velocity_accuracy * nz[5 + i];
                }
                g_meas[6] = dp;
            }
        }
    }
    double d_g[7] = {0, 0, 0, 0, 0, 0, 0};
    int d_g_det = 0, d_g64 = 0;
    if (c->ground_enabled && c->ground_delay > 0) {                              /* core.py:609-627 */
        int cap = c->ground_delay + 1;
        if (s->g_len == cap) {
            for (int k = 1; k < cap; ++k) { memcpy(s->g_ring[k - 1], s->g_ring[k], sizeof s->g_ring[0]); s->g_pos_is64[k - 1] = s->g_pos_is64[k]; }
            s->g_len--;
        }
        memcpy(s->g_ring[s->g_len], g_meas, sizeof g_meas);
        s->g_pos_is64[s->g_len] = g_det;
        s->g_len++;
        s->g_count++;
        if (s->g_count >= c->ground_delay && s->g_len > c->ground_delay) {
            memcpy(d_g, s->g_ring[0], sizeof d_g);
            d_g64 = s->g_pos_is64[0];
            d_g_det = g_det;                                                     /* :626 -- CURRENT flag (reference quirk) */
        }
    } else {
        memcpy(d_g, g_meas, sizeof d_g);
        d_g_det = g_det; d_g64 = g_det;
    }
    double g_quality = d_g[6];

    /* ---- datalink (core.py:440-474) ---- */
    double datalink = 0.0;
    int datalink_is_np = 0;
    if (c->ground_enabled) {
        double d[3];
        for (int i = 0; i < 3; ++i) d[i] = F(ip[i] - gp[i]);
        double lr = norm3(d, 0);
        if (!(lr > F(c->max_datalink_range))) {
            double x = F(lr / F(c->max_datalink_range));
            double rf = F(1.0 - F(x * x));
            double vm = norm3(iv, 0);
            double vr = F(vm / F(1000.0));
            double dop = (F(0.3) < vr) ? 1.0 - 0.3 : F(1.0 - vr);                   /* python min(a, 0.3) */
            if (!(nz[8] < c->datalink_packet_loss)) {
                datalink = clipd(F(F(rf * F(dop)) * F(0.95)), 0.0, 1.0);
                datalink_is_np = 1;
            }
        }
    }

    /* ---- fusion confidence (core.py:476-509, :638) ---- */
    double on_q = d_on_det ? c->radar_quality : 0.0;
    double fusion;
    if (!d_on_det && !d_g_det) fusion = 0.0;
    else if (d_on_det && !d_g_det) fusion = on_q * 0.5;
    else if (!d_on_det) fusion = d_g64 ? F(g_quality * F(0.6)) : g_quality * 0.6;
    else {
        double dd[3];
        for (int i = 0; i < 3; ++i) dd[i] = rr(d_on[i] - d_g[i], d_g64);
        double pe = norm3(dd, d_g64);
        double t = rr(pe / rr(200.0, d_g64), d_g64);
        double agree = (1.0 < t) ? 0.0 : rr(rr(1.0, d_g64) - t, d_g64);
        double A = 0.35 * on_q;
        if (d_g64) {  /* np.float32 quality, np.float64 agreement */
            double AB = F(F(A) + F(F(0.50) * g_quality));
            fusion = clipd(AB + 0.15 * agree, 0.0, 1.0);
        } else {      /* python-float quality (0.0), np.float32 agreement */
            double AB = A + 0.50 * g_quality;
            fusion = clipd(F(F(AB) + F(F(0.15) * agree)), 0.0, 1.0);
        }
    }

    /* ================= compute()  core.py:693-1032 ================= */
    for (int i = 0; i < 26; ++i) obs[i] = 0.0f;
    int meas_avail = 0, f64 = 0;
    double frp[3] = {0, 0, 0}, frv[3] = {0, 0, 0};
    if (d_on_det || d_g_det) {                                                   /* :732-759 */
        double fused[3];
        int m64;
        if (d_on_det && d_g_det) {
            double w_on = c->radar_quality, w_g = g_quality;
            double total = d_g64 ? F(F(w_on) + w_g) : w_on + w_g;
            m64 = d_g64;
            for (int i = 0; i < 3; ++i) {
                double a = F(d_on[i] * F(w_on));
                double b = d_g64 ? d_g[i] * w_g : 0.0;
                double sum = rr(a + b, m64);
                fused[i] = m64 ? sum / total : F(sum / F(total));
            }
        } else if (d_on_det) { m64 = 0; memcpy(fused, d_on, sizeof fused); }
        else { m64 = d_g64; memcpy(fused, d_g, sizeof fused); }
        double z[3];
        for (int i = 0; i < 3; ++i) z[i] = rr(ip[i] + fused[i], m64);            /* :749 */
        kf_update(s, z, m64);                                                    /* :752 */
        f64 = s->kf_x_is64;
        for (int i = 0; i < 3; ++i) { frp[i] = rr(s->kf_x[i] - ip[i], f64); frv[i] = rr(s->kf_x[3 + i] - iv[i], f64); }
        meas_avail = 1;
    } else {                                                                     /* :760-774 */
        kf_predict(s, c->dt);
        if (s->kf_init) {
            f64 = s->kf_x_is64;
            for (int i = 0; i < 3; ++i) { frp[i] = rr(s->kf_x[i] - ip[i], f64); frv[i] = rr(s->kf_x[3 + i] - iv[i], f64); }
        }
    }
    const double mr = c->max_range, mvel = c->max_velocity;
    if (meas_avail || s->kf_init) {                                              /* :778-906 */
        double rrange = norm3(frp, f64);
        double closing = rr(-dot3(frp, frv, f64) / rr(rrange + rr(1e-6, f64), f64), f64); /* :786 */
        double ifw[3];
        forward_vec(q, ifw);
        if (c->obs_mode == 2) {                                                  /* :791-868 los_frame */
            obs[0] = (float)clipd(rr(rrange / rr(mr, f64), f64), 0.0, 1.0);
            obs[1] = (float)clipd(rr(closing / rr(mvel, f64), f64), -1.0, 1.0);
            double lu[3] = {1.0, 0.0, 0.0};
            int lu64 = f64;
            if (rrange > rr(1e-6, f64)) { for (int i = 0; i < 3; ++i) lu[i] = rr(frp[i] / rrange, f64); }
            double tang[3], rate[3];
            for (int i = 0; i < 3; ++i) {
                tang[i] = rr(frv[i] - rr(closing * lu[i], f64), f64);            /* :810 */
                rate[i] = rr(tang[i] / rr(rrange + rr(1e-6, f64), f64), f64);    /* :811 */
            }
            double h[3], v[3];
            los_basis(lu, lu64, h, v);
            obs[2] = (float)clipd(rr(dot3(rate, h, f64) / rr(0.5, f64), f64), -1.0, 1.0);  /* :844-845 */
            obs[3] = (float)clipd(rr(dot3(rate, v, f64) / rr(0.5, f64), f64), -1.0, 1.0);  /* :848-849 */
            double ivm = norm3(iv, 0);
            if (ivm > F(1e-6)) {                                                 /* :852-858 */
                double u[3];
                for (int i = 0; i < 3; ++i) u[i] = F(iv[i] / ivm);
                obs[4] = (float)dot3(u, lu, f64);
            } else obs[4] = 0.0f;
            double tv[3];
            for (int i = 0; i < 3; ++i) tv[i] = rr(frv[i] + iv[i], f64);         /* :861 */
            double tvm = norm3(tv, f64);
            if (tvm > rr(1e-6, f64)) {
                double u[3], nl[3];
                for (int i = 0; i < 3; ++i) { u[i] = rr(tv[i] / tvm, f64); nl[i] = -lu[i]; }
                obs[5] = (float)dot3(u, nl, f64);
            } else obs[5] = 0.0f;
        } else if (c->obs_mode == 1) {                                           /* :870-876 body_frame */
            double bp[3], bv[3];
            world_to_body(frp, f64, q, bp);
            world_to_body(frv, f64, q, bv);
            for (int i = 0; i < 3; ++i) {
                obs[i] = (float)clipd(F(bp[i] / F(mr)), -1.0, 1.0);
                obs[3 + i] = (float)clipd(F(bv[i] / F(mvel)), -1.0, 1.0);
            }
        } else {                                                                 /* :878-882 world_frame */
            for (int i = 0; i < 3; ++i) {
                obs[i] = (float)clipd(rr(frp[i] / rr(mr, f64), f64), -1.0, 1.0);
                obs[3 + i] = (float)clipd(rr(frv[i] / rr(mvel, f64), f64), -1.0, 1.0);
            }
        }
        if (closing > 0) {                                                       /* :885-889 */
            double tti = rr(rrange / closing, f64);
            obs[13] = (float)clipd(rr(rr(1.0, f64) - rr(tti / rr(100.0, f64), f64), f64), -1.0, 1.0);
        } else obs[13] = -1.0f;
        double pu = F(F((double)s->kf_P[0] + (double)s->kf_P[7]) + (double)s->kf_P[14]); /* :892 trace */
        double tq = clipd(1.0 - pu / 10000.0, 0.0, 1.0);                         /* :893 */
        if (d_on_det) tq *= c->radar_quality;                                    /* :894-895 */
        obs[14] = (float)tq;
        obs[15] = (float)clipd(rr(closing / rr(mvel, f64), f64), -1.0, 1.0);     /* :899 */
        if (rrange > rr(1e-6, f64)) {                                            /* :902-906 */
            double tt[3];
            for (int i = 0; i < 3; ++i) tt[i] = rr(frp[i] / rrange, f64);
            obs[16] = (float)dot3(ifw, tt, f64);
        } else obs[16] = 1.0f;
    } else {                                                                     /* :907-917 */
        for (int i = 0; i < 6; ++i) obs[i] = -2.0f;
        obs[13] = -1.0f; obs[14] = 0.0f; obs[15] = 0.0f; obs[16] = 0.0f;
    }
    /* [6-8] own velocity (:919-964) */
    if (c->obs_mode == 2) {
        double sp = norm3(iv, 0);
        obs[6] = (float)clipd(F(sp / F(mvel)), 0.0, 1.0);
        if (meas_avail || s->kf_init) {
            double r2 = norm3(frp, f64);
            double lu[3] = {1.0, 0.0, 0.0};
            if (r2 > rr(1e-6, f64)) { for (int i = 0; i < 3; ++i) lu[i] = rr(frp[i] / r2, f64); }
            double h[3], v[3];
            los_basis(lu, f64, h, v);
            obs[7] = (float)clipd(rr(dot3(iv, h, f64) / rr(mvel, f64), f64), -1.0, 1.0);
            obs[8] = (float)clipd(rr(dot3(iv, v, f64) / rr(mvel, f64), f64), -1.0, 1.0);
        } else { obs[7] = 0.0f; obs[8] = 0.0f; }
    } else if (c->obs_mode == 1) {
        double bv[3];
        world_to_body(iv, 0, q, bv);
        for (int i = 0; i < 3; ++i) obs[6 + i] = (float)clipd(F(bv[i] / F(mvel)), -1.0, 1.0);
    } else {
        for (int i = 0; i < 3; ++i) obs[6 + i] = (float)clipd(F(iv[i] / F(mvel)), -1.0, 1.0);
    }
    /* [9-11] orientation (:966-974) */
    if (c->obs_mode == 0) {
        double e[3];
        quat_to_euler(q, e);
        for (int i = 0; i < 3; ++i) obs[9 + i] = (float)F(e[i] / F(M_PI));
    }
    obs[12] = (float)clipd(F((double)s->fuel / F(100.0)), 0.0, 1.0);             /* :977 */
    /* ground block (:979-1024) */
    int dl_ok = datalink_is_np ? (datalink > F(0.1)) : (datalink > 0.1);
    if (d_g_det && dl_ok) {
        const double *gpv = d_g, *gvv = d_g + 3;
        if (c->obs_mode == 2) {
            double gr = norm3(gpv, d_g64);
            double gc = rr(-dot3(gpv, gvv, d_g64) / rr(gr + rr(1e-6, d_g64), d_g64), d_g64);
            obs[17] = (float)clipd(rr(gr / rr(mr, d_g64), d_g64), 0.0, 1.0);
            obs[18] = (float)clipd(rr(gc / rr(mvel, d_g64), d_g64), -1.0, 1.0);
            if (gr > rr(1e-6, d_g64)) {
                double tv[3];
                for (int i = 0; i < 3; ++i) tv[i] = rr(gvv[i] - rr(gc * rr(gpv[i] / gr, d_g64), d_g64), d_g64);
                double lrate = rr(norm3(tv, d_g64) / gr, d_g64);
                obs[19] = (float)clipd(rr(lrate / rr(0.5, d_g64), d_g64), 0.0, 1.0);
            } else obs[19] = 0.0f;
            obs[20] = obs[21] = obs[22] = 0.0f;
        } else if (c->obs_mode == 1) {
            double bp[3], bv[3];
            world_to_body(gpv, d_g64, q, bp);
            world_to_body(gvv, d_g64, q, bv);
            for (int i = 0; i < 3; ++i) {
                obs[17 + i] = (float)clipd(F(bp[i] / F(mr)), -1.0, 1.0);
                obs[20 + i] = (float)clipd(F(bv[i] / F(mvel)), -1.0, 1.0);
            }
        } else {
            for (int i = 0; i < 3; ++i) {
                obs[17 + i] = (float)clipd(rr(gpv[i] / rr(mr, d_g64), d_g64), -1.0, 1.0);
                obs[20 + i] = (float)clipd(rr(gvv[i] / rr(mvel, d_g64), d_g64), -1.0, 1.0);
            }
        }
        obs[23] = (float)g_quality;
    } else {
        for (int i = 17; i < 23; ++i) obs[i] = -2.0f;
        obs[23] = 0.0f;
    }
    obs[24] = (float)datalink;
    obs[25] = (float)fusion;
}

/* ------------------------------------------------------------------------------------------
 * reset()  environment.py:353-603
 * ---------------------------------------------------------------------------------------- */
void orc_reset(const orc_config *c, orc_state *s, const double *nz, float *obs) {
    /* rings + KF (:362, core.py:341-347) */
    s->on_count = s->on_len = 0;
    s->g_count = s->g_len = 0;
    memset(s->on_ring, 0, sizeof s->on_ring); memset(s->on_det, 0, sizeof s->on_det);
    memset(s->g_ring, 0, sizeof s->g_ring); memset(s->g_pos_is64, 0, sizeof s->g_pos_is64);
    kf_reset(s);
    double tp[3] = {F(c->target_pos[0]), F(c->target_pos[1]), F(c->target_pos[2])};
    /* missile(s) (:386-435): one, or volley_size of them, each with its own four draws in spawn order */
    const int K = c->volley_mode ? c->volley_size : 1;
    double mp[3];   /* missile 0 after the loop: `self.missile_state = self.missile_states[0]` (:439) */
    for (int k = K - 1; k >= 0; --k) {   /* (descending only so that mp ends as missile 0; the draws are per-slot) */
        const double *mz = k == 0 ? nz : nz + 32 + 4 * (k - 1);
        if (c->mis_spawn_spherical) {                                            /* :390-406 */
            double radius = c->mis_radius[0] + (c->mis_radius[1] - c->mis_radius[0]) * mz[0];
            double az = (c->mis_azimuth_deg[0] + (c->mis_azimuth_deg[1] - c->mis_azimuth_deg[0]) * mz[1]) * M_PI / 180.0;
            double el = (c->mis_elevation_deg[0] + (c->mis_elevation_deg[1] - c->mis_elevation_deg[0]) * mz[2]) * M_PI / 180.0;
            mp[0] = F(tp[0] + radius * cos(el) * cos(az));
            mp[1] = F(tp[1] + radius * cos(el) * sin(az));
            mp[2] = F(tp[2] + radius * sin(el));
        } else {
            for (int i = 0; i < 3; ++i) mp[i] = F(c->mis_pos_lo[i] + (c->mis_pos_hi[i] - c->mis_pos_lo[i]) * mz[i]); /* :409 */
        }
        double speed = c->mis_speed[0] + (c->mis_speed[1] - c->mis_speed[0]) * mz[3]; /* :415 */
        double tt[3];
        for (int i = 0; i < 3; ++i) tt[i] = F(tp[i] - mp[i]);
        double ttd = norm3(tt, 0);
        for (int i = 0; i < 3; ++i) {
            /* :421-423; the spawned-on-target fallback (:426) is unreachable for any box that excludes the target */
            double v = (ttd > F(1e-6)) ? F(F(tt[i] / ttd) * F(speed)) : 0.0;
            s->v_pos[k][i] = (float)mp[i]; s->v_vel[k][i] = (float)v;
            if (k == 0) { s->mis_pos[i] = (float)mp[i]; s->mis_vel[i] = (float)v; }
        }
        s->v_active[k] = 1;
    }
    s->prio = 0; s->n_intercepted = 0;
    /* interceptor (:442-467) */
    double ipos[3];
    for (int i = 0; i < 3; ++i) ipos[i] = F(c->int_pos_lo[i] + (c->int_pos_hi[i] - c->int_pos_lo[i]) * nz[4 + i]);
    double ivel[3];
    double rel[3];
    for (int i = 0; i < 3; ++i) rel[i] = F(mp[i] - ipos[i]);
    double reld = norm3(rel, 0);
    if (c->int_vel_toward_missile && reld > F(1e-6)) {                           /* :452-462 */
        double sp = c->int_speed[0] + (c->int_speed[1] - c->int_speed[0]) * nz[7];
        for (int i = 0; i < 3; ++i) ivel[i] = F(F(rel[i] / reld) * F(sp));
    } else {
        for (int i = 0; i < 3; ++i) ivel[i] = F(c->int_vel_lo[i] + (c->int_vel_hi[i] - c->int_vel_lo[i]) * nz[7 + i]); /* :467 */
    }
    /* volley: per-missile minimum distances (:469-473); the interceptor points at the CLOSEST missile (:476-487),
     * while the first observation and _prev_distance use missile 0 (:439, :579) */
    double orel[3] = {rel[0], rel[1], rel[2]}, oreld = reld;
    if (c->volley_mode) {
        double best = INFINITY;
        for (int k = 0; k < K; ++k) {
            double r[3];
            for (int i = 0; i < 3; ++i) r[i] = F((double)s->v_pos[k][i] - ipos[i]);
            double dk = norm3(r, 0);
            s->v_min[k] = (float)dk;
            if (dk < best) { best = dk; oreld = dk; for (int i = 0; i < 3; ++i) orel[i] = r[i]; }
        }
    }
    /* initial orientation: rotate +Z onto the LOS (:489-530), float64 until the final cast */
    double quat[4] = {1.0, 0.0, 0.0, 0.0};
    if (oreld > F(1e-6)) {
        double fd[3];
        for (int i = 0; i < 3; ++i) fd[i] = F(orel[i] / oreld);
        double ax[3] = {-fd[1], fd[0], 0.0};
        double axl = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        double cosang = fd[2];
        if (axl > 1e-6) {
            double ang = acos(clipd(cosang, -1.0, 1.0));
            double half = ang / 2.0, sh = sin(half);
            quat[0] = cos(half); quat[1] = ax[0] / axl * sh; quat[2] = ax[1] / axl * sh; quat[3] = ax[2] / axl * sh;
        } else if (!(cosang > 0)) { quat[0] = 0.0; quat[1] = 1.0; }               /* :523-527 */
    }
    for (int i = 0; i < 3; ++i) { s->int_pos[i] = (float)ipos[i]; s->int_vel[i] = (float)ivel[i]; }
    for (int i = 0; i < 4; ++i) s->int_quat[i] = (float)quat[i];
    s->fuel = 100.0f;                                                            /* :537 */
    for (int i = 0; i < 3; ++i) { s->wind[i] = F(c->base_wind[i]); s->thrust_actual[i] = 0.0f; } /* :542,:549 */
    s->wind_is64 = 0;
    /* domain randomisation (:552-562, physics_randomizer.py:137-297) */
    if (c->domain_randomization) {
        const double *z = nz + 19;
        double mult[13];
        for (int k = 0; k < 13; ++k) {
            if (k == 1) { mult[k] = 0.0 + c->dr_variations[k] * z[k]; continue; } /* temperature offset :169-171 */
            double lo = 0.1, hi = 3.0;
            if (k == 6) { lo = 0.5; hi = 1.0; }
            if (k == 7) { lo = 0.5; hi = 2.0; }
            mult[k] = clipd(1.0 + c->dr_variations[k] * z[k], lo, hi);           /* :223-241 */
        }
        if (c->atmosphere) s->T0 = s->T0 + mult[1];                              /* :258-261 (accumulates) */
        if (c->mach_drag) { s->base_cd = 0.3 * mult[2]; s->transonic_peak = 3.0 * mult[3]; } /* :270-280 */
        if (c->onboard_delay > 0) {                                              /* :289-297 */
            int nd = (int)(3 * mult[4]);
            nd = nd > 10 ? 10 : nd; nd = nd < 1 ? 1 : nd;
            s->on_delay = nd;
        }
    }
    s->steps = 0; s->total_fuel_used = 0.0;                                      /* :565-566 */
    observe(c, s, nz + 10, obs);                                                 /* :570 */
    s->prev_distance = (float)reld;                                              /* :579-589 */
    s->last_distance = s->prev_distance;
    s->min_distance = s->prev_distance;
    s->worsening = 0; s->crossed = 0;
}

/* one-time per-env initialisation of what reset() does not touch (constructor state) */
static void init_ctor_state(const orc_config *c, orc_state *s) {
    memset(s, 0, sizeof *s);
    s->T0 = 288.15; s->base_cd = 0.3; s->transonic_peak = c->transonic_peak_multiplier;
    s->on_delay = c->onboard_delay;
}
void orc_init(const orc_config *c, orc_state *s, int32_t n) { for (int32_t i = 0; i < n; ++i) init_ctor_state(c, s + i); }

/* ------------------------------------------------------------------------------------------
 * step()  environment.py:605-859
 * ---------------------------------------------------------------------------------------- */
void orc_step(const orc_config *c, orc_state *s, const float *action, const double *nz, orc_out *out) {
    const double dt = c->dt, dtf = F(c->dt);
    s->steps += 1;                                                               /* :607 */
    double a[6];
    for (int i = 0; i < 6; ++i) a[i] = action[i];
    double los_a0 = a[0];
    if (c->obs_mode == 2) {                                                      /* :618-620, :965-1063 */
        double rel[3], lu[3] = {1.0, 0.0, 0.0}, h[3], v[3];
        for (int i = 0; i < 3; ++i) rel[i] = F((double)s->mis_pos[i] - (double)s->int_pos[i]);
        double rg = norm3(rel, 0);
        if (rg > F(1e-6)) for (int i = 0; i < 3; ++i) lu[i] = F(rel[i] / rg);
        los_basis(lu, 0, h, v);
        double w[3];
        for (int i = 0; i < 3; ++i) w[i] = F(F(F(a[0] * lu[i]) + F(a[1] * h[i])) + F(a[2] * v[i]));
        a[0] = w[0]; a[1] = w[1]; a[2] = w[2];
    }
    /* SafetyClamp.apply  core.py:1069-1100 */
    int clamped = 0;
    if (s->fuel <= 0.0f) { a[0] *= 0; a[1] *= 0; a[2] *= 0; clamped = 1; }
    double am = norm3(a, 0);
    if (am > F(50.0)) { double k = F(F(50.0) / am); for (int i = 0; i < 3; ++i) a[i] = F(a[i] * k); clamped = 1; }
    double gm = norm3(a + 3, 0);
    if (gm > F(5.0)) { double k = F(F(5.0) / gm); for (int i = 3; i < 6; ++i) a[i] = F(a[i] * k); clamped = 1; }

    /* ---- _update_interceptor  environment.py:861-956 ---- */
    double thr[3], ang[3];
    for (int i = 0; i < 3; ++i) { thr[i] = F(a[i] * F(10000.0)); ang[i] = F(a[3 + i] * F(20.0)); }
    if (c->thrust_lag) {                                                         /* :874-878 */
        for (int i = 0; i < 3; ++i) {
            double err = F(thr[i] - (double)s->thrust_actual[i]);
            double act = F((double)s->thrust_actual[i] + F(F(err * dtf) / F(c->thrust_tau)));
            s->thrust_actual[i] = (float)act;
            thr[i] = act;
        }
    }
    double tmag = norm3(thr, 0);
    double fc = F(F(F(tmag / F(500.0)) * F(0.1)) * dtf);                         /* :884 */
    s->fuel = (float)F((double)s->fuel - fc);
    s->total_fuel_used = F(s->total_fuel_used + fc);
    if (s->fuel <= 0.0f) {                                                       /* :888-892 */
        s->fuel = 0.0f;
        for (int i = 0; i < 3; ++i) { thr[i] = 0.0; if (c->thrust_lag) s->thrust_actual[i] = 0.0f; }
    }
    double tacc[3];
    for (int i = 0; i < 3; ++i) tacc[i] = F(thr[i] / F(500.0));                  /* :896 */
    double alt = s->int_pos[2] > 0.0f ? (double)s->int_pos[2] : 0.0;             /* :899 */
    double rho = 1.225, sos = 343.0;
    if (c->atmosphere) atmosphere(alt, s->T0, &rho, &sos);                       /* :900-906 */
    int w64 = s->wind_is64;
    double va[3];
    for (int i = 0; i < 3; ++i) va[i] = rr((double)s->int_vel[i] - s->wind[i], w64); /* :910 */
    double dacc[3];
    if (c->mach_drag && norm3(va, w64) > rr(1e-6, w64)) {                        /* :912-917 */
        double f[3];
        mach_drag_force(c, s, va, w64, rho, sos, 1.0, f);
        for (int i = 0; i < 3; ++i) dacc[i] = rr(f[i] / rr(500.0, w64), w64);
    } else simple_drag_accel(c, va, w64, rho, 500.0, dacc);                      /* :920-921 */
    const double grav[3] = {0.0, 0.0, F(-9.81)};
    double acc[3];
    for (int i = 0; i < 3; ++i) acc[i] = rr(rr(tacc[i] + dacc[i], w64) + grav[i], w64); /* :924 */
    if (c->validation) nan_guard(acc, 50.0);
    for (int i = 0; i < 3; ++i) {                                                /* :933-934 */
        double dv = w64 ? acc[i] * dt : F(acc[i] * dtf);
        s->int_vel[i] = (float)F((double)s->int_vel[i] + dv);
        s->int_pos[i] = (float)F((double)s->int_pos[i] + F((double)s->int_vel[i] * dtf));
    }
    double wn = norm3(ang, 0);                                                   /* :940-956 */
    double angle = F(wn * dtf);
    if (angle > F(1e-6)) {
        double half = F(angle / 2.0);
        double ch = F(cos(half)), sh = F(sin(half));
        double q1[4] = {ch, F(F(ang[0] / wn) * sh), F(F(ang[1] / wn) * sh), F(F(ang[2] / wn) * sh)};
        double q2[4] = {s->int_quat[0], s->int_quat[1], s->int_quat[2], s->int_quat[3]};
        double w1 = q1[0], x1 = q1[1], y1 = q1[2], z1 = q1[3], w2 = q2[0], x2 = q2[1], y2 = q2[2], z2 = q2[3];
        double r[4];                                                             /* :1322-1331 */
        r[0] = F(F(F(F(w1 * w2) - F(x1 * x2)) - F(y1 * y2)) - F(z1 * z2));
        r[1] = F(F(F(F(w1 * x2) + F(x1 * w2)) + F(y1 * z2)) - F(z1 * y2));
        r[2] = F(F(F(F(w1 * y2) - F(x1 * z2)) + F(y1 * w2)) + F(z1 * x2));
        r[3] = F(F(F(F(w1 * z2) + F(x1 * y2)) - F(y1 * x2)) + F(z1 * w2));
        double n = F(sqrt(F(F(r[0] * r[0]) + F(r[1] * r[1]) + F(r[2] * r[2]) + F(r[3] * r[3]))));
        for (int i = 0; i < 4; ++i) s->int_quat[i] = (float)F(r[i] / n);
    }

    /* ---- _update_missile_state  environment.py:1069-1117 (every ACTIVE missile of a volley, :631-636) ---- */
    const int K = c->volley_mode ? c->volley_size : 1;
    for (int km = 0; km < K; ++km) {
        float *mpos = c->volley_mode ? s->v_pos[km] : s->mis_pos, *mvel = c->volley_mode ? s->v_vel[km] : s->mis_vel;
        const double *ez = km == 0 ? nz : nz + 20 + 3 * (km - 1);   /* this missile's evasion draws */
        if (c->volley_mode && !s->v_active[km]) continue;
        double malt = mpos[2] > 0.0f ? (double)mpos[2] : 0.0;
        double mrho = 1.225, msos = 343.0;
        if (c->atmosphere) atmosphere(malt, s->T0, &mrho, &msos);
        double mva[3], md[3];
        for (int i = 0; i < 3; ++i) mva[i] = rr((double)mvel[i] - s->wind[i], w64);
        if (c->mach_drag && norm3(mva, w64) > rr(1e-6, w64)) {
            double f[3];
            mach_drag_force(c, s, mva, w64, mrho, msos, 2.0, f);
            double ratio = (0.3 * 1.5) / 0.3;                                    /* :1090,:1095 */
            for (int i = 0; i < 3; ++i) md[i] = rr(rr(f[i] * rr(ratio, w64), w64) / rr(1000.0, w64), w64);
        } else simple_drag_accel(c, mva, w64, mrho, 1000.0, md);
        double macc[3];
        for (int i = 0; i < 3; ++i) {
            double ev = c->evasion ? ez[i] * 2.0 : 0.0;                          /* :1103-1105 */
            macc[i] = rr(md[i] + grav[i], w64) + ev;                             /* :1108 -> float64 */
        }
        if (c->validation) nan_guard(macc, 20.0);
        for (int i = 0; i < 3; ++i) {                                            /* :1116-1117 */
            mvel[i] = (float)F((double)mvel[i] + macc[i] * dt);
            mpos[i] = (float)F((double)mpos[i] + F((double)mvel[i] * dtf));
        }
    }

    /* ---- _update_wind  environment.py:1119-1129, physics_models.py:351-387 ---- */
    if (c->enhanced_wind) {
        double walt = s->int_pos[2] > 0.0f ? (double)s->int_pos[2] : 0.0;
        double prof;                                                             /* :304-327 */
        if (walt <= F(10.0)) prof = 1.0;
        else if (walt <= F(c->boundary_layer_height)) prof = (double)powf((float)F(walt / F(10.0)), (float)0.143);
        else prof = F(pow(c->boundary_layer_height / 10.0, 0.143));
        double w[3];
        for (int i = 0; i < 3; ++i) w[i] = F(F(c->base_wind[i]) * F(prof));
        double ti;                                                               /* :329-349 */
        if (walt <= F(10.0)) ti = F(c->turbulence_intensity * 2.0);
        else if (walt <= F(c->boundary_layer_height))
            ti = F(F(c->turbulence_intensity) * F(1.0 - F(F(walt / F(c->boundary_layer_height)) * F(0.7))));
        else ti = F(c->turbulence_intensity * 0.3);
        if (ti > 0) {                                                            /* :370-378 */
            double scale = F(ti * norm3(w, 0));
            double lp = 1.0 - exp(-dt / 0.1);
            for (int i = 0; i < 3; ++i) w[i] = F(w[i] + (0.0 + scale * nz[3 + i]) * lp);
        }
        if (nz[6] < 0.001) {                                                     /* :381-385 */
            double g[3] = {nz[7], nz[8], nz[9]};
            double gn = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]) + 1e-6;
            double gmag = c->gust_scale * nz[10];
            for (int i = 0; i < 3; ++i) w[i] = F(w[i] + (g[i] / gn) * gmag);
        }
        for (int i = 0; i < 3; ++i) s->wind[i] = w[i];
        s->wind_is64 = 0;
    } else if (c->wind_variability > 0) {                                        /* :1127-1129 */
        for (int i = 0; i < 3; ++i) {
            double change = nz[3 + i] * c->wind_variability;
            double t1 = s->wind_is64 ? 0.95 * s->wind[i] : F(F(0.95) * s->wind[i]);
            s->wind[i] = t1 + 0.05 * (F(c->base_wind[i]) + change);
        }
        s->wind_is64 = 1;
    }

    /* ---- volley: priority missile = closest active one, first wins ties, missile 0 if none is active (:236-267,
     * :643-650); from here on it is `self.missile_state` (observation, reward alignment, next step's LOS frame) ---- */
    double vdist[ORC_MAX_VOLLEY] = {0, 0, 0, 0};
    if (c->volley_mode) {
        int sel = -1;
        double best = 0.0;
        for (int k = 0; k < K; ++k) {
            double r[3];
            for (int i = 0; i < 3; ++i) r[i] = F((double)s->v_pos[k][i] - (double)s->int_pos[i]);
            vdist[k] = norm3(r, 0);
            if (s->v_active[k] && (sel < 0 || vdist[k] < best)) { sel = k; best = vdist[k]; }
        }
        s->prio = sel < 0 ? 0 : sel;
        for (int i = 0; i < 3; ++i) { s->mis_pos[i] = s->v_pos[s->prio][i]; s->mis_vel[i] = s->v_vel[s->prio][i]; }
    }
    /* ---- intercept / termination  environment.py:657-814 ---- */
    double rel[3];
    for (int i = 0; i < 3; ++i) rel[i] = F((double)s->mis_pos[i] - (double)s->int_pos[i]);
    double distance = norm3(rel, 0);
    int intercepted;
    if (c->volley_mode) {                                                        /* :661-692 */
        const double thr = c->proximity_fuze ? F(c->proximity_kill_radius) : F(c->intercept_radius);
        intercepted = 0;
        for (int k = 0; k < K; ++k) {
            if (!s->v_active[k]) continue;
            if (vdist[k] < (double)s->v_min[k]) s->v_min[k] = (float)vdist[k];
            if (vdist[k] < thr) { intercepted = 1; s->n_intercepted += 1; s->v_active[k] = 0; }
        }
        int any = 0;
        distance = 0.0;                                                          /* :691: 0.0 when nothing is left */
        for (int k = 0; k < K; ++k)
            if (s->v_active[k] && (!any || vdist[k] < distance)) { distance = vdist[k]; any = 1; }
    } else if (c->proximity_fuze) intercepted = distance < F(c->proximity_kill_radius);  /* :700-703 */
    else intercepted = distance < F(c->intercept_radius);
    if (distance < (double)s->min_distance) s->min_distance = (float)distance;   /* :706 */
    if (intercepted && !s->crossed) s->crossed = 1;                              /* :709-710 */
    int fuze = 0;
    if (c->proximity_fuze && (double)s->min_distance < F(c->proximity_kill_radius)) { fuze = 1; intercepted = 1; } /* :715-717 */
    int terminated = 0, truncated = 0, hit_target = 0;
    int ground = s->mis_pos[2] <= 0.0f;
    double gd2[3] = {F((double)s->mis_pos[0] - F(c->target_pos[0])), F((double)s->mis_pos[1] - F(c->target_pos[1])), 0.0};
    int near_target = F(sqrt(F(F(gd2[0] * gd2[0]) + F(gd2[1] * gd2[1])))) < F(500.0);
    if (c->volley_mode) {                                                        /* :724-748 */
        int all_inactive = 1;
        for (int k = 0; k < K; ++k) {
            if (s->v_pos[k][2] <= 0.0f) {                                        /* on the ground: neutralised, whoever it was */
                s->v_active[k] = 0;
                double g0 = F((double)s->v_pos[k][0] - F(c->target_pos[0])), g1 = F((double)s->v_pos[k][1] - F(c->target_pos[1]));
                if (F(sqrt(F(F(g0 * g0) + F(g1 * g1)))) < F(500.0)) hit_target = 1;
            }
            if (s->v_active[k]) all_inactive = 0;
        }
        if (all_inactive) terminated = 1;
        else if (fuze) terminated = 1;
    } else if (c->precision_mode) {                                              /* :752-767 */
        if (ground) { terminated = 1; if (near_target) hit_target = 1; }
    } else {                                                                     /* :769-786 */
        if (intercepted) terminated = 1;
        else if (fuze) terminated = 1;
        else if (ground) { terminated = 1; if (near_target) hit_target = 1; }
    }
    if (s->int_pos[2] < 0.0f) terminated = 1;                                    /* :789-811 */
    else if (s->fuel <= 0.0f) terminated = 1;
    else if (s->steps > 1000) {
        if (distance > (double)s->last_distance) s->worsening += 1;
        else s->worsening = s->worsening - 5 > 0 ? s->worsening - 5 : 0;
        s->last_distance = (float)distance;
        if (s->worsening > 500 && distance > F(2500.0)) terminated = 1;
    }
    if (s->steps >= c->max_steps) truncated = 1;                                 /* :813-814 */

    /* ---- observation (:817) ---- */
    observe(c, s, nz + 11, out->obs);

    /* ---- reward  environment.py:1131-1320 ---- */
    double reward = 0.0;
    if (c->precision_mode) {
        if (terminated) {                                                        /* :1155-1201 */
            double md = s->min_distance;
            if (s->crossed) {
                double rad = c->intercept_radius;
                reward = 3000.0;
                if (md < F(rad)) {
                    double ir = F(F(F(rad) - md) / F(rad));
                    reward = F(reward + F(ir * F(1000.0)));
                }
                reward = F(reward + F((double)ref_np_expf((float)F(-md / F(25.0))) * F(500.0)));
                reward = F(reward + F((double)ref_np_expf((float)F(-md / F(10.0))) * F(1000.0)));
                reward = F(reward + F((double)ref_np_expf((float)F(-md / F(3.0))) * F(500.0)));
                reward = F(reward + F((c->max_steps - s->steps) * 0.3));
            } else {
                reward = F(-md * F(0.5));
                if (-2000.0 > reward) reward = -2000.0;
                if (hit_target) reward = F(reward - 1000.0);
                else if (s->int_pos[2] < 0.0f) reward = F(reward - 500.0);
                else if (s->fuel <= 0.0f) reward = F(reward - 300.0);
            }
        } else {                                                                 /* :1203-1271 */
            double delta = F((double)s->prev_distance - distance);
            double cv = F(delta / dtf);
            reward = F(clipd(F(cv / F(100.0)), -0.5, 2.0) * F(0.5));
            if (distance < F(50.0)) {
                reward = F(reward + F(delta * F(5.0)));
                reward = F(reward + F((double)ref_np_expf((float)F(-distance / F(10.0))) * F(1.0)));
            } else if (distance < F(150.0)) reward = F(reward + F(delta * F(3.0)));
            else if (distance < F(500.0)) reward = F(reward + F(delta * F(1.5)));
            else reward = F(reward + F(delta * F(0.8)));
            double iv[3] = {s->int_vel[0], s->int_vel[1], s->int_vel[2]};
            double isp = norm3(iv, 0);
            if (isp > F(1.0) && distance > F(10.0)) {                            /* :1241-1250 */
                double lu[3], vu[3];
                for (int i = 0; i < 3; ++i) { lu[i] = F(rel[i] / distance); vu[i] = F(iv[i] / isp); }
                reward = F(reward + F(dot3(vu, lu, 0) * F(0.3)));
            }
            if (c->obs_mode == 2) reward = F(reward + F(los_a0 * F(0.4)));      /* :1256-1264 */
            reward = F(reward - F(0.2));
            s->prev_distance = (float)distance;
        }
    } else if (intercepted) {                                                    /* :1274-1282 */
        reward = 5000.0 + (c->max_steps - s->steps) * 0.5;
    } else if (terminated) {                                                     /* :1284-1296 */
        reward = F(-distance * F(0.5));
        if (-2000.0 > reward) reward = -2000.0;
        if (hit_target) reward = F(reward - 1000.0);
        else if (s->int_pos[2] < 0.0f) reward = F(reward - 500.0);
        else if (s->fuel <= 0.0f) reward = F(reward - 300.0);
    } else {                                                                     /* :1298-1320 */
        double delta = F((double)s->prev_distance - distance);
        double cv = F(delta / dtf);
        reward = F(clipd(F(cv / F(100.0)), -0.5, 2.0) * F(0.3));
        if (distance < F(200.0)) reward = F(reward + F(delta * F(2.0)));
        else if (distance < F(500.0)) reward = F(reward + F(delta * F(1.0)));
        else reward = F(reward + F(delta * F(0.5)));
        reward = F(reward - F(0.5));
        s->prev_distance = (float)distance;
    }
    out->reward = reward;
    out->terminated = terminated; out->truncated = truncated; out->intercepted = intercepted;
    out->hit_target = hit_target; out->fuze_triggered = fuze; out->clamped = clamped;
    out->distance = (float)distance; out->min_distance = s->min_distance;
    out->fuel_used = (float)s->total_fuel_used; out->fuel_remaining = s->fuel;   /* :833-834 */
    if (c->volley_mode) {                                                        /* :846-847 */
        out->missiles_intercepted = s->n_intercepted;
        out->missiles_remaining = 0;
        for (int k = 0; k < K; ++k) out->missiles_remaining += s->v_active[k] != 0;
    } else {
        out->missiles_intercepted = intercepted ? 1 : 0;
        out->missiles_remaining = intercepted ? 0 : 1;
    }
}

/* ------------------------------------------------------------------------------------------
 * batched helpers
 * ---------------------------------------------------------------------------------------- */
void orc_reset_batch(const orc_config *cfg, orc_state *st, int32_t n, const double *noise, float *obs) {
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < n; ++i) orc_reset(cfg, st + i, noise + (int64_t)i * ORC_RESET_SLOTS, obs + (int64_t)i * 26);
}
void orc_step_batch(const orc_config *cfg, orc_state *st, int32_t n, const float *actions, const double *step_noise,
                    const double *reset_noise, orc_out *out, float *terminal_obs, int32_t auto_reset) {
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < n; ++i) {
        orc_step(cfg, st + i, actions + (int64_t)i * 6, step_noise + (int64_t)i * ORC_STEP_SLOTS, out + i);
        if (auto_reset && (out[i].terminated || out[i].truncated)) {
            memcpy(terminal_obs + (int64_t)i * 26, out[i].obs, 26 * sizeof(float));
            orc_reset(cfg, st + i, reset_noise + (int64_t)i * ORC_RESET_SLOTS, out[i].obs);
        }
    }
}
int32_t orc_sizeof_state(void) { return (int32_t)sizeof(orc_state); }
int32_t orc_sizeof_config(void) { return (int32_t)sizeof(orc_config); }
int32_t orc_sizeof_out(void) { return (int32_t)sizeof(orc_out); }


/* ------------------------------------------------------------------------------------------
 * ref_math.h checks (tests/test_ref_math.py): the restated glibc powf against this host's powf over a range of
 * float32 bit patterns; the restated numpy exp kernel in batch form (compared with np.exp by the test).
 * ---------------------------------------------------------------------------------------- */
long orc_check_powf(uint32_t lo_bits, uint32_t hi_bits, float y, uint32_t *first_bad) {
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t u = (int64_t)lo_bits; u <= (int64_t)hi_bits; ++u) {
        uint32_t ub = (uint32_t)u;
        float x;
        memcpy(&x, &ub, 4);
        float a = powf(x, y), b = ref_powf(x, y);
        if (memcmp(&a, &b, 4) != 0) {
            bad += 1;
            if (first_bad) *first_bad = ub;
        }
    }
    return bad;
}
void orc_ref_powf_batch(const float *x, float y, float *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = ref_powf(x[i], y);
}
void orc_np_expf_batch(const float *x, float *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = ref_np_expf(x[i]);
}
