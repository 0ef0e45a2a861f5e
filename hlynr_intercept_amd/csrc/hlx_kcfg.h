// hlx_kcfg.h -- the kernel's per-configuration constants and how they derive from an hlx_config.
// Plain C++ (no HIP): shared by the library (hlx_host.inc), the kernels (hlx_kargs.h) and the build-time generator of the
// baked constant tables (hlx_bake_gen.cpp).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/hlx.h"

namespace hlx {

// Per-step ("hot") constants: ride in the kernel-argument block and end up in SGPRs.
struct KCfg {
    uint32_t flags;
    int32_t max_steps, g_delay, o_delay, o_cap;
    int32_t volley_k;         // missiles per episode (volley mode), else 0
    float dt;                 // F(dt)
    double dt64, inv_dtf;     // dt ; 1 / (double)F(dt)
    float max_range, max_velocity, inv_max_range, inv_max_velocity;
    float target[3];
    float subsonic, supersonic, mach_span, peak_m1, cd_super;
    double super_mult;
    float base_wind[3];
    double wind_var;
    float bl_height, bl_prof, ti_low, ti_mid, ti_high;
    double turb_lp, gust_scale, inv_tau;
    float kill_radius, radar_quality, radar_range, inv_radar_range;
    double radar_quality64;
    float ground_pos[3], g_max_range, inv_g_max_range, g_base_q, max_datalink, inv_max_datalink, weather;
    float sin_min_elev, sin_max_elev;   // elevation window as sines (asin is monotonic)
    double g_range_acc, g_vel_acc, packet_loss;
    float q11, q12, q22;      // Kalman process noise (core.py:34-42, q = 5^2)
};
// Spawn / domain-randomisation ("cold") constants: only finished environments read them, so they live
// in device memory behind a pointer instead of occupying ~90 SGPRs of every wave.
struct KCold {
    double mis_lo[3], mis_span[3], mis_radius[2], mis_az[2], mis_el[2], mis_speed[2];
    double int_lo[3], int_span[3], ivel_lo[3], ivel_span[3], int_speed[2];
    double dr_var[5];
};


// The reference decides "inside the beam" / "inside the elevation window" by comparing arccos / arcsin of a float32
// argument with an angle (core.py:546-553, :401-406).  Both functions are monotone, so the decision is a comparison of the
// ARGUMENT with the float32 at which the host libm's function crosses the angle -- found here by bisection over float32
// bit patterns, with the same acosf / asinf the oracle (and numpy's scalar path) calls.  The kernel forms the argument with
// the reference's float32 operations, hence takes the reference's decision at every input, boundary cases included.
inline int64_t f2ord(float f) { int32_t i; memcpy(&i, &f, 4); return i < 0 ? (int64_t)INT32_MIN - (int64_t)i : (int64_t)i; }
inline float ord2f(int64_t o) { int32_t i = (int32_t)(o < 0 ? (int64_t)INT32_MIN - o : o); float f; memcpy(&f, &i, 4); return f; }
// pred is true on a (possibly empty) lower part of [-1, 1] and false above it: the smallest float32 where it is false (2.0f if none)
template <class P> float first_false(P pred) {
    if (pred(1.0f)) return 2.0f;
    if (!pred(-1.0f)) return -1.0f;
    int64_t lo = f2ord(-1.0f), hi = f2ord(1.0f);   // pred(lo) true, pred(hi) false
    while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (pred(ord2f(mid))) lo = mid; else hi = mid;
    }
    return ord2f(hi);
}
inline float beam_cos_threshold(double half_beam) {       // beam_angle > half_beam  <=>  c < threshold
    return first_false([half_beam](float c) { return (double)acosf(c) > half_beam; });
}


inline void build_kcfg(const hlx_config& c, KCfg& k, KCold& d) {
    // Derives the kernel's float32 / float64 constants from the reference's Python-float parameters
    // with the roundings numpy's promotion rules imply (a Python float meeting a float32 operand is
    // rounded to float32 first; products of two Python floats are formed in float64 and then rounded).
    memset(&k, 0, sizeof k);
    memset(&d, 0, sizeof d);
    k.flags = c.flags;
    k.max_steps = c.max_steps;
    k.volley_k = (c.flags & HLX_F_VOLLEY) ? c.volley_size : 0;
    k.dt = (float)c.dt; k.dt64 = c.dt; k.inv_dtf = 1.0 / (double)(float)c.dt;
    k.max_range = (float)c.max_range; k.max_velocity = (float)c.max_velocity;
    k.inv_max_range = (float)(1.0 / c.max_range); k.inv_max_velocity = (float)(1.0 / c.max_velocity);
    for (int i = 0; i < 3; ++i) {
        k.target[i] = (float)c.target_pos[i];
        d.mis_lo[i] = c.mis_pos_lo[i]; d.mis_span[i] = c.mis_pos_hi[i] - c.mis_pos_lo[i];
        d.int_lo[i] = c.int_pos_lo[i]; d.int_span[i] = c.int_pos_hi[i] - c.int_pos_lo[i];
        d.ivel_lo[i] = c.int_vel_lo[i]; d.ivel_span[i] = c.int_vel_hi[i] - c.int_vel_lo[i];
        k.base_wind[i] = (float)c.base_wind[i];
        k.ground_pos[i] = (float)c.ground_pos[i];
    }
    // {lo, span}
    d.mis_radius[0] = c.mis_radius[0]; d.mis_radius[1] = c.mis_radius[1] - c.mis_radius[0];
    d.mis_az[0] = c.mis_azimuth_deg[0]; d.mis_az[1] = c.mis_azimuth_deg[1] - c.mis_azimuth_deg[0];
    d.mis_el[0] = c.mis_elevation_deg[0]; d.mis_el[1] = c.mis_elevation_deg[1] - c.mis_elevation_deg[0];
    d.mis_speed[0] = c.mis_speed[0]; d.mis_speed[1] = c.mis_speed[1] - c.mis_speed[0];
    d.int_speed[0] = c.int_speed[0]; d.int_speed[1] = c.int_speed[1] - c.int_speed[0];
    k.subsonic = (float)c.subsonic_mach; k.supersonic = (float)c.supersonic_mach;
    k.mach_span = (float)(c.supersonic_mach - c.subsonic_mach);                   // physics_models.py:213-214
    k.peak_m1 = (float)(c.transonic_peak_multiplier - 1.0);                        // physics_models.py:215 (peak - 1.0) in Python floats
    k.cd_super = (float)(0.3 * c.supersonic_multiplier);                          // physics_models.py:220
    k.super_mult = c.supersonic_multiplier;
    k.wind_var = c.wind_variability;
    k.bl_height = (float)c.boundary_layer_height;
    k.bl_prof = (float)pow(c.boundary_layer_height / 10.0, 0.143);           // physics_models.py:327
    k.ti_low = (float)(c.turbulence_intensity * 2.0);                             // physics_models.py:342
    k.ti_mid = (float)c.turbulence_intensity;                                     // :346
    k.ti_high = (float)(c.turbulence_intensity * 0.3);                            // :349
    k.turb_lp = 1.0 - exp(-c.dt / 0.1);                                      // physics_models.py:375-376
    k.gust_scale = c.gust_scale;
    k.inv_tau = 1.0 / (double)(float)c.thrust_tau;
    for (int i = 0; i < 5; ++i) d.dr_var[i] = c.dr_variations[i];
    k.kill_radius = (float)c.proximity_kill_radius;
    k.radar_quality = (float)c.radar_quality; k.radar_quality64 = c.radar_quality;
    k.radar_range = (float)c.radar_range; k.inv_radar_range = (float)(1.0 / c.radar_range);
    k.g_max_range = (float)c.ground_max_range; k.inv_g_max_range = (float)(1.0 / c.ground_max_range);
    {   // core.py:401-406: elevation < min -> s < sin_min_elev ; elevation > max -> s > sin_max_elev (see first_false above)
        const double lo = c.ground_min_elev, hi = c.ground_max_elev;
        k.sin_min_elev = first_false([lo](float s) { return (double)asinf(s) < lo; });
        const float above = first_false([hi](float s) { return !((double)asinf(s) > hi); });   // smallest s with asinf(s) > max
        k.sin_max_elev = above > 1.0f ? 2.0f : ord2f(f2ord(above) - 1);                          // largest s still inside
    }
    k.g_range_acc = c.ground_range_accuracy; k.g_vel_acc = c.ground_velocity_accuracy;
    k.g_base_q = (float)c.ground_base_quality; k.max_datalink = (float)c.max_datalink_range; k.inv_max_datalink = (float)(1.0 / c.max_datalink_range);
    k.packet_loss = c.datalink_packet_loss; k.weather = (float)c.weather_factor;
    k.g_delay = (c.flags & HLX_F_GROUND) ? c.ground_delay : 0;
    k.o_delay = c.onboard_delay;
    k.o_cap = (c.onboard_delay > 0) ? ((c.flags & HLX_F_DOMAIN_RAND) ? HLX_RING_CAP : c.onboard_delay + 1) : 1;
    const double dt = c.dt, q = 25.0;                                             // core.py:333-334
    k.q11 = (float)(q * pow(dt, 4) / 4.0); k.q12 = (float)(q * pow(dt, 3) / 2.0); k.q22 = (float)(q * dt * dt);
}

}  // namespace hlx
