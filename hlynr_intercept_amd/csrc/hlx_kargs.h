// hlx_kargs.h -- kernel argument block shared by the kernels and the host side of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hlx.h"

namespace hlx {

// internal flag: the kernel reads the feature flags at run time instead of having them folded
constexpr uint32_t KF_DYNAMIC = 1u << 31;
// flags that change generated code (the curriculum flags are host-only)
constexpr uint32_t KF_CODEGEN_MASK = HLX_F_ATMOSPHERE | HLX_F_MACH_DRAG | HLX_F_ENH_WIND | HLX_F_THRUST_LAG |
                                     HLX_F_DOMAIN_RAND | HLX_F_VALIDATION | HLX_F_EVASION | HLX_F_PRECISION |
                                     HLX_F_PROX_FUZE | HLX_F_GROUND | HLX_F_SPHERICAL | HLX_F_TOWARD_MISSILE |
                                     HLX_F_OBS_BODY | HLX_F_OBS_LOS;

// State arena: float4 groups, struct-of-arrays: arena[group][env]
enum : int {
    G_IPOS = 0,  // interceptor position xyz, fuel
    G_IVEL,      // interceptor velocity xyz, prev_distance
    G_QUAT,      // orientation w x y z
    G_MPOS,      // missile position xyz, episode min distance
    G_MVEL,      // missile velocity xyz, last_distance (smart early termination)
    G_WIND,      // wind xyz, packed {steps:13, worsening:13, crossed:1, kf_init:1, onboard_delay:4}
    G_KFP,       // Kalman position estimate xyz, p_pp
    G_KFV,       // Kalman velocity estimate xyz, p_pv
    G_MISC,      // p_vp, p_vv, episode return, sea-level temperature T0
    G_THRUST,    // actual thrust xyz (thrust lag), pad
    G_DR,        // base_cd, transonic peak multiplier (domain randomisation), pad, pad
    N_GROUPS
};

struct KCfg {
    uint32_t flags;
    int32_t max_steps;
    float dt, max_range, max_velocity;
    float target[3];
    float mis_lo[3], mis_span[3], mis_radius[2], mis_az[2], mis_el[2], mis_speed[2];
    float int_lo[3], int_span[3], ivel_lo[3], ivel_span[3], int_speed[2];
    float subsonic, supersonic, peak, super_mult;
    float base_wind[3], wind_var, bl_height, bl_prof, turb, turb_lp, gust_scale, thrust_tau;
    float dr_var[5];
    float kill_radius, radar_quality, radar_range;
    float ground_pos[3], g_max_range, g_min_elev, g_max_elev, g_range_acc, g_vel_acc, g_base_q;
    float max_datalink, packet_loss, weather;
    int32_t g_delay, o_delay, o_cap;
    float q11, q12, q22;  // Kalman process noise (core.py:34-42, q = 5^2)
};

struct KArgs {
    KCfg c;
    float radius, half_beam, on_rel, g_rel;  // curriculum scalars in force for this launch
    float4* arena;
    float4* gring;  // [g_delay+1][2][N]
    float4* oring;  // [o_cap][N]
    int32_t n;
    const float* actions;
    float* obs;
    float* reward;
    uint8_t* term;
    uint8_t* trunc;
    float* terminal_obs;
    int32_t* done_idx;
    int32_t* done_cnt;  // [2], double-buffered by vec-step parity
    hlx_info_soa info;
    const float* step_noise;
    const float* reset_noise;
    const unsigned long long* t_dev;  // optional device-resident clock base (hipGraph replay)
    unsigned long long t_add;
    uint32_t seed_lo, seed_hi;
    long long env_offset;
    const uint8_t* reset_mask;
};

}  // namespace hlx
