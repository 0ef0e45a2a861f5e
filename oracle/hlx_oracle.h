/*
 * hlx_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's per-environment step/reset
 * (RomanSlack/Hlynr_Intercept, rl_system/environment.py, core.py, physics_models.py,
 * physics_randomizer.py).  It exists to check the HIP kernels: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product path
 * (hlynr_intercept_amd/) never links, imports or calls anything in this directory.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py replays every fixture under
 * tests/golden/ (captured by tests/golden/make_golden.py from the reference itself,
 * numpy 2.2.6) through this code.
 *
 * Random variates are explicit inputs ("noise slots", unit draws U(0,1) / N(0,1) / Exp(1));
 * the slot layout is the one documented in tests/golden/make_golden.py and include/hlx.h.
 */
#ifndef HLX_ORACLE_H
#define HLX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_STEP_SLOTS 32   /* 0-19 as documented; 20+3(k-1).. evasion normals of volley missile k >= 1 */
#define ORC_RESET_SLOTS 48  /* 0-31 as documented; 32+4(k-1).. spawn uniforms (position x3, speed) of volley missile k >= 1 */
#define ORC_MAX_VOLLEY 4
#define ORC_MAX_DELAY 10
#define ORC_RING_CAP (ORC_MAX_DELAY + 1)

/* Python floats of the reference are doubles here; what is float32 there is rounded at use. */
typedef struct {
    double dt, max_range, max_velocity;
    int32_t max_steps;
    double target_pos[3];
    int32_t mis_spawn_spherical;
    double mis_pos_lo[3], mis_pos_hi[3];
    double mis_radius[2], mis_azimuth_deg[2], mis_elevation_deg[2], mis_speed[2];
    double int_pos_lo[3], int_pos_hi[3], int_vel_lo[3], int_vel_hi[3];
    int32_t int_vel_toward_missile;
    double int_speed[2];
    int32_t atmosphere, mach_drag, enhanced_wind, thrust_lag, domain_randomization, validation, evasion;
    double subsonic_mach, supersonic_mach, transonic_peak_multiplier, supersonic_multiplier;
    double base_wind[3], wind_variability, boundary_layer_height, turbulence_intensity, gust_scale, thrust_tau;
    double dr_variations[13];
    int32_t precision_mode, proximity_fuze;
    double proximity_kill_radius;
    double radar_quality, radar_range;
    int32_t onboard_delay;
    int32_t ground_enabled;
    double ground_pos[3], ground_max_range, ground_min_elev, ground_max_elev;
    double ground_range_accuracy, ground_velocity_accuracy, ground_base_quality;
    double max_datalink_range, datalink_packet_loss;
    int32_t ground_delay;
    double weather_factor;
    int32_t obs_mode; /* 0 world, 1 body, 2 los */
    int32_t volley_mode, volley_size; /* environment.py:42-43: K missiles per episode (K <= ORC_MAX_VOLLEY) */
    /* per-vec-step curriculum scalars (host evaluates the schedules) */
    double intercept_radius, beam_width_deg, onboard_reliability, ground_reliability;
} orc_config;

typedef struct {
    float int_pos[3], int_vel[3], int_quat[4];
    float fuel;
    float thrust_actual[3];
    float mis_pos[3], mis_vel[3];
    double wind[3];
    int32_t wind_is64;
    int32_t steps;
    float prev_distance, min_distance, last_distance;
    int32_t worsening, crossed;
    /* Kalman filter (core.py:12-133) */
    int32_t kf_init, kf_x_is64;
    double kf_x[6];
    float kf_P[36];
    /* delay rings, logical order oldest -> newest (core.py:147-223) */
    int32_t on_delay, on_count, on_len;
    double on_ring[ORC_RING_CAP][3];
    int32_t on_det[ORC_RING_CAP];
    int32_t g_count, g_len;
    double g_ring[ORC_RING_CAP][7]; /* rel_pos3, rel_vel3, quality */
    int32_t g_pos_is64[ORC_RING_CAP];
    /* per-env physics constants touched by domain randomisation */
    double T0, base_cd, transonic_peak;
    double total_fuel_used;
    int32_t structure_violations; /* KF covariance left the 3x(2x2) block structure / S not diagonal */
    /* volley mode (environment.py:44, 370-373): every missile of the volley; mis_pos / mis_vel above are the
     * reference's `self.missile_state`, i.e. the entry `prio` of this list (re-selected every step, :643-650) */
    float v_pos[ORC_MAX_VOLLEY][3], v_vel[ORC_MAX_VOLLEY][3];
    int32_t v_active[ORC_MAX_VOLLEY];
    float v_min[ORC_MAX_VOLLEY];        /* missile_min_distances */
    int32_t prio, n_intercepted;        /* index of self.missile_state; len(intercepted_missile_indices) */
} orc_state;

typedef struct {
    float obs[26];
    double reward;
    int32_t terminated, truncated, intercepted, hit_target, fuze_triggered, clamped;
    float distance, min_distance;
    int32_t missiles_intercepted, missiles_remaining;   /* info (:846-847) */
    float fuel_used, fuel_remaining;                    /* info['fuel_used'] = total_fuel_used (:834, :886), info['fuel_remaining'] (:833) */
} orc_out;

/* constructor state (T0, drag constants, onboard delay): call once per env before the first reset */
void orc_init(const orc_config *cfg, orc_state *st, int32_t n);
/* reset(): environment.py:353-603.  `noise` = ORC_RESET_SLOTS unit draws. */
void orc_reset(const orc_config *cfg, orc_state *st, const double *noise, float *obs26);
/* step(): environment.py:605-859.  `noise` = ORC_STEP_SLOTS unit draws. */
void orc_step(const orc_config *cfg, orc_state *st, const float *action6, const double *noise, orc_out *out);

/* Batched helpers (array-of-struct state; OpenMP over envs when built with -fopenmp).
 * step_batch applies VecEnv auto-reset semantics: on done, terminal obs is written to
 * terminal_obs[i] and the env is reset with reset_noise[i]. */
void orc_reset_batch(const orc_config *cfg, orc_state *st, int32_t n, const double *noise /*[n][ORC_RESET_SLOTS]*/,
                     float *obs /*[n][26]*/);
void orc_step_batch(const orc_config *cfg, orc_state *st, int32_t n, const float *actions /*[n][6]*/,
                    const double *step_noise /*[n][ORC_STEP_SLOTS]*/, const double *reset_noise /*[n][ORC_RESET_SLOTS]*/,
                    orc_out *out /*[n]*/, float *terminal_obs /*[n][26]*/, int32_t auto_reset);
int32_t orc_sizeof_state(void);
int32_t orc_sizeof_config(void);
int32_t orc_sizeof_out(void);

#ifdef __cplusplus
}
#endif
#endif
