/*
 * hlx.h -- C ABI of the MI355X-native batched intercept-environment step.
 *
 * Drop-in boundary for the hot path of RomanSlack/Hlynr_Intercept: the per-environment
 * `InterceptEnvironment.reset()/step()` (rl_system/environment.py:353-603, :605-859) together
 * with the observation builder (rl_system/core.py:511-1032) and the physics evaluators
 * (rl_system/physics_models.py:40-387, rl_system/physics_randomizer.py:137-297), for N
 * environments at once on one GPU.  Plain pointers and sizes only; no torch/numpy types.
 *
 * Conventions
 *  - every `float*`/`uint8_t*`/`int32_t*` I/O argument of hlx_reset/hlx_step/hlx_rollout is a
 *    DEVICE pointer owned by the caller (e.g. a torch ROCm tensor's data_ptr); it must stay valid
 *    until the stream has passed the call.  `stream` is a hipStream_t (NULL = default stream).
 *  - the library owns the struct-of-arrays state arena (one hipMalloc at hlx_create); hlx_step
 *    allocates nothing and never synchronises.  (Not capturable into a replayed hipGraph as is: the vec-step
 *    clock -- Philox counter and delay-ring phase -- is a kernel ARGUMENT that advances with every call, so a
 *    replay would repeat one step's draws; for launch-bound small batches use hlx_set_rollout_fused instead.)
 *  - every function returns HLX_OK (0) or a negative hlx_status; hlx_last_error() gives the
 *    message (thread-local).  HIP errors are mapped, never abort().
 *  - one handle per device; a handle is not thread-safe, distinct handles are independent.
 */
#ifndef HLX_H
#define HLX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HLX_OBS_DIM 26      /* environment.py:192-194  Box(-2, 1, (26,), float32) */
#define HLX_ACT_DIM 6       /* environment.py:195-197  Box(-1, 1, (6,), float32)  */
#define HLX_STEP_SLOTS 32   /* unit random draws one step may consume */
#define HLX_RESET_SLOTS 48  /* unit random draws one reset may consume */
#define HLX_MAX_VOLLEY 4    /* missiles per environment in volley mode (environment.py:42-44; inference.py:744 default 3) */
#define HLX_MAX_DELAY 10    /* physics_randomizer.py:293 clamps the onboard delay to [1, 10] samples */
#define HLX_RING_CAP (HLX_MAX_DELAY + 1)
#define HLX_MAX_STEPS 8191  /* per-episode step counter is packed into 13 bits */

typedef enum hlx_status {
    HLX_OK = 0,
    HLX_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    HLX_ERR_HIP = -2,       /* a HIP runtime call failed (message has hipGetErrorString) */
    HLX_ERR_NOMEM = -3,
    HLX_ERR_STATE = -4      /* call not valid in the handle's current state */
} hlx_status;

/* Step-slot layout (noise injection, parity mode):
 *   0-2 evasion N(0,1)      environment.py:1105        3-5 wind N(0,1)  environment.py:1128 / physics_models.py:372
 *   6 gust U                physics_models.py:381      7-9 gust direction N   :382     10 gust magnitude Exp(1) :384
 *   11 onboard detect U     core.py:563                12 ground detect U     core.py:417
 *   13-15 ground pos N      core.py:425                16-18 ground vel N     core.py:426   19 datalink U core.py:470
 * Reset-slot layout:
 *   0-2 missile position U (box; or radius/azimuth/elevation, spherical) environment.py:398-409
 *   3 missile speed U :415     4-6 interceptor position U :445     7-9 interceptor velocity U :467 (7 = speed U :461)
 *   10 onboard U  11 ground U  12-14 ground pos N  15-17 ground vel N  18 datalink U   (first observation, :570)
 *   19-31 domain-randomisation N x13 in draw order, physics_randomizer.py:165-214
 * Volley mode, missile k = 1..3 (missile 0 uses the slots above): step slots 20+3(k-1) .. +2 evasion N (drawn only
 * while that missile is active, environment.py:632-636); reset slots 32+4(k-1) .. +3 position U x3, speed U
 * (environment.py:389-415, one set of four per missile in spawn order).
 */

typedef enum hlx_flags {
    HLX_F_ATMOSPHERE = 1u << 0,    /* ISA density / speed of sound      physics_models.py:154-177 */
    HLX_F_MACH_DRAG = 1u << 1,     /* Mach-dependent Cd                 physics_models.py:197-264 */
    HLX_F_ENH_WIND = 1u << 2,      /* boundary-layer wind + turbulence  physics_models.py:351-387 */
    HLX_F_THRUST_LAG = 1u << 3,    /* first-order thrust response       environment.py:874-878 */
    HLX_F_DOMAIN_RAND = 1u << 4,   /* per-episode randomisation         physics_randomizer.py:137-297 */
    HLX_F_VALIDATION = 1u << 5,    /* nan_to_num guard                  environment.py:927-930,1111-1113 */
    HLX_F_EVASION = 1u << 6,       /* missile evasion noise             environment.py:1103-1105 */
    HLX_F_PRECISION = 1u << 7,     /* precision_mode reward/termination environment.py:752-767,1151-1271 */
    HLX_F_PROX_FUZE = 1u << 8,     /* proximity fuze                    environment.py:700-717 */
    HLX_F_GROUND = 1u << 9,        /* ground radar station present      core.py:296-320 */
    HLX_F_SPHERICAL = 1u << 10,    /* missile spherical spawn           environment.py:390-406 */
    HLX_F_TOWARD_MISSILE = 1u << 11, /* interceptor velocity_mode       environment.py:452-462 */
    HLX_F_OBS_BODY = 1u << 12,     /* observation_mode body_frame       core.py:870-876 */
    HLX_F_OBS_LOS = 1u << 13,      /* observation_mode los_frame + LOS action transform  core.py:791-868, environment.py:965-1063 */
    HLX_F_USE_CURRICULUM = 1u << 14, /* intercept-radius curriculum     environment.py:223-234 */
    HLX_F_RADAR_CURRICULUM = 1u << 15, /* radar curriculum dict non-empty environment.py:274-351 */
    HLX_F_VOLLEY = 1u << 16,       /* volley_mode: volley_size missiles per episode  environment.py:236-267, 386-439,
                                      470-487, 631-692, 724-748 */
    HLX_F_RADAR_DEBUG = 1u << 17   /* keep what info['radar_debug'] (core.py:650-683) needs: the onboard detection reason
                                      travels through the delay ring and hlx_info_soa.radar_debug may be given.  Selects
                                      the generic kernel variant; the specialised ones carry none of it. */
} hlx_flags;

/* Flat parameter set = the EFFECTIVE values the reference's constructor arrives at
 * (hlynr_intercept_amd/config.py resolves a reference config dict into this).  The reference's
 * parameters are Python floats, i.e. doubles; they stay doubles here so that the library can derive
 * its float32 / float64 kernel constants with the roundings the reference's arithmetic implies. */
typedef struct hlx_config {
    uint32_t flags;
    int32_t max_steps;
    int32_t onboard_delay;          /* samples, 0 = no onboard delay ring */
    int32_t ground_delay;           /* samples, 0 = no ground delay ring */
    double dt, max_range, max_velocity;
    double target_pos[3];
    double mis_pos_lo[3], mis_pos_hi[3];
    double mis_radius[2], mis_azimuth_deg[2], mis_elevation_deg[2], mis_speed[2];
    double int_pos_lo[3], int_pos_hi[3], int_vel_lo[3], int_vel_hi[3], int_speed[2];
    double subsonic_mach, supersonic_mach, transonic_peak_multiplier, supersonic_multiplier;
    double base_wind[3], wind_variability, boundary_layer_height, turbulence_intensity, gust_scale, thrust_tau;
    double dr_variations[13];
    double proximity_kill_radius;
    double radar_quality, radar_range, radar_beam_width;
    double ground_pos[3], ground_max_range, ground_min_elev, ground_max_elev;
    double ground_range_accuracy, ground_velocity_accuracy, ground_base_quality;
    double max_datalink_range, datalink_packet_loss, weather_factor;
    /* curriculum schedules, evaluated host-side by hlx_set_global_step */
    double initial_radius, final_radius, curriculum_steps;
    double rc_beam[4], rc_onboard[4], rc_ground[4], rc_noise[4]; /* {initial, final, start, end} */
    int32_t volley_size;            /* 1..HLX_MAX_VOLLEY, read only with HLX_F_VOLLEY (environment.py:43) */
    int32_t pad1;
} hlx_config;

/* Optional per-step side outputs (device pointers, each may be NULL = not wanted).
 * Mirrors the keys of the reference's `info` dict that callers read (environment.py:829-857). */
typedef struct hlx_info_soa {
    float *distance;          /* [N] info['distance'] */
    float *min_distance;      /* [N] info['min_distance'] */
    float *fuel;              /* [N] info['fuel_remaining'] */
    uint8_t *flags;           /* [N] bit0 intercepted, bit1 missile_hit_target, bit2 proximity_fuze_triggered,
                                     bit3 clamped, bit4 crossed_threshold, bit5 onboard radar detected (delayed),
                                     bit6 ground radar detected, bit7 a delayed onboard sample exists (the onboard delay
                                     line has filled: info['radar_quality'] = the configured quality, else 0.0) */
    float *episode_return;    /* [N] written only for envs that finished this step (Monitor 'r') */
    int32_t *episode_length;  /* [N] written only for envs that finished this step (Monitor 'l') */
    uint8_t *missiles;        /* [N] low nibble info['missiles_intercepted'], high nibble info['missiles_remaining']
                                     (environment.py:846-847) */
    float *interceptor_pos;   /* [3][N] plane-major x|y|z: info['interceptor_pos'] (environment.py:837; read by
                                     train_hrl_pretrain.py:180-198, inference.py:535-560) */
    float *missile_pos;       /* [3][N] plane-major: info['missile_pos'] = the priority missile (environment.py:836) */
    int32_t *steps;           /* [N] info['steps'] (environment.py:838) */
    float *missile_min_distances; /* [HLX_MAX_VOLLEY][N] plane-major, volley mode only: info['missile_min_distances']
                                     (environment.py:848) -- the closest approach to each missile of the volley so far;
                                     planes >= volley_size are not written.  (Outside volley mode the reference's value is
                                     [distance].) */
    float *radar_debug;       /* [8][N] plane-major, what info['radar_debug'] (environment.py:842, core.py:650-683) needs
                                     beyond the positions: planes 0-3 interceptor quaternion w,x,y,z; 4 delayed ground-radar
                                     quality; 5 an int32 bit pattern: bits0-2 the delayed onboard 'detection_reason'
                                     (0 none/detected, 1 out_of_range, 2 outside_beam, 3 poor_signal,
                                     4 sensor_delay_initialization), bit3 the ground radar detected THIS step (before its
                                     delay line); 6 data-link quality, 7 fusion confidence (observation entries 24, 25
                                     of this step).  Requires HLX_F_RADAR_DEBUG (hlx_step fails otherwise).
                                     hlynr_intercept_amd/episode_log.py assembles the dict. */
    float *fuel_used;         /* [N] info['fuel_used'] = the reference's `total_fuel_used` (environment.py:204, 566, 834, 886):
                                     the float32 running sum of the fuel each step of the episode consumed.  The sum is
                                     ENVIRONMENT STATE (hlx_env_state.fuel_used: it lives in the library's arena, travels through
                                     hlx_get_state / hlx_set_state, and advances in every form of the step -- fused and
                                     non-contract rollouts included); this plane receives a copy. */
    float *packed;            /* [3][N][4] dwords -- the nine standard keys above in three 16-byte words per environment, the
                                     layout the step kernel can store with three coalesced 16-byte-per-lane instructions (1 KiB per
                                     wave each) instead of eleven 4-byte and two 1-byte ones:
                                       word 0 [N]: distance, min_distance, fuel (= info['fuel_remaining']), fuel_used      (float32)
                                       word 1 [N]: interceptor_pos x, y, z (float32), steps (int32 bit pattern)
                                       word 2 [N]: missile_pos x, y, z (float32), one uint32: bits 0-7 = `flags` as above,
                                                   bits 8-15 = `missiles` as above, bit 16 terminated, bit 17 truncated
                                     When `packed` is given, distance / min_distance / fuel / fuel_used / flags / missiles /
                                     interceptor_pos / missile_pos / steps above must be NULL (hlx_step fails otherwise); the
                                     finished-environment planes (episode_return, episode_length), missile_min_distances and
                                     radar_debug stay separate.  hlynr_intercept_amd.HlynrVecEnv exposes the same info[...] keys
                                     as strided views of this buffer. */
} hlx_info_soa;

/* Logical per-environment state, array-of-struct, HOST memory: parity injection and checkpointing. */
typedef struct hlx_env_state {
    float int_pos[3], int_vel[3], int_quat[4], fuel;
    float thrust_actual[3];
    float mis_pos[3], mis_vel[3];
    float prev_distance, min_distance, last_distance;
    int32_t steps, worsening, crossed, kf_init;
    int32_t kf_x_is64;                  /* the reference's Kalman state array has become float64 (core.py:112) */
    int32_t episode;                    /* auto-resets this environment has gone through since its seed was set: the counter word of
                                           the NEXT auto-reset's spawn draws is episode + 1 (see hlx_set_episode_pool) */
    double wind[3];                     /* float64 in the reference's simple-wind mode (environment.py:1127-1129) */
    double kf_x[6];
    float kf_P[4];                      /* p_pp, p_pv, p_vp, p_vv : covariance is 3 identical 2x2 blocks */
    int32_t on_delay, on_len;
    float on_ring[HLX_RING_CAP][4];     /* oldest -> newest : rel_pos xyz, then 1 = detected, otherwise 0, or with
                                           HLX_F_RADAR_DEBUG minus the detection_reason code of hlx_info_soa.radar_debug */
    int32_t g_len;
    double g_ring[HLX_RING_CAP][8];     /* oldest -> newest : rel_pos xyz (float64 measurement), quality, rel_vel xyz,
                                           sample-was-a-detection flag */
    /* constants touched by domain randomisation (physics_randomizer.py:258-280).  The reference keeps them as Python
     * floats: the sea-level temperature accumulates over episodes in float64; the drag constants enter float32
     * expressions rounded once, and that is how the kernel holds them: F(base_cd), F(peak multiplier - 1.0),
     * F(base_cd * supersonic_multiplier). */
    double T0;
    float base_cd, transonic_peak_m1, cd_super;
    float ep_return;
    /* volley mode: every missile of the volley; mis_pos / mis_vel above are the reference's `self.missile_state`,
     * i.e. entry `prio` of this list (environment.py:643-650) */
    float v_pos[HLX_MAX_VOLLEY][3], v_vel[HLX_MAX_VOLLEY][3];
    float v_min[HLX_MAX_VOLLEY];        /* missile_min_distances */
    int32_t v_active[HLX_MAX_VOLLEY];
    int32_t prio, n_intercepted;
    float fuel_used;                    /* the reference's `total_fuel_used` (environment.py:204, 566, 886): float32 running sum of the
                                           episode's fuel consumption, 0 at every reset; info['fuel_used'] reports it */
    int32_t pad2;
} hlx_env_state;

typedef struct hlx_env hlx_env;

/* Create N environments on `device`.  `env_id_offset` is the global index of this shard's first
 * environment: the counter-based RNG is keyed by (seed, global env id, vec-step), so results do not
 * depend on how environments are sharded over GPUs.  Envs are NOT reset by create. */
int hlx_create(const hlx_config *cfg, int32_t n_envs, int32_t device, uint64_t seed, int64_t env_id_offset,
               hlx_env **out);
int hlx_destroy(hlx_env *env);

/* environment.py:353 reset().  mask: device uint8[N] (non-zero = reset that env) or NULL = all.
 * obs_out: device float[N][26] (rows of envs that are not reset are left untouched), may be NULL. */
int hlx_reset(hlx_env *env, const uint8_t *mask, float *obs_out, void *stream);
/* ... and reset()'s info with it (environment.py:595-601: missile_pos, interceptor_pos, distance, radar_detected, radar_quality): for the
 * environments being reset, info->packed (the only form accepted here) receives the words a step would write -- distance = min_distance
 * = the spawn distance, full fuel, nothing used, steps 0, both positions, and in the flag byte the detections of the FIRST observation
 * (bits 5-7; every other flag clear), missiles = none intercepted / all in flight.  Other environments' words are left alone. */
int hlx_reset_info(hlx_env *env, const uint8_t *mask, float *obs_out, const hlx_info_soa *info, void *stream);

/* Explicit resets and the random streams.  The spawn draws of an hlx_reset are Philox(seed, global env id, clock | epoch << 48):
 * `epoch` counts the hlx_reset calls this HANDLE has seen at the current clock value (it restarts at 0 whenever a step
 * advances the clock), so that two resets with no step between them start different episodes, as the reference's moving
 * generator does.  The epoch is per handle, not per environment: shards of one job draw what the unsharded batch draws
 * as long as every shard sees the same hlx_reset calls between two steps (masked ones included -- call hlx_reset on a shard
 * even when its slice of the mask is all zero), or the caller sets the epoch itself with hlx_set_reset_epoch (16 bits).
 * (Counter layout: the generator sees 56 bits of the launch's clock word -- the vec-step clock in bits 0-39, i.e. 1.1e12
 * steps, and for reset launches the epoch in bits 40-55.) */
int hlx_set_reset_epoch(hlx_env *env, uint32_t epoch);
uint32_t hlx_get_reset_epoch(const hlx_env *env);

/* Re-key the counter-based RNG (gym's `reset(seed=...)` / SB3's `VecEnv.seed`, scripts/compare_policies.py:150): env i
 * draws from Philox(seed, env_id_offset + i, vec-step clock) from the next launch on.  Follow with hlx_reset to start
 * episodes that depend on the new seed only through (seed, clock).  Successive hlx_reset calls at one clock value draw
 * different episodes (a reset epoch is part of the counter); hlx_set_seed restarts that epoch. */
int hlx_set_seed(hlx_env *env, uint64_t seed);

/* environment.py:605 step() for all N envs, with VecEnv auto-reset: an env that terminates or
 * truncates is reset inside the same launch; `obs` then holds the first observation of the new
 * episode and `terminal_obs[i]` (if not NULL) the last one of the finished episode.
 * actions [N][6], obs [N][26], reward [N], terminated [N], truncated [N];
 * done_idx/n_done (optional): compacted list of finished env indices and its length. */
int hlx_step(hlx_env *env, const float *actions, float *obs, float *reward, uint8_t *terminated,
             uint8_t *truncated, float *terminal_obs, int32_t *done_idx, int32_t *n_done,
             const hlx_info_soa *info, void *stream);

/* T consecutive steps from a pre-supplied action tape [T][N][6], launches issued from C
 * (benchmark / open-loop evaluation path; the reference's equivalent is a `for t: env.step(tape[t])` loop).
 * Outputs of step t go to slot (t % out_slots) of obs [out_slots][N][26],
 * reward/terminated/truncated [out_slots][N] (out_slots >= 1). */
int hlx_rollout(hlx_env *env, const float *actions, int32_t T, int32_t out_slots, float *obs, float *reward,
                uint8_t *terminated, uint8_t *truncated, void *stream);
/* How hlx_rollout issues its steps: 1 (default) = one kernel launch per step, exactly what hlx_step does;
 * k > 1 = fused rollout, up to k steps per launch with the environments' state held in registers between
 * steps (state groups touch HBM once per launch; per step only actions/ring samples in and
 * observation/reward/flags/ring samples out).  Results are bit-identical either way (same Philox keys).
 * Ignored (falls back to 1) while hlx_set_noise buffers are installed. */
int hlx_set_rollout_fused(hlx_env *env, int32_t steps_per_launch);
/* hlx_rollout has no terminal-observation output; with a buffer installed here (device float [N][26], NULL removes it) its
 * one-launch-per-step form writes, at every step, the terminal observation of the environments that finished in it
 * (rows of the others are left untouched), exactly as hlx_step's `terminal_obs` does.  Costs the finished environments'
 * waves a second trip through the observation code (single observation pass); ignored by the
 * fused form. */
int hlx_set_rollout_terminal_obs(hlx_env *env, float *terminal_obs);
/* The same for ALL optional outputs of hlx_step: with these installed, every launch of hlx_rollout's one-launch-per-step
 * form is exactly the launch hlx_step(..., terminal_obs, done_idx, n_done, info, ...) issues -- the form a VecEnv caller
 * gets (SB3 bootstraps from infos[i]['terminal_observation'], the reference's trainers read the info keys:
 * train_flat_ppo.py:292-299, train_hrl_pretrain.py:180-198).  The buffers are the same at every step (each step
 * overwrites them, as successive hlx_step calls with the same arguments would); any of them may be NULL.  The length of
 * the done list of step t is in the done counter (hlx_set_done_counter).  NULL, NULL, NULL removes them.  Ignored by the
 * fused form. */
int hlx_set_rollout_outputs(hlx_env *env, float *terminal_obs, int32_t *done_idx, const hlx_info_soa *info);
/* Where the kernel counts the environments that finished in a step: DEVICE int32[2] owned by the caller (zero-initialised
 * by this call), element (hlx_vec_steps() & 1) holds the count of the latest step once its launch has completed; the
 * launch of a step also re-arms the other element for the next one.  With caller storage installed, hlx_step's `n_done`
 * may point at that element (or be read from it) and no copy is enqueued; any other `n_done` pointer costs one 4-byte
 * device-to-device copy behind the launch.  NULL returns to the library's own storage. */
int hlx_set_done_counter(hlx_env *env, int32_t *counter2);

/* environment.py:269 set_training_step_count(): O(1) host-side; evaluates the curriculum
 * schedules (environment.py:223-234, :274-351) and the result rides along as kernel arguments. */
int hlx_set_global_step(hlx_env *env, int64_t global_step);
/* out[5] = {intercept_radius, beam_width_deg, onboard_reliability, ground_reliability, noise_level} */
int hlx_get_curriculum(hlx_env *env, double out[5]);

/* Parity mode: take the random draws from caller-supplied DEVICE arrays instead of Philox.
 * step_noise [HLX_STEP_SLOTS][N], reset_noise [HLX_RESET_SLOTS][N] (slot-major), FLOAT64 unit draws
 * (the reference draws float64 variates; replaying them exactly needs all 53 bits).  NULL restores Philox. */
int hlx_set_noise(hlx_env *env, const double *step_noise, const double *reset_noise);
/* Write the Philox draws of vec-step clock (current + clock_offset) into slot-major device arrays
 * (every slot, whether or not it ends up being consumed): clock_offset = 1 -> what the NEXT hlx_step
 * (and its auto-resets) will draw; clock_offset = 0 -> what an hlx_reset issued now will draw.
 * Philox produces float32 variates; they are stored widened to float64. */
int hlx_fill_noise(hlx_env *env, double *step_noise, double *reset_noise, int32_t clock_offset, void *stream);

/* Logical state export / injection (host array of n_envs hlx_env_state).  Synchronises. */
int hlx_get_state(hlx_env *env, hlx_env_state *host_out);
int hlx_set_state(hlx_env *env, const hlx_env_state *host_in);

/* Kernel timing with HIP events recorded on the launch stream: one pair around every hlx_step launch; for an hlx_rollout
 * call of T one-step launches, one pair from behind the first launch to behind the last (T - 1 launches back to back,
 * without the host's launch latency ahead of the first); for the fused form one pair around the call. */
int hlx_profile(hlx_env *env, int32_t enable);
/* total elapsed ms between the event pairs and the number of step launches they covered
 * (synchronises, then clears) */
int hlx_profile_read(hlx_env *env, double *total_ms, int64_t *launches);

/* Auto-resets and the random streams; the next-episode pool.  The spawn and first-observation draws of an AUTO-reset (an
 * environment that finishes inside hlx_step / hlx_rollout) are Philox(seed, global env id, k) with k = 1, 2, ... the index of
 * the episode among that environment's auto-resets since its seed was set (counter words {id, id >> 32, k, 0xFFFFFF00 | stream}:
 * a high word no clock value reaches) -- not the clock: the episode an environment starts next depends on nothing the steps
 * before it decide, so the library prepares it ahead of time.  Every `interval` step launches one extra launch computes spawn
 * state + first observation for the environments that have used their prepared episode since the last one (whole waves of
 * work), and a finished environment copies its entry inside the step launch instead of computing it there -- which is what
 * used to keep every step launch open.  An environment that finishes again before its entry has been renewed computes it
 * on the spot: same draws, same arithmetic, same bits; the pool changes when work is done, never a result.  hlx_reset,
 * hlx_set_seed (which restarts k) and hlx_set_state renew every entry before the next step launch; for curriculum updates see
 * hlx_get_episode_pool_stats below.  interval: > 0 step launches between fills, 0 = no pool,
 * -1 = the default (128).  hlx_get_episode_pool returns the interval in force (0 = off).  Prepared episodes are used by the
 * lone-wave load schedule only (hlx_set_load_schedule 2: batches of at most one wave per SIMD, where the stragglers of a launch
 * are exposed); under the other schedules every auto-reset is computed in place, from the same draws, and no fill is launched.
 * hlx_get_episode_pool_misses: auto-resets computed inside step launches so far because the prepared episode was absent or stale,
 *   pool on (synchronises; diagnostics).
 * hlx_get_episode_pool_crowded: ... and the ones computed there although it was ready.  Under the lone-wave schedule a WAVE copies the
 *   prepared episodes of its finished environments, up to four per step launch; the others of a wave with more -- in practice the one
 *   launch in which a whole batch started together runs into max_steps together -- compute theirs in place, all at once, which is
 *   the faster way through that launch (same bits either way). */
int hlx_set_episode_pool(hlx_env *env, int32_t interval);
int32_t hlx_get_episode_pool(const hlx_env *env);
int hlx_get_episode_pool_misses(hlx_env *env, int64_t *misses);
int hlx_get_episode_pool_crowded(hlx_env *env, int64_t *crowded);
/* The pool under a moving curriculum (environment.py:274-351; the reference's trainers call set_training_step_count after EVERY
 * step, train_flat_ppo.py:171-177).  A first observation reads three curriculum scalars.  The radar BEAM WIDTH -- the one the
 * shipped curriculum ramps (config.yaml:85-92) -- costs nothing: every prepared episode remembers the beam test it was computed
 * with and is used only if today's threshold decides it the same way (a new episode looks straight at its missile, so a ramp
 * invalidates next to nothing).  A sensor RELIABILITY that moves makes every prepared episode stale: while one is moving the pool
 * is suspended (next episodes are computed in place, as without a pool) and is filled again once the scalars have stood still
 * for 16 step launches.  out[4] = {auto-resets computed inside step launches with the pool in use, full fills, partial fills,
 * step launches issued while the pool was suspended} (synchronises; diagnostics). */
int hlx_get_episode_pool_stats(hlx_env *env, int64_t out[4]);

/* Load schedule of the step kernel (three instantiations of the same source, bit-identical results): 1 = the Kalman
 * groups and the delayed ground-ring sample are loaded as a second batch behind the Philox block (best while a SIMD
 * holds at most two waves), 2 = that, and the wave -- alone on its SIMD, with the register file to itself -- requests
 * the prepared next episode of a lane that finishes as soon as it knows (batches of at most one wave per SIMD: 65 536
 * environments on an MI355X), 0 = everything at entry (best once HBM-bound), -1 = choose from the batch size (the
 * default at hlx_create).  hlx_get_load_schedule returns the schedule in force. */
int hlx_set_load_schedule(hlx_env *env, int32_t mode);
int32_t hlx_get_load_schedule(const hlx_env *env);

/* Diagnostics: evaluate the step kernel's restated transcendentals element-wise on device arrays (current device).
 * kind 0: out[i] = powf(x[i], y) as glibc computes it (physics_models.py:100,324 are `np.float32 ** float` = libm powf);
 * kind 1: out[i] = np.exp(np.float32 x[i]) as numpy's float32 kernel computes it (physics_models.py:78,105,113,
 * environment.py:1174-1180,1222).  tests/ compare both bit for bit with the oracle's copies.
 * kind 2: out[i] = 1.0 where the kernel's short correctly rounded float32 square root of x[i] has the bits of IEEE sqrtf,
 * else 0.0; kind 3: the same for its short float64 square root on the argument |x[i] * y| + x[i] * x[i] (the norms
 * np.linalg.norm forms at environment.py:910-921, 1087-1099).  tests/ require 1.0 everywhere (kind 2: every float). */
int hlx_selftest_math(int32_t kind, const float *x, float y, float *out, int64_t n, void *stream);

int32_t hlx_num_envs(const hlx_env *env);
int64_t hlx_vec_steps(const hlx_env *env);              /* launches so far (the RNG/ring clock) */
/* Name of the shipped scenario preset whose constants the step launches of this handle carry as compile-time literals
 * ("medium/base", ...; hlynr_intercept_amd/build.py BAKED), or "" when the configuration is not one of them and the
 * constants are fetched at run time.  Same arithmetic either way. */
const char *hlx_kernel_baked(const hlx_env *env);
const char *hlx_kernel_variant(const hlx_env *env);     /* "base", "v2", "v2dr", "config", "config-easy", "config-volley", "generic" or "generic-volley" */
int32_t hlx_sizeof_config(void);
int32_t hlx_sizeof_env_state(void);
int32_t hlx_sizeof_info_soa(void);
/* ABI revision of this header (HLX_ABI_VERSION): bumped whenever a struct grows or an entry point changes meaning, so that a
 * binding can refuse a library built from another revision even where it cannot rebuild it (hlynr_intercept_amd/_lib.py). */
#define HLX_ABI_VERSION 4
int32_t hlx_abi_version(void);
/* 1 if this library is the SAFE build (-DHLX_HOT_FROM_MEMORY=1: the step kernel reads its constants from the parameter block in
 * memory instead of across lanes out of two vector registers -- what the build falls back to when the disassembly lint refuses the
 * product build; same results, slower), 0 for the product build. */
int32_t hlx_hot_words_from_memory(void);
const char *hlx_last_error(void);
const char *hlx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HLX_H */
