/*
 * hlx_obs.h -- C ABI of the on-device observation pipeline that sits directly behind the step:
 * frame stacking + running-statistics normalisation, i.e. what the reference's trainers wrap around the
 * vector environment:
 *
 *     envs = VecFrameStack(envs, n_stack=frame_stack)              rl_system/scripts/train_flat_ppo.py:384-387
 *     envs = VecNormalize(envs, norm_obs=True, norm_reward=False,  rl_system/scripts/train_flat_ppo.py:392-399
 *                         clip_obs=10.0, clip_reward=10.0, gamma=...)
 *     (same pair at train_hrl_pretrain.py:367, :380; statistics saved/loaded as vec_normalize.pkl,
 *      train_flat_ppo.py:528-531, inference.py:450-477)
 *
 * Both wrappers are Stable-Baselines3 classes (stable-baselines3>=2.0.0, rl_system/requirements.txt:5; NOT under
 * /root/reference and not installed in the build image).  Their published semantics, restated in
 * oracle/vec_wrappers.py and implemented here:
 *   VecFrameStack (channels-last for 1-D observations): stacked[:, -D:] is the newest frame; on every step the
 *     stack shifts left by D; an environment that finished has its stack zeroed before its first new observation
 *     is inserted; infos[i]["terminal_observation"] becomes (the three previous frames, terminal observation).
 *   VecNormalize: obs_rms.update(stacked batch) while training (batch mean / population variance over the N
 *     environments merged with Chan's parallel formula, count starts at 1e-4, mean 0, var 1), observation ->
 *     clip((obs - mean) / sqrt(var + epsilon), +-clip_obs) as float32; returns = returns * gamma + reward,
 *     ret_rms.update(returns) while training (also when norm_reward is off), reward -> clip(reward /
 *     sqrt(ret_var + epsilon), +-clip_reward) when norm_reward is on, returns[done] = 0; the terminal
 *     observation is normalised with the statistics of the same step.
 *
 * Data flow: the pipeline owns a ring of n_stack raw frame planes [n_stack][N][D].  The caller passes
 * hlx_obs_next_slot() as the `obs` argument of hlx_step / hlx_reset, so the step kernel writes the new frame
 * straight into the ring (no copy), then calls hlx_obs_push / hlx_obs_push_reset, which (training) reduces the
 * batch moments, merges the running statistics and emits the stacked, normalised [N][n_stack*D] float32 batch the
 * policy consumes.  All pointers are DEVICE pointers unless stated; nothing synchronises; every function returns
 * an hlx_status (hlx.h) and sets hlx_last_error().
 */
#ifndef HLX_OBS_H
#define HLX_OBS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hlx_obs hlx_obs;

typedef struct hlx_obs_config {
    int32_t n_envs;
    int32_t obs_dim;       /* D: 26 for the intercept environment */
    int32_t n_stack;       /* VecFrameStack n_stack (1 = no stacking) */
    int32_t device;
    int32_t norm_obs;      /* VecNormalize norm_obs (0 = the pipeline is a plain VecFrameStack) */
    int32_t norm_reward;   /* VecNormalize norm_reward */
    int32_t training;      /* VecNormalize.training: update the running statistics */
    int32_t pad0;
    double clip_obs;       /* 10.0 at the reference's call sites */
    double clip_reward;    /* 10.0 */
    double gamma;          /* config['training']['gamma'], default 0.99 */
    double epsilon;        /* SB3 default 1e-8 */
} hlx_obs_config;

int hlx_obs_create(const hlx_obs_config *cfg, hlx_obs **out);
int hlx_obs_destroy(hlx_obs *p);

/* Where the NEXT raw observation batch [N][D] must be written (pass it as `obs` to hlx_step / hlx_reset). */
float *hlx_obs_next_slot(hlx_obs *p);

/* VecFrameStack.reset + VecNormalize.reset: the frame in next_slot is every environment's first observation;
 * stacks are zero-filled, returns zeroed, statistics updated (training), stacked_out [N][n_stack*D] emitted. */
int hlx_obs_push_reset(hlx_obs *p, float *stacked_out, void *stream);

/* VecFrameStack.step_wait + VecNormalize.step_wait for the frame in next_slot.
 * terminated/truncated [N] (either may be NULL), terminal_obs [N][D] raw (NULL = no terminal stacks wanted),
 * reward [N] raw (NULL = no return statistics this step); outputs: stacked_out [N][n_stack*D],
 * terminal_stacked_out [N][n_stack*D] (rows of finished environments only; may be NULL), reward_out [N]
 * (normalised reward, or a copy when norm_reward is off; may be NULL). */
int hlx_obs_push(hlx_obs *p, const uint8_t *terminated, const uint8_t *truncated, const float *terminal_obs,
                 const float *reward, float *stacked_out, float *terminal_stacked_out, float *reward_out,
                 void *stream);

/* hlx_step + hlx_obs_push in ONE call (what `VecNormalize(VecFrameStack(envs)).step_wait()` does on top of the reference's
 * env.step, train_flat_ppo.py:384-399): the step kernel writes the new frame into hlx_obs_next_slot() and the pipeline's
 * launches follow on the same stream, so that a Python caller crosses the FFI once per training step instead of three
 * times.  `env` is the hlx_env (hlx.h) whose observations the pipeline stacks; reward / terminated / truncated /
 * terminal_obs [N][D] are the step's raw outputs (required), done_idx / n_done / info as for hlx_step (optional);
 * stacked_out, terminal_stacked_out, reward_out as for hlx_obs_push. */
struct hlx_env;
struct hlx_info_soa;
int hlx_obs_step(hlx_obs *p, struct hlx_env *env, const float *actions, float *reward, uint8_t *terminated, uint8_t *truncated,
                 float *terminal_obs, int32_t *done_idx, int32_t *n_done, const struct hlx_info_soa *info, float *stacked_out,
                 float *terminal_stacked_out, float *reward_out, void *stream);

/* Re-emit the current stacks without advancing anything: normalise = 0 gives VecNormalize.get_original_obs(). */
int hlx_obs_emit(hlx_obs *p, int32_t normalise, float *stacked_out, void *stream);

int hlx_obs_set_mode(hlx_obs *p, int32_t training, int32_t norm_obs, int32_t norm_reward);

/* Running statistics, HOST arrays: obs_mean/obs_var [n_stack*D]; scalars obs_count, ret_mean, ret_var, ret_count
 * packed in scalars[4].  Synchronises.  (What VecNormalize.save / .load persist.) */
int hlx_obs_get_stats(hlx_obs *p, double *obs_mean, double *obs_var, double scalars[4]);
int hlx_obs_set_stats(hlx_obs *p, const double *obs_mean, const double *obs_var, const double scalars[4]);

int32_t hlx_obs_feature_dim(const hlx_obs *p);   /* n_stack * D */

#ifdef __cplusplus
}
#endif
#endif /* HLX_OBS_H */
