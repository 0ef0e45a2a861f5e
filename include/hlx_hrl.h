/*
 * hlx_hrl.h -- C ABI of the on-device HRL controller logic: the per-environment
 * integer/threshold part of the reference's hierarchical wrapper, for N environments at once:
 *
 *   abstract_observation(full_obs) -> 7-D selector state          rl_system/hrl/observation_abstraction.py:19-78
 *   extract_env_state_for_transitions(obs, env_info=None)         rl_system/hrl/observation_abstraction.py:81-130
 *       (called with the OBSERVATION-derived values, hrl/wrappers.py:104: the normalised obs[0:3] norm is compared
 *        with metre-valued thresholds -- reproduced as is)
 *   OptionManager.get_forced_transition / _is_critical_transition rl_system/hrl/option_manager.py:62-172
 *   thresholds / hysteresis bands / min-dwell steps               rl_system/hrl/option_definitions.py:47-86
 *   HierarchicalManager.select_action cadence, _switch_option, reset   rl_system/hrl/manager.py:84-221
 *   SelectorPolicy "rules" / "fixed" modes                        rl_system/hrl/selector_policy.py:132-200
 *
 * What stays outside (the policies themselves): the selector network ("model" mode) and the three specialists'
 * RecurrentPPO networks are the caller's; the controller tells it, per environment, which option is active and
 * whether the selector is due, and takes the selector's choices back.  hlynr_intercept_amd/hrl.py groups the
 * specialists' forward passes by active option and keeps recurrent specialists' LSTM states in a device-resident
 * bank (one live state per environment: manager.py:104-107, 210-215).
 *
 * All array arguments are DEVICE pointers; nothing synchronises; every function returns an hlx_status (hlx.h).
 */
#ifndef HLX_HRL_H
#define HLX_HRL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hlx_hrl hlx_hrl;

enum { HLX_OPT_SEARCH = 0, HLX_OPT_TRACK = 1, HLX_OPT_TERMINAL = 2 };            /* option_definitions.py:10-14 */
enum { HLX_SEL_FIXED = 0, HLX_SEL_RULES = 1, HLX_SEL_EXTERNAL = 2 };             /* selector_policy.py modes */
/* hlx_hrl_step info byte: bit0 option switched, bits1-2 switch reason (0 continue, 1 selector, 2 forced),
 * bit3 a forced transition fired, bit4 the selector was due this step, bits5-6 'hrl/selector_choice',
 * bit7 this is the first decision since the controller (re)started (hlx_hrl_reset, or a done flag of the previous step): with
 * bit0, the steps on which a recurrent specialist's state starts afresh (manager.py:104-107, 210-215) */

typedef struct hlx_hrl_config {
    int32_t n_envs;
    int32_t obs_dim;             /* 26, or 26*k for frame-stacked observations (the latest frame is the last 26) */
    int32_t device;
    int32_t decision_interval;   /* manager.py:39, default 100 */
    int32_t selector_mode;       /* HLX_SEL_* */
    int32_t enable_forced, enable_hysteresis, enable_min_dwell;   /* manager.py:40-42 */
    int32_t default_option;      /* manager.py:43, SEARCH */
    int32_t min_dwell[3];        /* option_definitions.py:82-86: 50, 50, 30 */
    double lock_min, lock_search, close_range, terminal_fuel_min, miss_imminent;   /* :47-53: 0.3 0.7 200 0.1 400 */
    double fuel_critical;                                                          /* option_manager.py:104: 0.10 */
    double h_lock_acquire, h_lock_maintain, h_terminal_enter, h_terminal_exit;     /* :73-78: 0.75 0.55 200 250 */
} hlx_hrl_config;

/* Fills thresholds, bands and dwell times with the reference's defaults (the integer switches are left alone). */
void hlx_hrl_default_thresholds(hlx_hrl_config *cfg);

int hlx_hrl_create(const hlx_hrl_config *cfg, hlx_hrl **out);
int hlx_hrl_destroy(hlx_hrl *h);

/* HierarchicalManager.reset() for the environments where mask != 0 (NULL = all). */
int hlx_hrl_reset(hlx_hrl *h, const uint8_t *mask, void *stream);

/* abstract_observation only (no state change): obs [N][obs_dim] -> abstract_out [N][7].  What an external selector
 * network is fed before hlx_hrl_step takes its choices. */
int hlx_hrl_abstract(hlx_hrl *h, const float *obs, float *abstract_out, void *stream);

/* One HierarchicalManager.select_action for every environment.
 * done_a / done_b [N]: terminated / truncated flags of the PREVIOUS environment step (either may be NULL): those
 *   environments' managers are reset first, as HRLActionWrapper.reset() does when the episode restarts.
 * selector_choice [N] int32: read where the selector is due and no forced transition fired (HLX_SEL_EXTERNAL only;
 *   clamped to 0..2 as manager.py:155 does); NULL in the other modes.
 * outputs (each may be NULL): abstract_out [N][7], option_out [N] active option after the step, info_out [N]. */
int hlx_hrl_step(hlx_hrl *h, const float *obs, const uint8_t *done_a, const uint8_t *done_b, const int32_t *selector_choice,
                 float *abstract_out, uint8_t *option_out, uint8_t *info_out, void *stream);

/* Controller state, HOST arrays [N][4] int32: option, HRLState.steps_in_option, OptionManager.steps_in_current_option,
 * total_steps.  Synchronises. */
int hlx_hrl_get_state(hlx_hrl *h, int32_t *host_out);
int hlx_hrl_set_state(hlx_hrl *h, const int32_t *host_in);

/* ---- Option-major residency of per-environment recurrent state (round 4) ---------------------------------------------------
 * The specialists are recurrent (RecurrentPPO: hrl/specialist_policies.py:95-183; separate 256-unit actor / critic LSTMs,
 * train_hrl_pretrain.py:421-425 -> 4 KiB of hidden state per environment) and each one runs on the environments whose ACTIVE
 * option is its own (manager.py:109-215).  Gathering those rows out of environment-major state banks and scattering them back is
 * 0.5 GB of traffic per step at 65 536 environments.  Instead the controller keeps a ROW ORDER in which every option's
 * environments are one contiguous run -- order[row] = environment, pos[environment] = row -- so that a specialist's state is a
 * slice of each bank, and hlx_hrl_regroup restores that property after the options have moved by relocating ONLY the rows that
 * have to move: an environment that switches option takes a row of its new run (its state restarts there anyway:
 * manager.py:210-215), and the row it leaves is filled from the end of its old run.  A few hundred 1-KiB rows per step instead of
 * 65 536 x 4 KiB twice.
 *
 * hlx_hrl_regroup(h, option, banks, row_bytes, n_banks, scratch, scratch_bytes, stream)
 *   option    [N] device: the active option of every environment after hlx_hrl_step (its option_out).
 *   banks     HOST array of n_banks (<= 8) DEVICE pointers, bank b a [N][row_bytes[b]] array in ROW order (row r belongs to
 *             environment order[r]); rows are moved inside them.  n_banks may be 0 (only order / pos are maintained).
 *   row_bytes HOST array: the row size of every bank, a positive multiple of 4 (multiples of 16 move in 16-byte units).
 *   scratch   device, 16-byte aligned, at least hlx_hrl_regroup_scratch_bytes(N, row_bytes, n_banks) bytes (staging for the rows
 *             that move; checked against scratch_bytes); may be NULL when n_banks == 0.
 *   Enqueues four small launches (count, classify, gather, scatter; the run lengths reach the host through pinned memory, written
 *   by the classify launch's last block); nothing synchronises.  Afterwards rows [0, c0) hold the
 *   environments whose option is 0, [c0, c0 + c1) option 1, the rest option 2.  WHICH row inside its run an environment gets is
 *   not defined (rows are claimed with atomics); everything a caller computes per environment is independent of it.
 * hlx_hrl_group_counts(h, out) waits for the latest regroup on the HOST (one event) and returns {c0, c1, c2, rows moved}.
 * hlx_hrl_order / hlx_hrl_pos: the two DEVICE int32[N] maps (valid after the latest regroup in stream order).
 * hlx_hrl_bind_order(h, order, pos): keep the two maps in CALLER-owned device arrays (e.g. torch tensors, which the caller indexes
 *   with) instead of the library's own; NULL, NULL returns to those.  The order restarts from the identity (synchronises).
 * hlx_hrl_reset leaves the order alone: the next regroup sees the reset environments' new options like any other switch. */
int hlx_hrl_regroup(hlx_hrl *h, const uint8_t *option, void *const *banks, const int64_t *row_bytes, int32_t n_banks, void *scratch,
                    int64_t scratch_bytes, void *stream);
int64_t hlx_hrl_regroup_scratch_bytes(int32_t n_envs, const int64_t *row_bytes, int32_t n_banks);
/* Around the specialists' forward passes, in the row order of the latest regroup (device pointers, nothing synchronises):
 * hlx_hrl_rows: obs_rows[r][:] = obs[order[r]][:] ([N][obs_dim]) and starts_rows[r] = 1 where environment order[r]'s recurrent
 *   state starts afresh on this step (info bits 0 and 7 of hlx_hrl_step), else 0 (uint8; may be NULL);
 * hlx_hrl_unrows: actions[order[r]][:] = act_rows[r][:] ([N][act_dim]). */
int hlx_hrl_rows(hlx_hrl *h, const float *obs, const uint8_t *info, float *obs_rows, uint8_t *starts_rows, void *stream);
int hlx_hrl_unrows(hlx_hrl *h, const float *act_rows, int32_t act_dim, float *actions, void *stream);
int hlx_hrl_group_counts(hlx_hrl *h, int32_t out[4]);
const int32_t *hlx_hrl_order(const hlx_hrl *h);
const int32_t *hlx_hrl_pos(const hlx_hrl *h);
int hlx_hrl_bind_order(hlx_hrl *h, int32_t *order, int32_t *pos);

#ifdef __cplusplus
}
#endif
#endif /* HLX_HRL_H */
