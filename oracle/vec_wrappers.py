"""TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else): numpy restatement of the two
Stable-Baselines3 wrappers the reference's trainers put around the vector environment
(rl_system/scripts/train_flat_ppo.py:384-399, train_hrl_pretrain.py:367-387):

    VecFrameStack(venv, n_stack)   -> stable_baselines3/common/vec_env/vec_frame_stack.py +
                                      stacked_observations.py  (StackedObservations.reset / .update)
    VecNormalize(venv, ...)        -> stable_baselines3/common/vec_env/vec_normalize.py +
                                      common/running_mean_std.py (RunningMeanStd.update_from_moments)

PARITY UNPINNED: stable-baselines3 (>=2.0.0, rl_system/requirements.txt:5) is a third-party dependency that is
neither under /root/reference nor installed in this image, and the reference ships no fixture of wrapped
observations.  What follows restates the published SB3 2.x algorithm; the GPU pipeline (include/hlx_obs.h) is tested
against it.  One deliberate difference, stated in the tests: SB3's RunningMeanStd.update takes np.mean / np.var of
the float32 batch (float32 accumulation); this oracle, like the GPU, reduces the batch in float64 (`batch_f32=True`
reproduces SB3's float32 batch moments so that the size of that difference is measured, not assumed).
"""
from __future__ import annotations

import numpy as np


class RunningMeanStd:
    """SB3 common/running_mean_std.py: mean 0, var 1, count epsilon=1e-4; Chan et al. parallel merge."""

    def __init__(self, shape=(), epsilon: float = 1e-4):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = float(epsilon)

    def update(self, arr: np.ndarray, batch_f32: bool = False) -> None:
        if batch_f32:     # what SB3 computes: reductions of a float32 array stay float32
            batch_mean, batch_var = np.mean(arr, axis=0), np.var(arr, axis=0)
        else:
            a = np.asarray(arr, np.float64)
            batch_mean, batch_var = np.mean(a, axis=0), np.var(a, axis=0)
        self.update_from_moments(batch_mean, batch_var, arr.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count) -> None:
        delta = batch_mean - self.mean
        tot_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot_count
        m_a = self.var * self.count
        m_b = batch_var * batch_count
        m_2 = m_a + m_b + np.square(delta) * self.count * batch_count / (self.count + batch_count)
        new_var = m_2 / (self.count + batch_count)
        self.mean, self.var, self.count = new_mean, new_var, batch_count + self.count


class FrameStack:
    """VecFrameStack for 1-D observations (channels-last): newest frame in stacked[:, -D:]."""

    def __init__(self, n_envs: int, obs_dim: int, n_stack: int):
        self.D, self.S = obs_dim, n_stack
        self.stacked = np.zeros((n_envs, obs_dim * n_stack), np.float32)

    def reset(self, obs: np.ndarray) -> np.ndarray:              # StackedObservations.reset
        self.stacked[...] = 0
        self.stacked[:, -self.D:] = obs
        return self.stacked.copy()

    def step(self, obs: np.ndarray, dones: np.ndarray, terminal_obs: np.ndarray):
        """Returns (stacked, terminal_stacked): terminal_stacked[i] is meaningful where dones[i]."""
        self.stacked = np.roll(self.stacked, shift=-self.D, axis=-1)   # StackedObservations.update
        terminal = np.zeros_like(self.stacked)
        for i in np.nonzero(dones)[0]:
            terminal[i] = np.concatenate((self.stacked[i, :-self.D], terminal_obs[i]))
            self.stacked[i] = 0
        self.stacked[:, -self.D:] = obs
        return self.stacked.copy(), terminal


class Normalize:
    """VecNormalize over already-stacked observations."""

    def __init__(self, n_envs: int, feat: int, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0,
                 clip_reward=10.0, gamma=0.99, epsilon=1e-8, batch_f32=False):
        self.obs_rms, self.ret_rms = RunningMeanStd((feat,)), RunningMeanStd(())
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        self.returns = np.zeros(n_envs, np.float64)
        self.batch_f32 = batch_f32

    def normalize_obs(self, obs: np.ndarray) -> np.ndarray:
        if not self.norm_obs:
            return obs
        z = np.clip((obs - self.obs_rms.mean) / np.sqrt(self.obs_rms.var + self.epsilon), -self.clip_obs, self.clip_obs)
        return z.astype(np.float32)

    def normalize_reward(self, reward: np.ndarray) -> np.ndarray:
        if not self.norm_reward:
            return reward
        return np.clip(reward / np.sqrt(self.ret_rms.var + self.epsilon), -self.clip_reward, self.clip_reward).astype(np.float32)

    def reset(self, stacked: np.ndarray) -> np.ndarray:
        self.returns[...] = 0
        if self.training and self.norm_obs:
            self.obs_rms.update(stacked, self.batch_f32)
        return self.normalize_obs(stacked)

    def step(self, stacked, reward, dones, terminal_stacked):
        if self.training and self.norm_obs:
            self.obs_rms.update(stacked, self.batch_f32)
        obs_n = self.normalize_obs(stacked)
        if self.training:                                          # _update_reward, also when norm_reward is off
            self.returns = self.returns * self.gamma + reward
            self.ret_rms.update(self.returns, False)
        reward_n = self.normalize_reward(reward)
        term_n = self.normalize_obs(terminal_stacked)              # rows of finished envs are the meaningful ones
        self.returns[dones] = 0
        return obs_n, reward_n, term_n
