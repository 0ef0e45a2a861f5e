"""CPU checks of the numpy restatement of SB3's VecFrameStack / VecNormalize (oracle/vec_wrappers.py): the properties
the published algorithm guarantees, so that the GPU tests compare against something that is itself pinned down."""
import numpy as np

from oracle.vec_wrappers import FrameStack, Normalize, RunningMeanStd


def test_running_mean_std_merge_equals_whole_data_moments():
    rng = np.random.default_rng(0)
    data = rng.normal(3.0, 2.0, (5, 400, 7))
    rms = RunningMeanStd((7,), epsilon=1e-4)
    for chunk in data:
        rms.update(chunk)
    flat = data.reshape(-1, 7)
    # Chan's merge reproduces the moments of the concatenation (up to the 1e-4 pseudo-count of the prior N(0, 1))
    np.testing.assert_allclose(rms.mean, flat.mean(0), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rms.var, flat.var(0), rtol=1e-5)
    assert rms.count == 2000 + 1e-4


def test_frame_stack_matches_brute_force_history():
    rng = np.random.default_rng(1)
    n, D, S, T = 5, 3, 4, 25
    fs = FrameStack(n, D, S)
    hist = [[] for _ in range(n)]                      # per env: frames of the current episode
    obs = rng.normal(size=(n, D)).astype(np.float32)
    out = fs.reset(obs)
    for i in range(n):
        hist[i] = [obs[i]]
    for t in range(T):
        obs = rng.normal(size=(n, D)).astype(np.float32)
        term = rng.normal(size=(n, D)).astype(np.float32)
        dones = rng.random(n) < 0.2
        out, terminal = fs.step(obs, dones, term)
        for i in range(n):
            prev = hist[i]
            if dones[i]:
                want_t = np.zeros(S * D, np.float32)
                frames = (prev + [term[i]])[-S:]
                want_t[S * D - len(frames) * D:] = np.concatenate(frames)
                assert np.array_equal(terminal[i], want_t)
                hist[i] = [obs[i]]
            else:
                hist[i] = prev + [obs[i]]
            frames = hist[i][-S:]
            want = np.zeros(S * D, np.float32)
            want[S * D - len(frames) * D:] = np.concatenate(frames)
            assert np.array_equal(out[i], want)


def test_normalize_clips_and_zeroes_returns_on_done():
    n, F = 64, 8
    rng = np.random.default_rng(2)
    nz = Normalize(n, F, norm_reward=True, clip_obs=2.0, clip_reward=1.5)
    nz.reset(rng.normal(size=(n, F)).astype(np.float32))
    for t in range(5):
        dones = rng.random(n) < 0.3
        o, r, _ = nz.step(rng.normal(0, 5, (n, F)).astype(np.float32), rng.normal(0, 50, n).astype(np.float32), dones,
                          np.zeros((n, F), np.float32))
        assert o.dtype == np.float32 and np.abs(o).max() <= 2.0 and np.abs(r).max() <= 1.5
        assert np.all(nz.returns[dones] == 0)
    frozen = Normalize(n, F, training=False)
    frozen.step(np.ones((n, F), np.float32), np.ones(n, np.float32), np.zeros(n, bool), np.zeros((n, F), np.float32))
    assert frozen.obs_rms.count == 1e-4 and np.all(frozen.returns == 0)   # evaluation mode touches nothing
