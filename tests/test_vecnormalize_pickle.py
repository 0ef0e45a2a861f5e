"""`vec_normalize.pkl` compatibility (train_flat_ppo.py:528-531 writes it, inference.py:450-477 loads it
with SB3's `VecNormalize.load`).  stable-baselines3 is absent from the build image, so a stand-in package with SB3 2.x's
pickling behaviour of `VecNormalize` / `RunningMeanStd` (`__getstate__` drops venv, class_attributes, returns;
`load` = pickle.load + set_venv) is put on the path of a child interpreter.  No GPU: only the file formats."""
import os
import pickle
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from hlynr_intercept_amd.wrappers import read_vecnormalize_pickle, write_vecnormalize_pickle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent('''
    # stand-in for stable_baselines3 (only what VecNormalize pickles touch)
    import pickle
    import numpy as np

    class RunningMeanStd:
        def __init__(self, epsilon=1e-4, shape=()):
            self.mean, self.var, self.count = np.zeros(shape, np.float64), np.ones(shape, np.float64), epsilon

    class Box:
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    class VecNormalize:
        def __init__(self, venv, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0, clip_reward=10.0, gamma=0.99,
                     epsilon=1e-8, norm_obs_keys=None):
            self.venv, self.num_envs = venv, venv.num_envs
            self.observation_space, self.action_space = venv.observation_space, venv.action_space
            self.class_attributes, self.render_mode = {}, None
            # VecEnv.__init__ (SB3 2.x): per-env reset bookkeeping + metadata -- present in tests/golden/sb3/vec_normalize_final.pkl
            self.reset_infos, self._seeds = [{} for _ in range(self.num_envs)], [None for _ in range(self.num_envs)]
            self._options, self.metadata = [{} for _ in range(self.num_envs)], {"render_modes": []}
            self.norm_obs_keys = norm_obs_keys
            self.obs_rms, self.ret_rms = RunningMeanStd(shape=self.observation_space.shape), RunningMeanStd(shape=())
            self.clip_obs, self.clip_reward = clip_obs, clip_reward
            self.returns = np.zeros(self.num_envs)
            self.gamma, self.epsilon, self.training, self.norm_obs, self.norm_reward = gamma, epsilon, training, norm_obs, norm_reward
            self.old_obs, self.old_reward = np.array([]), np.array([])

        def __getstate__(self):
            state = self.__dict__.copy()
            del state["venv"]; del state["class_attributes"]; del state["returns"]
            return state

        def __setstate__(self, state):
            self.__dict__.update(state)
            assert "venv" not in state
            self.venv = None

        def set_venv(self, venv):
            if self.venv is not None:
                raise ValueError("Trying to set venv of already initialized VecNormalize wrapper.")
            self.venv, self.num_envs, self.class_attributes = venv, venv.num_envs, {}
            assert self.observation_space.shape == venv.observation_space.shape      # utils.check_shape_equal
            self.returns = np.zeros(self.num_envs)

        @staticmethod
        def load(load_path, venv):
            with open(load_path, "rb") as f:
                vec_normalize = pickle.load(f)
            vec_normalize.set_venv(venv)
            return vec_normalize

        def save(self, save_path):
            with open(save_path, "wb") as f:
                pickle.dump(self, f)
''')


def _stub_tree(tmp_path):
    """tmp/stable_baselines3/common/{vec_env/__init__.py, running_mean_std.py} re-exporting the stand-ins."""
    base = tmp_path / "stable_baselines3"
    (base / "common" / "vec_env").mkdir(parents=True)
    (base / "_stub.py").write_text(STUB)
    (base / "__init__.py").write_text("")
    (base / "common" / "__init__.py").write_text("")
    (base / "common" / "running_mean_std.py").write_text("from stable_baselines3._stub import RunningMeanStd\n")
    (base / "common" / "vec_env" / "__init__.py").write_text("from stable_baselines3._stub import VecNormalize, Box\n")
    return str(tmp_path)


def _child(code, stub_dir, timeout=120):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([stub_dir, ROOT]))
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_sb3_written_pickle_is_readable_without_sb3(tmp_path):
    stub = _stub_tree(tmp_path)
    pkl = str(tmp_path / "vec_normalize.pkl")
    _child(f"""
        import numpy as np
        from stable_baselines3.common.vec_env import VecNormalize, Box
        class Venv: num_envs = 16; observation_space = Box(-np.inf, np.inf, (104,)); action_space = Box(-1, 1, (6,))
        v = VecNormalize(Venv(), norm_reward=False, clip_obs=7.5, gamma=0.97)
        v.obs_rms.mean[:] = np.arange(104) * 0.01; v.obs_rms.var[:] = 1.0 + np.arange(104) * 0.1; v.obs_rms.count = 12345.0
        v.ret_rms.mean, v.ret_rms.var, v.ret_rms.count = np.float64(3.5), np.float64(42.0), 999.0
        v.training = False
        v.save({pkl!r})
    """, stub)
    assert "stable_baselines3" not in sys.modules
    with pytest.raises((ImportError, AttributeError, ModuleNotFoundError)):
        with open(pkl, "rb") as f:
            pickle.load(f)                                   # the plain unpickler needs SB3 ...
    d = read_vecnormalize_pickle(pkl)                         # ... the tolerant one does not
    assert np.array_equal(d["obs_mean"], np.arange(104) * 0.01) and np.array_equal(d["obs_var"], 1.0 + np.arange(104) * 0.1)
    assert (d["obs_count"], d["ret_mean"], d["ret_var"], d["ret_count"]) == (12345.0, 3.5, 42.0, 999.0)
    assert (d["clip_obs"], d["clip_reward"], d["gamma"], d["epsilon"]) == (7.5, 10.0, 0.97, 1e-8)
    assert (d["norm_obs"], d["norm_reward"], d["training"]) == (True, False, False)


def test_our_save_is_a_genuine_sb3_pickle_where_sb3_is_present(tmp_path):
    stub = _stub_tree(tmp_path)
    pkl = str(tmp_path / "ours.pkl")
    out = _child(f"""
        import numpy as np, pickle
        from stable_baselines3.common.vec_env import VecNormalize, Box
        from hlynr_intercept_amd.wrappers import write_vecnormalize_pickle, read_vecnormalize_pickle
        state = dict(format="hlynr-vecnormalize-v1", obs_mean=np.linspace(-1, 1, 104), obs_var=np.linspace(0.5, 2, 104), obs_count=65536.0 * 7,
                     ret_mean=-3.25, ret_var=17.0, ret_count=65536.0 * 7, clip_obs=10.0, clip_reward=10.0, gamma=0.99, epsilon=1e-8,
                     norm_obs=True, norm_reward=False, training=True, n_stack=4)
        kind = write_vecnormalize_pickle({pkl!r}, state, Box(-np.inf, np.inf, (104,)), Box(-1, 1, (6,)), 65536)
        class Venv: num_envs = 4; observation_space = Box(-np.inf, np.inf, (104,)); action_space = Box(-1, 1, (6,))
        v = VecNormalize.load({pkl!r}, Venv())               # what inference.py:450-477 does
        assert isinstance(v, VecNormalize) and v.num_envs == 4 and v.returns.shape == (4,) and v.venv is not None
        assert np.array_equal(v.obs_rms.mean, state["obs_mean"]) and np.array_equal(v.obs_rms.var, state["obs_var"])
        assert v.obs_rms.count == state["obs_count"] and float(v.ret_rms.var) == 17.0 and v.ret_rms.mean.shape == ()
        assert (v.clip_obs, v.gamma, v.epsilon, v.training, v.norm_obs, v.norm_reward) == (10.0, 0.99, 1e-8, True, True, False)
        back = read_vecnormalize_pickle({pkl!r})            # and our own loader reads it back
        assert np.array_equal(back["obs_mean"], state["obs_mean"]) and back["ret_mean"] == -3.25
        print(kind)
    """, stub)
    assert out.strip().endswith("sb3")
    d = read_vecnormalize_pickle(pkl)                         # the parent has no SB3: tolerant path
    assert d["obs_count"] == 65536.0 * 7 and d["training"] is True


def test_dict_format_round_trip_and_rejection_of_foreign_files(tmp_path):
    state = dict(format="hlynr-vecnormalize-v1", obs_mean=np.zeros(26), obs_var=np.ones(26), obs_count=1.0, ret_mean=0.0, ret_var=1.0,
                 ret_count=1.0, clip_obs=10.0, clip_reward=10.0, gamma=0.99, epsilon=1e-8, norm_obs=True, norm_reward=True,
                 training=True, n_stack=1)
    p = str(tmp_path / "d.pkl")
    assert write_vecnormalize_pickle(p, state, None, None, 8) == "dict"          # no SB3 in this interpreter
    assert read_vecnormalize_pickle(p)["gamma"] == 0.99
    q = str(tmp_path / "foreign.pkl")
    with open(q, "wb") as f:
        pickle.dump({"something": "else"}, f)
    with pytest.raises(ValueError):
        read_vecnormalize_pickle(q)


# ---------------------------------------------------------------------------------------------------------------------
# A file Stable-Baselines3 itself wrote (tests/golden/sb3/README.md): pins the on-disk half of SURVEY.md 8 f1.
GENUINE = os.path.join(ROOT, "tests", "golden", "sb3", "vec_normalize_final.pkl")
GENUINE_KEYS = {"num_envs", "observation_space", "action_space", "reset_infos", "_seeds", "_options", "render_mode", "metadata",
                "norm_obs", "norm_obs_keys", "obs_rms", "ret_rms", "clip_obs", "clip_reward", "gamma", "epsilon", "training",
                "norm_reward", "old_reward", "old_obs"}


def _attribute_bag(path):
    from hlynr_intercept_amd.wrappers import _TolerantUnpickler
    with open(path, "rb") as f:
        return _TolerantUnpickler(f).load()


def test_genuine_sb3_file_is_read_without_sb3_value_for_value():
    assert "stable_baselines3" not in sys.modules and "gymnasium" not in sys.modules
    with pytest.raises((ImportError, AttributeError, ModuleNotFoundError)):
        with open(GENUINE, "rb") as f:
            pickle.load(f)                                   # names stable_baselines3 / gymnasium classes
    d = read_vecnormalize_pickle(GENUINE)
    assert d["obs_mean"].shape == (17,) and d["obs_mean"].dtype == np.float64 and d["obs_var"].shape == (17,)
    # spot values read off the file (float64 reprs round-trip exactly)
    assert d["obs_mean"][0] == 0.5039820088684809 and d["obs_mean"][1] == -1.6107384443599881 and d["obs_mean"][16] == 0.0
    assert d["obs_var"][0] == 1.3113113488131205 and d["obs_var"][1] == 5.6994204242424145
    assert d["obs_count"] == 1001480.0001000001 and d["ret_count"] == 1001472.0001000001      # 8 envs: obs counts the reset batch too
    assert d["ret_mean"] == -14.405738809215823 and d["ret_var"] == 41.546753377302906
    assert (d["clip_obs"], d["clip_reward"], d["gamma"], d["epsilon"]) == (10.0, 10.0, 0.99, 1e-8)
    assert (d["norm_obs"], d["norm_reward"], d["training"]) == (True, True, True)
    # SB3's RunningMeanStd invariants hold in the file: var > 0, constant features sit at the epsilon-count floor
    assert np.all(d["obs_var"] > 0) and np.isclose(d["obs_var"][16], 1e-4 / d["obs_count"], rtol=1e-6)


def test_the_attribute_set_sb3_pickles_is_the_one_this_package_and_the_stand_in_write(tmp_path):
    bag = _attribute_bag(GENUINE)
    assert (type(bag).__module__, type(bag).__name__) == ("stable_baselines3.common.vec_env.vec_normalize", "VecNormalize")
    assert set(bag.__dict__) == GENUINE_KEYS                 # __getstate__ dropped venv / class_attributes / returns
    assert set(bag.obs_rms.__dict__) == {"mean", "var", "count"} and type(bag.obs_rms).__module__ == "stable_baselines3.common.running_mean_std"
    assert (type(bag.observation_space).__module__, type(bag.observation_space).__name__) == ("gymnasium.spaces.box", "Box")
    assert bag.num_envs == 8 and len(bag.reset_infos) == 8 and bag.metadata == {"render_modes": []}
    # the stand-in the other tests run against pickles exactly this attribute set ...
    stub = _stub_tree(tmp_path)
    a, b = str(tmp_path / "stub.pkl"), str(tmp_path / "ours.pkl")
    _child(f"""
        import numpy as np
        from stable_baselines3.common.vec_env import VecNormalize, Box
        from hlynr_intercept_amd.wrappers import write_vecnormalize_pickle, read_vecnormalize_pickle
        class Venv: num_envs = 8; observation_space = Box(-1, 1, (17,)); action_space = Box(-1, 1, (6,))
        VecNormalize(Venv()).save({a!r})
        # ... and so does this package's writer: genuine file -> our writer -> SB3's own load() -> the same numbers
        d = read_vecnormalize_pickle({GENUINE!r})
        assert write_vecnormalize_pickle({b!r}, d, Venv.observation_space, Venv.action_space, 8) == "sb3"
        v = VecNormalize.load({b!r}, Venv())
        assert np.array_equal(v.obs_rms.mean, d["obs_mean"]) and np.array_equal(v.obs_rms.var, d["obs_var"])
        assert v.obs_rms.count == d["obs_count"] and float(v.ret_rms.mean) == d["ret_mean"] and v.ret_rms.count == d["ret_count"]
    """, stub)
    for path in (a, b):
        assert set(_attribute_bag(path).__dict__) == GENUINE_KEYS, path
    back, d = read_vecnormalize_pickle(b), read_vecnormalize_pickle(GENUINE)
    for k in d:
        assert np.array_equal(back[k], d[k]), k


def test_genuine_file_round_trips_through_the_dict_format_too(tmp_path):
    d = read_vecnormalize_pickle(GENUINE)
    state = dict(d, format="hlynr-vecnormalize-v1", n_stack=1)
    p = str(tmp_path / "dict.pkl")
    assert write_vecnormalize_pickle(p, state, None, None, 8) == "dict"
    back = read_vecnormalize_pickle(p)
    for k in d:
        assert np.array_equal(back[k], d[k]), k
