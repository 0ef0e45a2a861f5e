"""hlynr_intercept_amd: the batched `InterceptEnvironment.step()/reset()` of RomanSlack/Hlynr_Intercept as one HIP kernel on an
MI355X, behind the reference's environment API (include/hlx.h is the C ABI).

The faces, imported on first use (nothing here imports torch or loads the library):
    HlynrVecEnv          vec_env      SB3 `VecEnv` + device-tensor API over N environments
    HlynrGymVectorEnv    gym_vector   `gymnasium.vector.VectorEnv`, tensors in and out
    ShardedHlynrVecEnv   sharded      one object over several GPUs
    InterceptEnvironment single_env   ONE environment with the reference's class name and `gym.Env` semantics
    VecFrameStack, VecNormalize  wrappers   the two SB3 wrappers of the reference's trainers, on the device
    HRLController        hrl          the hierarchical wrapper's per-environment logic, on the device
"""
_EXPORTS = {"HlynrVecEnv": "vec_env", "HlynrGymVectorEnv": "gym_vector", "ShardedHlynrVecEnv": "sharded",
            "InterceptEnvironment": "single_env", "VecFrameStack": "wrappers", "VecNormalize": "wrappers", "HRLController": "hrl"}
__all__ = sorted(_EXPORTS)


def __getattr__(name):
    if name in _EXPORTS:
        import importlib

        return getattr(importlib.import_module("." + _EXPORTS[name], __name__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
