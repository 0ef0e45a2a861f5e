"""The in-kernel random number generator is Philox4x32-10 keyed as DESIGN.md 5 says: key = the env-set seed, counter =
(global env id lo, hi, vec-step lo, (vec-step hi << 8) | stream).  CPU: the plain-Python restatement reproduces the
published known-answer vectors.  GPU: the uniforms `hlx_fill_noise` exports are exactly that function's words."""
import numpy as np
import pytest

from tests.philox_ref import philox4x32, u01

# Random123 kat_vectors, "philox4x32 10": counter, key -> output
KAT = [
    ((0x00000000,) * 4, (0x00000000,) * 2, (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


@pytest.mark.parametrize("ctr,key,out", KAT)
def test_reference_implementation_known_answers(ctr, key, out):
    assert philox4x32(ctr, key) == out


@pytest.mark.gpu
def test_device_uniforms_are_philox4x32_10_words():
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv

    seed, offset, n = 0x1234_5678_9ABC_DEF0, (1 << 33) + 12345, 130
    env = HlynrVecEnv(scenario_config("medium", "base"), num_envs=n, seed=seed, env_id_offset=offset)
    env.reset()
    env.step(np.zeros((n, 6), np.float32))
    key = (seed & 0xFFFFFFFF, seed >> 32)
    for for_reset in (False, True):
        sn, rn = (x.cpu().numpy() for x in env.fill_noise(for_reset=for_reset))
        t = int(env._lib.hlx_vec_steps(env._h)) + (0 if for_reset else 1)
        # explicit resets carry the 16-bit reset epoch in bits 40-55 of the counter word: the number of hlx_reset calls at the
        # CURRENT clock value -- zero here, a step has advanced the clock since the reset above (include/hlx.h)
        for i in (0, 1, 63, 64, 129):
            gid = offset + i

            def words(stream):
                return philox4x32((gid & 0xFFFFFFFF, gid >> 32, t & 0xFFFFFFFF, ((t >> 32) << 8) | stream), key)

            x = words(0)                                   # RS_STEP_U: onboard, ground, datalink, gust uniforms
            assert [sn[11, i], sn[12, i], sn[19, i], sn[6, i]] == [u01(w) for w in x], (for_reset, i)
            # RS_RESET_U0..2, the ten spawn uniforms: an explicit reset draws them from the clock word, an auto-reset from the
            # index of the episode that starts (here the first auto-reset of every environment), under a high word that no
            # clock value reaches (include/hlx.h, hlx_set_episode_pool)
            def spawn_words(stream):
                if for_reset:
                    return words(stream)
                return philox4x32((gid & 0xFFFFFFFF, gid >> 32, 1, (0xFFFFFF << 8) | stream), key)

            r0, r1, r2 = spawn_words(8), spawn_words(9), spawn_words(10)
            assert list(rn[0:10, i]) == [u01(w) for w in (*r0, *r1, *r2[:2])], (for_reset, i)
    # ... and a reset with no step in between: epoch 1 at the same clock value
    env.reset()
    sn, rn = (x.cpu().numpy() for x in env.fill_noise(for_reset=True))
    t = int(env._lib.hlx_vec_steps(env._h)) | (1 << 40)
    gid = offset
    r0 = philox4x32((gid & 0xFFFFFFFF, gid >> 32, t & 0xFFFFFFFF, ((t >> 32) << 8) | 8), key)
    assert list(rn[0:4, 0]) == [u01(w) for w in r0]
    env.close()
