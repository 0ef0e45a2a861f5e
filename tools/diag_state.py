"""Diagnostic: GPU vs oracle side by side on a golden fixture; reports the first bitwise state difference."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.golden_util import load_fixture, inject_initial_state
from hlynr_intercept_amd.config import resolve_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv
import oracle.oracle as orc

FIELDS = ["int_pos", "int_vel", "int_quat", "mis_pos", "mis_vel", "wind", "thrust_actual", "kf_x"]

def snap_gpu(st):
    d = {f: np.array(getattr(st, f)[:], np.float64) for f in FIELDS}
    d["fuel"] = np.array([st.fuel], np.float64); d["kf_P"] = np.array(st.kf_P[:], np.float64)
    d["flags"] = np.array([st.steps, st.kf_init, st.kf_x_is64, st.crossed, st.worsening], np.float64)
    d["dist"] = np.array([st.prev_distance, st.min_distance, st.last_distance], np.float64)
    return d

def snap_orc(st):
    d = {f: np.array(getattr(st, f)[:], np.float64) for f in FIELDS}
    d["fuel"] = np.array([st.fuel], np.float64)
    P = np.array(st.kf_P[:], np.float64).reshape(6, 6)
    d["kf_P"] = np.array([P[0, 0], P[0, 3], P[3, 0], P[3, 3]])
    d["flags"] = np.array([st.steps, st.kf_init, st.kf_x_is64, st.crossed, st.worsening], np.float64)
    d["dist"] = np.array([st.prev_distance, st.min_distance, st.last_distance], np.float64)
    return d

for name in sys.argv[1:]:
    fx = load_fixture(name)
    rc = resolve_config(fx["config"])
    env = HlynrVecEnv(resolved=rc, num_envs=1)
    if fx["global_step_or_none"] is not None:
        env.set_training_step_count(fx["global_step_or_none"])
    cfg = orc.make_config(rc, fx["global_step_or_none"]); L = orc.lib()
    ost = orc.OrcState(); L.orc_init(C.byref(cfg), C.addressof(ost), 1)
    dev = env.device
    T = len(fx["action"])
    sn_all = torch.tensor(np.nan_to_num(fx["step_noise"], nan=0.5), dtype=torch.float64, device=dev)[:, :, None].contiguous()
    rn0 = torch.tensor(np.nan_to_num(fx["reset_noise0"], nan=0.5), dtype=torch.float64, device=dev)[:, None].contiguous()
    if "reset_noise" in fx:
        rn_all = torch.tensor(np.nan_to_num(fx["reset_noise"], nan=0.5), dtype=torch.float64, device=dev)[:, :, None].contiguous()
    actions = torch.tensor(fx["action"], dtype=torch.float32, device=dev)[:, None, :].contiguous()
    env.set_noise(sn_all[0], rn0); env.reset_torch()
    oobs = (C.c_float * 26)(); nz = np.ascontiguousarray(np.nan_to_num(fx["reset_noise0"], nan=0.5))
    L.orc_reset(C.byref(cfg), C.byref(ost), nz.ctypes.data_as(C.POINTER(C.c_double)), oobs)
    inject_initial_state(ost, fx)
    st = env.get_state()
    for fld in ("int_pos", "int_vel", "int_quat", "mis_pos", "mis_vel"):
        arr = getattr(st[0], fld)
        for k, x in enumerate(fx["init_" + fld]): arr[k] = float(x)
    st[0].fuel = float(fx["init_fuel"]); st[0].steps = int(fx["init_steps"])
    st[0].prev_distance = float(fx["init_prev_distance"]); st[0].min_distance = float(fx["init_min_distance"])
    st[0].last_distance = float(fx["init_last_distance"]); st[0].worsening = int(fx["init_worsening"]); st[0].crossed = int(fx["init_crossed"])
    env.set_state(st)
    out = orc.OrcOut(); k_reset = 0; reported = 0
    n_rew_bad = 0
    for t in range(T):
        rn = rn_all[k_reset] if fx["did_reset"][t] else rn0
        env.set_noise(sn_all[t], rn)
        obs, rew, term, trunc, info = env.step_torch(actions[t])
        a = np.ascontiguousarray(fx["action"][t], np.float32); z = np.ascontiguousarray(np.nan_to_num(fx["step_noise"][t], nan=0.5))
        L.orc_step(C.byref(cfg), C.byref(ost), a.ctypes.data_as(C.POINTER(C.c_float)), z.ctypes.data_as(C.POINTER(C.c_double)), C.byref(out))
        if fx["did_reset"][t]:
            nz = np.ascontiguousarray(np.nan_to_num(fx["reset_noise"][k_reset], nan=0.5))
            L.orc_reset(C.byref(cfg), C.byref(ost), nz.ctypes.data_as(C.POINTER(C.c_double)), oobs); k_reset += 1
        g, o = snap_gpu(env.get_state()[0]), snap_orc(ost)
        r_g, r_o = float(rew[0]), out.reward
        if abs(r_g - r_o) > 1e-5 * max(1, abs(r_o)): n_rew_bad += 1
        diffs = [(k, g[k], o[k]) for k in g if not np.array_equal(g[k], o[k])]
        if diffs and reported < 3:
            reported += 1
            print(f"{name} t={t}: " + "; ".join(f"{k}: gpu={gv} orc={ov} d={gv-ov}" for k, gv, ov in diffs[:4]))
            print(f"     reward gpu={r_g!r} orc={r_o!r} dist gpu={float(info['distance'][0])!r} orc={out.distance!r}")
    print(f"{name}: steps={T} first-diff-reports={reported} reward>1e-5 steps={n_rew_bad}")
    env.close()
