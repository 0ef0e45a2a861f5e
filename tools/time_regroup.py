"""Cost of the HRL controller's option-major regroup (include/hlx_hrl.h) by itself: python tools/time_regroup.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hlynr_intercept_amd.hrl import HRLController

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctl = HRLController(n, obs_dim=104)
dev = ctl.device
ctl.lstm_state = tuple(torch.zeros((1, n, 256), device=dev) for _ in range(4))
g = torch.Generator(device=dev).manual_seed(0)
option = torch.randint(0, 3, (n,), generator=g, device=dev, dtype=torch.uint8)
ctl._regroup(option)
for frac in (0.0, 0.01, 0.03, 0.66):
    opts = []
    for _ in range(50):
        ch = torch.rand(n, generator=g, device=dev) < frac
        option = torch.where(ch, torch.randint(0, 3, (n,), generator=g, device=dev, dtype=torch.uint8), option).contiguous()
        opts.append(option)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    moved = 0
    for o in opts:
        ctl._regroup(o)          # launches + the host wait for the run lengths
        moved += ctl.rows_moved
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(opts)
    # the same without moving state rows (no banks): what the planning + the wait cost alone
    banks, ctl.lstm_state = ctl.lstm_state, None
    t0 = time.perf_counter()
    for o in opts:
        ctl._regroup(o)
    torch.cuda.synchronize()
    dt0 = (time.perf_counter() - t0) / len(opts)
    ctl.lstm_state = banks
    ctl._regroup(option)
    print(f"n={n}: {100 * frac:4.0f} % of the environments redraw their option per step: regroup + wait {1e6 * dt:7.1f} us per step (host clock), "
          f"{moved / len(opts):8.0f} rows of 4 x 1 KiB moved per step; plan + wait alone (nothing to move) {1e6 * dt0:6.1f} us", flush=True)
ctl.close()
