"""BASELINE.json configs[3] and configs[4] as reproducible bench modes (`bench.py --config 4|5`); not the headline.

Measurement harnesses around the device pipeline, NOT re-implementations of the reference's trainers: the networks are
plain torch modules of the reference's shapes with random weights (the caller's side of the boundary), the losses are
stand-ins that produce gradients of the right size.  What is measured is what the hot path and its neighbours cost
when a policy sits in the loop, and how large the one real exchange step -- the gradient all-reduce -- is next to it.

config 4 (rl_system/scripts/train_flat_ppo.py:371-399 vec env + VecFrameStack(4) + VecNormalize, :431-448 PPO with
          CustomMLP 104 -> 512 -> 512 -> 256 + LayerNorm, :512 model.learn): hard scenario, 65 536 envs per rank,
          `--rollout-steps` policy-in-the-loop steps, then `--minibatches` forward/backward passes each followed by ONE
          all-reduce of the flat gradient (~1.8 MB fp32; RCCL with the nccl backend).
config 5 (rl_system/inference.py:392-396 --volley, rl_system/hrl/wrappers.py:72-175, hrl/specialist_policies.py:95-141):
          volley K = 3, config.yaml physics, VecFrameStack(4), HRLController (rules selector) and three recurrent
          specialists (LSTM 104 -> 256, actor and critic LSTMs: 4 KB of hidden state per environment, resident in HBM
          and gathered / scattered by active option), 65 536 envs per rank.
"""
import time

ENVS_PER_GPU = 65536


def _ev(torch):
    return torch.cuda.Event(enable_timing=True)


class Sections:
    """Accumulates time per named section: event pairs on the current stream, read after one synchronise (on a machine
    without a GPU -- the gloo rehearsal of the harness logic in tests/ -- host clocks)."""

    def __init__(self, torch):
        self.torch, self.pairs, self.gpu = torch, {}, torch.cuda.is_available()

    def __call__(self, name):
        sec = self

        class _Ctx:
            def __enter__(self):
                if sec.gpu:
                    self.a, self.b = _ev(sec.torch), _ev(sec.torch)
                    self.a.record()
                else:
                    self.a = time.perf_counter()

            def __exit__(self, *exc):
                if sec.gpu:
                    self.b.record()
                else:
                    self.b = time.perf_counter()
                sec.pairs.setdefault(name, []).append((self.a, self.b))

        return _Ctx()

    def totals_us(self):
        if not self.gpu:
            return {k: 1e6 * sum(b - a for a, b in v) for k, v in self.pairs.items()}
        self.torch.cuda.synchronize()
        return {k: 1e3 * sum(a.elapsed_time(b) for a, b in v) for k, v in self.pairs.items()}


def flat_policy(torch, obs_dim=104, act_dim=6):
    """train_flat_ppo.py:37-84 CustomMLP (orthogonal init, LayerNorm, ReLU) + SB3's action / value heads and log_std."""
    nn = torch.nn
    layers, d = [], obs_dim
    for h in (512, 512, 256):
        lin = nn.Linear(d, h)
        nn.init.orthogonal_(lin.weight, gain=2 ** 0.5)
        nn.init.constant_(lin.bias, 0.0)
        layers += [lin, nn.LayerNorm(h), nn.ReLU()]
        d = h

    class Policy(nn.Module):
        def __init__(self):
            super().__init__()
            self.body = nn.Sequential(*layers)
            self.pi, self.vf = nn.Linear(256, act_dim), nn.Linear(256, 1)
            self.log_std = nn.Parameter(torch.zeros(act_dim))

        def forward(self, x):
            h = self.body(x)
            return self.pi(h), self.vf(h).squeeze(-1)

    return Policy()


def all_reduce_flat_grads(params, dist, world):
    """One collective per minibatch: gradients flattened into a single bucket (~1.8 MB), summed, averaged, scattered back."""
    import torch
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    if dist is not None and world > 1:
        dist.all_reduce(flat)
        flat /= world
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
    return flat.numel() * flat.element_size()


def rollout_with_policy(step_fn, obs, policy, n_steps, sections, torch, keep=4):
    """`n_steps` of policy-in-the-loop stepping.  step_fn(actions) -> (obs, reward, terminated, truncated).
    Keeps the last `keep` (obs, action, reward) triples as the stand-in minibatch source.  Returns (obs, kept, done count)."""
    kept, n_done = [], None
    std = policy.log_std.detach().exp()
    with torch.no_grad():
        for t in range(n_steps):
            with sections("policy"):
                mean, _ = policy(obs)
                act = torch.clamp(mean + std * torch.randn_like(mean), -1.0, 1.0)
            with sections("env+pipeline"):
                obs_next, rew, term, trunc = step_fn(act)
            if t >= n_steps - keep:
                kept.append((obs.clone(), act, rew.clone()))
            done = (term | trunc).sum()
            n_done = done if n_done is None else n_done + done
            obs = obs_next
    return obs, kept, n_done


def config4(args, rank, local_rank, world, dist):
    import torch
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.shard import gather_over_ranks, max_over_ranks, shard_range
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    from hlynr_intercept_amd.wrappers import VecFrameStack, VecNormalize

    n = args.envs_per_gpu
    dev = torch.device("cuda", local_rank)
    offset, _ = shard_range(n * world, world, rank)
    base = HlynrVecEnv(resolved=resolve_config(scenario_config("hard", "config")), num_envs=n, device=local_rank, seed=1000,
                       env_id_offset=offset)
    env = VecNormalize(VecFrameStack(base, 4), norm_obs=True, norm_reward=False, clip_obs=10.0, gamma=0.997)
    torch.manual_seed(1234)                      # identical replicas on every rank (data-parallel policy)
    policy = flat_policy(torch).to(dev)
    params = [p for p in policy.parameters()]
    opt = torch.optim.Adam(params, lr=3e-4)
    sections = Sections(torch)

    def step_fn(a):
        o, r, te, tr, _ = env.step_torch(a)
        return o, r, te, tr

    obs = env.reset_torch()
    # desynchronise the episodes with random actions (the normalisation statistics warm up on the way)
    g = torch.Generator(device=dev).manual_seed(rank)
    for _ in range(max(0, args.desync) // 8):
        obs = step_fn(torch.rand((n, 6), generator=g, device=dev) * 2 - 1)[0]
    obs, _, _ = rollout_with_policy(step_fn, obs, policy, 8, Sections(torch), torch)       # warm
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    obs, kept, n_done = rollout_with_policy(step_fn, obs, policy, args.rollout_steps, sections, torch)
    torch.cuda.synchronize(dev)
    t_roll = time.perf_counter() - t0
    # minibatch updates: forward + backward on a stand-in loss, ONE flat all-reduce each, optimiser step
    bucket = 0

    def update(k, sec):
        o, a, r = kept[k % len(kept)]
        with sec("update fwd+bwd"):
            mean, value = policy(o)
            loss = ((mean - a) ** 2).mean() + 0.5 * ((value - r) ** 2).mean() + 1e-3 * policy.log_std.sum()
            opt.zero_grad(set_to_none=False)
            loss.backward()
        with sec("gradient all-reduce"):
            nbytes = all_reduce_flat_grads(params, dist, world)
        with sec("optimiser"):
            opt.step()
        return nbytes

    update(0, Sections(torch))                   # warm: library kernel selection, allocator, communicator set-up
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.minibatches):
        bucket = update(k, sections)
    torch.cuda.synchronize(dev)
    t_upd = time.perf_counter() - t0
    red = dev if args.backend == "nccl" else None
    per_rank_ms = [1e3 * x for x in gather_over_ranks(t_roll, dist, red)]
    t_roll, t_upd = max_over_ranks(t_roll, dist, red), max_over_ranks(t_upd, dist, red)
    tot = sections.totals_us()
    T, M = args.rollout_steps, max(1, args.minibatches)
    line = {
        "metric": "env-steps/sec whole-node, hard scenario, policy in the loop (BASELINE.json configs[3])",
        "value": n * world * T / t_roll, "unit": "env-steps/s", "n_gpus": world, "steps": T, "warmup": 8,
        "ms_per_step": 1e3 * t_roll / T, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"hard scenario, config.yaml physics, {n} envs/GPU, VecNormalize(VecFrameStack(4)) on device, "
                               f"CustomMLP 104-512-512-256 policy (torch, fp32, random weights) in the loop, n_steps {T}; "
                               f"{M} minibatch updates with one flat gradient all-reduce each",
                   "kernel_variant": base.kernel_variant, "backend": args.backend if world > 1 else None},
        "shares_us_per_step": {"policy forward + sampling": tot.get("policy", 0.0) / T, "env step + frame stack + normalise": tot.get("env+pipeline", 0.0) / T},
        "update_us_per_minibatch": {"forward + backward": tot.get("update fwd+bwd", 0.0) / M, "gradient all-reduce": tot.get("gradient all-reduce", 0.0) / M,
                                    "optimiser": tot.get("optimiser", 0.0) / M, "wall": 1e6 * t_upd / M},
        "ranks": {"dist_world_size": dist.get_world_size() if dist is not None else 1, "backend": args.backend if dist is not None else None,
                  "per_rank_rollout_ms": per_rank_ms},
        "gradient_bucket_bytes": bucket, "parameters": sum(p.numel() for p in params),
        "episodes_finished_in_rollout": int(n_done.item()) if n_done is not None else 0,
        "note": "measurement harness, not a PPO implementation: the loss is a stand-in that yields gradients of the policy's size",
    }
    env.close()
    return line


def config5(args, rank, local_rank, world, dist):
    import torch
    from hlynr_intercept_amd.hrl import HRLController, SEARCH, TERMINAL, TRACK
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.shard import gather_over_ranks, max_over_ranks, shard_range
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    from hlynr_intercept_amd.wrappers import VecFrameStack

    n = args.envs_per_gpu
    dev = torch.device("cuda", local_rank)
    offset, _ = shard_range(n * world, world, rank)
    base = HlynrVecEnv(scenario_config("medium", "config", {"volley_mode": True, "volley_size": 3}), num_envs=n, device=local_rank,
                       seed=1000, env_id_offset=offset)
    env = VecFrameStack(base, 4)
    # a MIXED option population (round-3 review: the rules selector had every environment of the measured run in one option, so
    # the grouping path was barely exercised): an external selector with seeded choices, decided every 25 steps per environment,
    # on top of the forced transitions the observations trigger
    sel_gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    ctl = HRLController(n, obs_dim=104, device=local_rank, decision_interval=25,
                        selector=lambda a: torch.randint(0, 3, (a.shape[0],), generator=sel_gen, device=a.device, dtype=torch.int32))
    torch.manual_seed(99)

    class LstmCell(torch.nn.Module):
        """One LSTM layer advanced by one time step, its state updated IN PLACE (the state tensors are views of the controller's
        resident banks: nothing is copied back).  Same arithmetic as torch.nn.LSTM on a length-1 sequence."""

        def __init__(self, n_in, hidden):
            super().__init__()
            self.ih, self.hh = torch.nn.Linear(n_in, 4 * hidden), torch.nn.Linear(hidden, 4 * hidden)

        def forward(self, x, h, c):
            i, f, g, o = (self.ih(x) + self.hh(h)).chunk(4, dim=1)
            c.mul_(torch.sigmoid(f)).addcmul_(torch.sigmoid(i), torch.tanh(g))
            torch.mul(torch.sigmoid(o), torch.tanh(c), out=h)
            return h

    class Specialist(torch.nn.Module):
        """RecurrentPPO-shaped specialist (train_hrl_pretrain.py:421-425): separate 256-unit actor and critic LSTMs."""

        def __init__(self):
            super().__init__()
            self.actor, self.critic = LstmCell(104, 256), LstmCell(104, 256)
            self.pi = torch.nn.Linear(256, 6)

        def forward(self, rows, state, starts):
            k = rows.shape[0]
            if state is None:            # very first call: the controller learns the state's shape from what comes back
                state = tuple(torch.zeros((1, k, 256), device=rows.device) for _ in range(4))
                fresh = state
            else:
                fresh = None
            keep = (~starts).to(rows.dtype).view(1, k, 1)           # episode_start: begin from zeros
            for s_ in state:
                s_.mul_(keep)
            ha, ca, hc, cc = (s_[0] for s_ in state)                # [k, 256] views of the resident banks
            out = self.actor(rows, ha, ca)
            self.critic(rows, hc, cc)
            return torch.tanh(self.pi(out)), fresh                  # state updated in place: nothing to write back

    spec = {k: Specialist().to(dev) for k in (SEARCH, TRACK, TERMINAL)}
    sections = Sections(torch)
    obs = env.reset_torch()
    term = trunc = None
    names = {"controller": "controller_us", "grouping": "grouping_us", "gather / scatter": "lstm_state_gather_scatter_us",
             "specialist forward": "specialist_forward_us"}

    def loop(T, sec):
        nonlocal obs, term, trunc
        ctl.section = lambda name: sec(names[name])
        with torch.no_grad():
            for _ in range(T):
                actions, option, info = ctl.select_actions_recurrent(obs, spec, term, trunc)
                with sec("env_step_frame_stack_us"):
                    obs, rew, term, trunc, _ = env.step_torch(actions)
        ctl.section = None

    loop(20, Sections(torch))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    T = min(args.steps, 200)
    t0 = time.perf_counter()
    loop(T, sections)
    torch.cuda.synchronize(dev)
    local = time.perf_counter() - t0
    per_rank_ms = [1e3 * x for x in gather_over_ranks(local, dist, dev if args.backend == "nccl" else None)]
    elapsed = max_over_ranks(local, dist, dev if args.backend == "nccl" else None)
    tot = sections.totals_us()
    state_bytes = sum(x.numel() * x.element_size() for x in ctl.lstm_state) if ctl.lstm_state else 0
    line = {
        "metric": "env-steps/sec whole-node, volley multi-threat + HRL controller (BASELINE.json configs[4])",
        "value": n * world * T / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": T, "warmup": 20,
        "ms_per_step": 1e3 * elapsed / T, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"medium scenario, volley K=3, config.yaml physics, {n} envs/GPU, VecFrameStack(4), HRLController (external "
                               f"selector with seeded choices, decision interval 25, forced transitions on), three recurrent specialists (actor + "
                               f"critic LSTM cell 104->256 as torch ops updating the resident state in place, random weights)",
                   "kernel_variant": base.kernel_variant},
        "shares_us_per_step": {k: v / T for k, v in tot.items()},
        "everything_but_the_specialists_forward_us": sum(v for k, v in tot.items() if k not in ("specialist_forward_us", "env_step_frame_stack_us")) / T,
        "rows_moved_last_step": ctl.rows_moved,
        "unattributed_us_per_step": 1e6 * elapsed / T - sum(tot.values()) / T,      # host waits (the one per step for the run lengths) and launch gaps: GPU idle between sections
        "ranks": {"dist_world_size": dist.get_world_size() if dist is not None else 1, "backend": args.backend if dist is not None else None,
                  "per_rank_ms": per_rank_ms},
        "lstm_state_resident_bytes": state_bytes, "lstm_state_bytes_per_env": state_bytes // max(1, n),
        "options_now": torch.bincount(ctl.option.to(torch.int64), minlength=3).tolist(),
    }
    env.close()
    ctl.close()
    return line
