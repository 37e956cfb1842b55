"""Gymnasium-VectorEnv-style host shim over the C ABI (include/salp_vec.h).

`SalpVectorEnv` exposes the surface the reference's consumers use on `SalpSnakeEnv`
(src/salp/environments/salp_snake_env.py:17) for N environments at once:

    reset(seed=None, options=None) -> (obs, info)                       snake:133-155
    step(actions) -> (obs, reward, terminated, truncated, info)         snake:157-202
    num_envs, single_observation_space, single_action_space,
    observation_space, action_space, close()

Observations / rewards / flags are torch tensors resident on the GPU (`output="torch"`, the
default) or numpy arrays (`output="numpy"`, host-pointer ABI).  Autoreset is same-step, as in
gymnasium 0.29's VectorEnv: `info["final_observation"]` holds terminal observations of the
envs flagged in `info["_final_observation"]`.  Attribute pokes of the reference's eval scripts
(`env.robot_pos = ...`, eval/collect_navigation_data.py:76-89) map to `get_state()/set_state()`.

`SalpSB3VecEnv` adapts it to stable-baselines3's `VecEnv` duck type (`step_async/step_wait`,
`dones = terminated | truncated`, `infos[i]["terminal_observation"]`,
`infos[i]["TimeLimit.truncated"]`); `SalpSB3VecEnv.as_sb3_vecenv(...)` makes it a true `VecEnv` subclass where
stable-baselines3 is installed, which is what `SAC("MlpPolicy", env, ...)` (train.py:60-70) checks for
(parity unpinned: SB3 is not importable in the build environment).

PyTorch is used for device buffers and streams only; every simulation call goes through the
C ABI.  If the HIP library or a GPU is missing the constructor raises — there is no CPU path.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Sequence, Union

import numpy as np

from . import _capi
from ._capi import SALP_DEVICE_PTRS, SalpLib
from .config import SalpSnakeConfig, load_env_config
from .spaces import batch_space, single_action_space, single_observation_space

ConfigLike = Union[SalpSnakeConfig, str, Dict[str, Any]]


def _as_config(cfg: ConfigLike, **overrides) -> SalpSnakeConfig:
    if isinstance(cfg, SalpSnakeConfig):
        return cfg
    if isinstance(cfg, str):
        return load_env_config(cfg, **overrides)
    return SalpSnakeConfig(**{**cfg, **overrides})


class SalpVectorEnv:
    metadata = {"render_modes": [], "autoreset_mode": "same-step"}

    def __init__(self, config: ConfigLike = "single_food", num_envs: int = 4096, device: Union[str, int] = "cuda:0",
                 seed: int = 0, env_index_base: int = 0, output: str = "torch", **overrides):
        self.cfg = _as_config(config, **overrides)
        self.num_envs = int(num_envs)
        self.output = output
        if output not in ("torch", "numpy"):
            raise ValueError("output must be 'torch' or 'numpy'")
        self._device_index = int(str(device).split(":")[1]) if isinstance(device, str) and ":" in str(device) else (
            int(device) if not isinstance(device, str) else 0)
        self.seed_value = int(seed)
        self.env_index_base = int(env_index_base)
        self._lib = SalpLib(self.cfg, self.num_envs, self._device_index, self.seed_value, self.env_index_base)
        self.obs_dim, self.act_dim = self._lib.obs_dim, self._lib.act_dim
        self.single_observation_space = single_observation_space(self.cfg)
        self.single_action_space = single_action_space(self.cfg)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self._torch = None
        if output == "torch":
            import torch
            if not torch.cuda.is_available():
                raise _capi.SalpError("output='torch' needs a ROCm GPU visible to PyTorch")
            self._torch = torch
            self.device = torch.device("cuda", self._device_index)
        else:
            self.device = None
        self._bufs: Dict[str, Any] = {}

    # ------------------------------------------------------------------ buffers
    def _buf(self, name, shape, dtype):
        b = self._bufs.get(name)
        if b is None or tuple(b.shape) != tuple(shape):
            if self._torch is not None:
                td = {np.float32: self._torch.float32, np.uint8: self._torch.uint8, np.int32: self._torch.int32,
                      np.float64: self._torch.float64}[dtype]
                b = self._torch.empty(shape, dtype=td, device=self.device)
            else:
                b = np.empty(shape, dtype=dtype)
            self._bufs[name] = b
        return b

    @property
    def _flags(self):
        return SALP_DEVICE_PTRS if self._torch is not None else 0

    @property
    def _stream(self):
        if self._torch is None:
            return 0
        return int(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _actions_in(self, actions, lead_shape):
        shape = tuple(lead_shape) + (self.act_dim,)
        if self._torch is not None:
            t = self._torch
            if not isinstance(actions, t.Tensor):
                actions = t.as_tensor(np.asarray(actions, dtype=np.float32))
            a = actions.to(device=self.device, dtype=t.float32).reshape(shape).contiguous()
            return a
        return np.ascontiguousarray(np.asarray(actions, dtype=np.float32).reshape(shape))

    # ------------------------------------------------------------------ Gymnasium surface
    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None, mask=None):
        """Resets every env (or those in `mask`).  `seed` re-keys the draw streams (the reference's
        reset(seed) only seeds gymnasium's unused np_random, snake:136): `salp_vec_reseed` — new key words, draw counters
        at 0, every env reset, in place.  The C handle, its device state and this env's output buffers are kept (round 2
        destroyed and re-created the handle: a hipGraph captured earlier then replayed into freed memory), so graphs from
        `capture_policy_steps` / `train_sac_graphed` stay valid across `reset(seed=...)`; a poked base_num_food_items
        survives.  A seed restarts EVERY env's stream, so it cannot be combined with a partial `mask`."""
        if seed is not None:      # any explicit seed restarts the draw streams: reset(seed=s) twice gives the same episodes
            if mask is not None and not bool(np.all(np.asarray(mask.cpu() if hasattr(mask, "cpu") else mask))):
                raise ValueError("reset(seed=..., mask=...) with a partial mask: a seed re-keys the draw streams of ALL envs; "
                                 "reseed with mask=None, or reset the subset without a seed")
            self.seed_value = int(seed)
            obs = self._buf("obs", (self.num_envs, self.obs_dim), np.float32)
            self._lib.reseed(self.seed_value, obs, self._flags, self._stream)
            return obs, {}
        obs = self._buf("obs", (self.num_envs, self.obs_dim), np.float32)
        m = None
        if mask is not None:
            if self._torch is not None:
                m = self._torch.as_tensor(mask).to(device=self.device, dtype=self._torch.uint8).contiguous()
            else:
                m = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
        self._lib.reset(m, obs, self._flags, self._stream)
        return obs, {}

    def _prepare_step(self):
        """Allocates the step outputs once and caches their raw pointers (the per-call cost of
        `step` is then one ctypes call: at H = 1 the host, not the kernel, is the bottleneck)."""
        import ctypes
        n = self.num_envs
        obs = self._buf("obs", (n, self.obs_dim), np.float32)
        rew = self._buf("reward", (n,), np.float32)
        term = self._buf("terminated", (n,), np.uint8)
        trunc = self._buf("truncated", (n,), np.uint8)
        info_i = self._buf("info", (n, _capi.INFO_COLS), np.int32)
        fin = self._buf("final_obs", (n, self.obs_dim), np.float32)
        ptr = (lambda t: ctypes.c_void_p(t.data_ptr())) if self._torch is not None else \
              (lambda a: a.ctypes.data_as(ctypes.c_void_p))
        if self._torch is not None:
            term_b, trunc_b = term.view(self._torch.bool), trunc.view(self._torch.bool)   # zero-copy 0/1 bytes
            done = self._torch.empty((n,), dtype=self._torch.bool, device=self.device)
        else:
            term_b, trunc_b = term.view(np.bool_), trunc.view(np.bool_)
            done = np.empty((n,), dtype=np.bool_)
        info = {"food_collected": info_i[:, 0], "steps_since_food": info_i[:, 1], "collision": info_i[:, 2],
                "final_observation": fin, "_final_observation": done}
        self._step_cache = dict(obs=obs, rew=rew, term=term_b, trunc=trunc_b, info=info, done=done,
                                p_obs=ptr(obs), p_rew=ptr(rew), p_term=ptr(term), p_trunc=ptr(trunc),
                                p_fin=ptr(fin), p_info=ptr(info_i), fn=self._lib.lib.salp_vec_step, h=self._lib._h,
                                flags=self._flags, vp=ctypes.c_void_p)

    def step(self, actions, want_final_observation: bool = True):
        """One step of every env.  The returned tensors are the env's own output buffers: they are
        overwritten by the next call (clone what must be kept).  `info["score"]` of the reference
        (snake:196) is `info["food_collected"] * food_reward`."""
        c = getattr(self, "_step_cache", None)
        if c is None:
            self._prepare_step()
            c = self._step_cache
        t = self._torch
        if t is not None and isinstance(actions, t.Tensor) and actions.is_cuda and actions.device == self.device \
                and actions.dtype == t.float32 and actions.is_contiguous() \
                and actions.numel() == self.num_envs * self.act_dim:
            a = actions
            p_act = c["vp"](a.data_ptr())
            stream = c["vp"](t.cuda.current_stream(self.device).cuda_stream)
        else:
            a = self._actions_in(actions, (self.num_envs,))
            p_act = self._lib._ptr(a)
            stream = c["vp"](self._stream)
        rc = c["fn"](c["h"], p_act, c["p_obs"], c["p_rew"], c["p_term"], c["p_trunc"],
                     c["p_fin"] if want_final_observation else None, c["p_info"], c["flags"], stream)
        if rc != 0:
            _capi.check(self._lib.lib, rc, "salp_vec_step")
        if want_final_observation:
            if t is not None:
                t.logical_or(c["term"], c["trunc"], out=c["done"])
            else:
                np.logical_or(c["term"], c["trunc"], out=c["done"])
        return c["obs"], c["rew"], c["term"], c["trunc"], c["info"]

    def capture_policy_steps(self, policy, n_steps: int = 1, record=None, want_final_observation: bool = True):
        """Captures `n_steps` x (`policy(obs) -> actions`, `step(actions)`) into one hipGraph and returns the
        `torch.cuda.CUDAGraph`; every `replay()` advances all envs by `n_steps`.  For small batches the
        acting loop is launch-bound (one step kernel runs in a few microseconds), and a replay costs one
        host call whatever `n_steps` is.  `record(k, obs, actions, reward, terminated, truncated, info)`,
        if given, is called inside the capture after step k (e.g. to copy the transition into
        preallocated replay storage with torch ops); `obs` is the observation the policy saw (a clone).
        The body runs once un-captured as a warm-up (so does `record`); the env state is restored after it.
        `salp_vec_step` with device pointers is only kernel launches on the caller's stream, so it is
        capturable; device-generated actions (`rollout(actions=None)`) are not, their step index lives
        on the host.  After a replay the env's step outputs (`step`'s return buffers) hold the last step."""
        t = self._torch
        if t is None:
            raise _capi.SalpError("capture_policy_steps needs output='torch'")
        if getattr(self, "_step_cache", None) is None:
            self._prepare_step()
        c = self._step_cache
        obs = c["obs"]

        def body():
            for k in range(int(n_steps)):
                seen = obs.clone() if record is not None else obs
                a = policy(seen).to(t.float32).reshape(self.num_envs, self.act_dim).contiguous()
                o, r, te, tr, info = self.step(a, want_final_observation=want_final_observation)
                if record is not None:
                    record(k, seen, a, r, te, tr, info)

        # warm-up on a side stream (allocator pools, lazy kernel loads), then restore the envs
        f64, i32 = self.get_state()
        obs0 = obs.clone()
        side = t.cuda.Stream(device=self.device)
        side.wait_stream(t.cuda.current_stream(self.device))
        with t.cuda.stream(side):
            body()
        t.cuda.current_stream(self.device).wait_stream(side)
        t.cuda.synchronize(self.device)
        self.set_state(f64, i32)
        obs.copy_(obs0)
        g = t.cuda.CUDAGraph()
        with t.cuda.graph(g):
            body()
        # capture does not execute: state and obs are still those of before the call
        return g

    def rollout(self, actions=None, horizon: Optional[int] = None, want_obs: bool = True,
                want_final_observation: bool = False, out: Optional[dict] = None) -> dict:
        """`horizon` steps in one kernel launch.  actions: [H, N, act_dim] or None (device-generated)."""
        n = self.num_envs
        if actions is not None:
            if hasattr(actions, "shape"):
                horizon = int(actions.shape[0])
            else:
                horizon = len(actions)
            a = self._actions_in(actions, (horizon, n))
            aout = None
        else:
            if horizon is None:
                raise ValueError("horizon is required when actions is None")
            a = None
            aout = self._buf("r_act", (horizon, n, self.act_dim), np.float32)
        H = int(horizon)
        out = out or {}
        obs = out.get("obs") if "obs" in out else (self._buf("r_obs", (H, n, self.obs_dim), np.float32) if want_obs else None)
        rew = out.get("reward") if "reward" in out else self._buf("r_reward", (H, n), np.float32)
        term = out.get("terminated") if "terminated" in out else self._buf("r_term", (H, n), np.uint8)
        trunc = out.get("truncated") if "truncated" in out else self._buf("r_trunc", (H, n), np.uint8)
        fin = self._buf("r_final", (H, n, self.obs_dim), np.float32) if want_final_observation else None
        self._lib.rollout(a, H, obs, rew, term, trunc, fin, aout, self._flags, self._stream)
        return dict(obs=obs, reward=rew, terminated=term, truncated=trunc, final_obs=fin,
                    actions=a if a is not None else aout)

    def observe(self):
        obs = self._buf("obs", (self.num_envs, self.obs_dim), np.float32)
        self._lib.observe(obs, self._flags, self._stream)
        return obs

    def close(self):
        self._lib.close()
        self._bufs.clear()
        self._step_cache = None

    # ------------------------------------------------------------------ state access
    def get_state(self):
        """(f64 [SALP_F_COUNT(F), N], i32 [SALP_I_COUNT, N]) host numpy arrays; rows in _capi.F_* / I_*."""
        F = self.cfg.num_food_items
        f64 = np.empty((_capi.F_FOOD0 + 2 * F, self.num_envs), np.float64)
        i32 = np.empty((_capi.I_COUNT, self.num_envs), np.int32)
        if self._torch is not None:
            self._torch.cuda.current_stream(self.device).synchronize()
        self._lib.get_state(f64, i32, 0, 0)
        return f64, i32

    def set_state(self, f64=None, i32=None):
        f = None if f64 is None else np.ascontiguousarray(f64, dtype=np.float64)
        i = None if i32 is None else np.ascontiguousarray(i32, dtype=np.int32)
        if self._torch is not None:
            self._torch.cuda.current_stream(self.device).synchronize()
        self._lib.set_state(f, i, 0, 0)

    # the attributes the reference's callers read/poke, as whole-batch arrays
    @property
    def robot_pos(self):
        f, _ = self.get_state()
        return np.stack([f[_capi.F_X], f[_capi.F_Y]], axis=1)

    @property
    def robot_velocity(self):
        f, _ = self.get_state()
        return np.stack([f[_capi.F_VX], f[_capi.F_VY]], axis=1)

    @property
    def robot_angle(self):
        return self.get_state()[0][_capi.F_THETA]

    @property
    def food_positions(self):
        f, _ = self.get_state()
        F = self.cfg.num_food_items
        return np.stack([f[_capi.F_FOOD0:_capi.F_FOOD0 + F].T, f[_capi.F_FOOD0 + F:_capi.F_FOOD0 + 2 * F].T], axis=2)

    @property
    def num_food_items(self):
        """Per env, the number of foods of its CURRENT episode (what continuous_trainer.py:380 reads from the single
        env: `base_num_food_items`, or the episode's draw with random_food_count).  Recovered from the state: live
        foods, plus the captured ones when foods do not respawn."""
        f64, i32 = self.get_state()
        F = self.cfg.num_food_items
        live = (~np.isnan(f64[_capi.F_FOOD0:_capi.F_FOOD0 + F])).sum(axis=0)
        return live if self.cfg.respawn_food else live + i32[_capi.I_FOOD_COLLECTED]

    @property
    def base_num_food_items(self) -> int:
        """The attribute the reference's curriculum writes (continuous_trainer.py:409-411; snake:36): foods
        placed at every later reset, 0..cfg.num_food_items (the slots the env was created with)."""
        return self._lib.base_num_food

    @base_num_food_items.setter
    def base_num_food_items(self, k: int):
        self._lib.set_base_num_food(k)

    def stats(self) -> dict:
        return self._lib.stats()

    def clear_stats(self):
        self._lib.clear_stats()

    @property
    def global_step(self) -> int:
        return self._lib.global_step


class _LazyInfos(Sequence):
    """SB3 wants `list[dict]` per step; at thousands of envs that list is the bottleneck, so the
    dicts are materialised on access."""

    def __init__(self, n, food, ssf, coll, dones, truncs, terminal_obs, food_reward):
        self._n, self._food, self._ssf, self._coll = n, food, ssf, coll
        self._dones, self._truncs, self._tobs, self._fr = dones, truncs, terminal_obs, food_reward

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        d = {"food_collected": int(self._food[i]), "steps_since_food": int(self._ssf[i]),
             "collision": bool(self._coll[i]), "score": float(self._food[i]) * self._fr,
             "TimeLimit.truncated": bool(self._truncs[i])}
        if self._dones[i]:
            d["terminal_observation"] = self._tobs[i]
        return d


class SalpSB3VecEnv:
    """The method surface of stable-baselines3's `VecEnv` over SalpVectorEnv (numpy in / numpy out): `reset`,
    `step_async` / `step_wait`, `dones = terminated | truncated`, `infos[i]["terminal_observation"]`,
    `"TimeLimit.truncated"`, `get_attr / set_attr / env_method / env_is_wrapped`.  It is a duck type, NOT a subclass:
    stable-baselines3 is not installable in the build environment, so whether `SAC("MlpPolicy", env)` accepts it as
    is — SB3's `_wrap_env` tests `isinstance(env, VecEnv)` — is parity unpinned; `as_sb3_vecenv()` returns an
    instance of a real `VecEnv` subclass when stable-baselines3 is importable."""

    @classmethod
    def as_sb3_vecenv(cls, *args, **kwargs):
        """With stable-baselines3 importable: an instance of a class deriving from BOTH this adapter and SB3's
        `VecEnv` (so `isinstance(env, VecEnv)` holds and SB3 does not wrap it in a DummyVecEnv); raises
        ImportError otherwise."""
        from stable_baselines3.common.vec_env import VecEnv      # ImportError where SB3 is absent
        sub = type("SalpSB3VecEnvSubclass", (cls, VecEnv), {})
        self = sub.__new__(sub)
        cls.__init__(self, *args, **kwargs)
        VecEnv.__init__(self, self.num_envs, self.observation_space, self.action_space)
        return self

    def __init__(self, config: ConfigLike = "single_food", num_envs: int = 8, device: Union[str, int] = 0,
                 seed: int = 0, **overrides):
        self.venv = SalpVectorEnv(config, num_envs, device=device, seed=seed, output="numpy", **overrides)
        self.num_envs = self.venv.num_envs
        self.observation_space = self.venv.single_observation_space
        self.action_space = self.venv.single_action_space
        self.render_mode = None
        self._actions = None

    def reset(self):
        obs, _ = self.venv.reset()
        return obs.copy()

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.float32)

    def step_wait(self):
        obs, rew, term, trunc, info = self.venv.step(self._actions)
        dones = term | trunc
        infos = _LazyInfos(self.num_envs, info["food_collected"].copy(), info["steps_since_food"].copy(),
                           info["collision"].copy(), dones, trunc & ~term, info["final_observation"].copy(),
                           float(self.venv.cfg.food_reward))
        return obs.copy(), rew.copy(), dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.venv.close()

    def seed(self, seed=None):
        if seed is not None:
            self.venv.reset(seed=seed)
        return [seed] * self.num_envs

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices

    def get_attr(self, attr_name, indices=None):
        idx = list(self._indices(indices))
        if hasattr(self.venv.cfg, attr_name):
            return [getattr(self.venv.cfg, attr_name)] * len(idx)
        v = getattr(self.venv, attr_name)
        if np.ndim(v) == 0:
            return [v] * len(idx)
        return [v[i] for i in idx]

    def set_attr(self, attr_name, value, indices=None):
        if attr_name == "base_num_food_items":      # the one parameter the reference's trainers poke
            self.venv.base_num_food_items = value
            return
        raise AttributeError(f"{attr_name}: parameters are fixed at construction; state goes through set_state()")

    def env_method(self, method_name, *args, indices=None, **kwargs):
        raise AttributeError(f"env_method({method_name}) is not supported by the batched simulator")

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(list(self._indices(indices)))

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode=None):
        return None
