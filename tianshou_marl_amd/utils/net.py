"""Actor / critic networks of the path as ONE flat HBM parameter vector.

Mirrors the reference's network stack for discrete-action PPO:
  Net(state_shape, hidden_sizes=[H, H]) -> DiscreteActor(softmax_output=False) / DiscreteCritic
  (/root/reference/tianshou/utils/net/common.py:90-181,246-369; discrete.py:27-124) joined by
  ActorCritic (common.py:461-474) so that one optimizer sees actor + critic parameters.
The fused kernels (csrc/mlp_fused.hip) read the parameters as a single flat f32 vector in
ActorCritic.parameters() order; per-layer tensors are views into it, so `state_dict()` can be
exchanged with the reference (`to_reference_state_dict`).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from .. import ops


def _layer_shapes(obs_dim: int, hidden: int, n_out: int):
    return [("w0", (hidden, obs_dim)), ("b0", (hidden,)), ("w1", (hidden, hidden)), ("b1", (hidden,)),
            ("w2", (n_out, hidden)), ("b2", (n_out,))]


class DiscreteActorCritic(nn.Module):
    """Separate actor and critic MLP trunks, obs[D] -> H -> H -> (A logits | 1 value), flat parameters."""

    def __init__(self, obs_dim: int, n_act: int, hidden: int = 64, device: str | torch.device = "cuda",
                 init: str = "orthogonal", seed: int | None = None) -> None:
        super().__init__()
        self.obs_dim, self.n_act, self.hidden = int(obs_dim), int(n_act), int(hidden)
        n = ops.policy_param_count(self.obs_dim, self.hidden, self.n_act)
        self.flat = nn.Parameter(torch.zeros(n, dtype=torch.float32, device=device), requires_grad=False)
        self._slices: OrderedDict[str, tuple[int, tuple[int, ...]]] = OrderedDict()
        o = 0
        for net, n_out in (("actor", self.n_act), ("critic", 1)):
            for name, shape in _layer_shapes(self.obs_dim, self.hidden, n_out):
                self._slices[f"{net}.{name}"] = (o, shape)
                o += int(np.prod(shape))
        assert o == n
        # padded LDS-layout copy of the parameters, refreshed by the Adam kernel (csrc/adam.hip) and by
        # sync_image(); the fused kernels stage it with straight 16-B copies
        self.image, self.image_map = ops.policy_image(self.obs_dim, self.hidden, self.n_act, device)
        self.reset_parameters(init, seed)

    def sync_image(self) -> None:
        """Re-derive the padded image from `flat` (call after modifying `flat` outside the optimizer)."""
        ops.scatter_image(self.flat.data, self.image, self.image_map)

    def view(self, name: str) -> torch.Tensor:
        o, shape = self._slices[name]
        return self.flat.data[o:o + int(np.prod(shape))].view(shape)

    def named_views(self):
        return [(k, self.view(k)) for k in self._slices]

    @torch.no_grad()
    def reset_parameters(self, init: str = "orthogonal", seed: int | None = None) -> None:
        """orthogonal weights + zero bias (the reference scripts' init, test/discrete/test_ppo_discrete.py:103-106)
        or torch's nn.Linear default (kaiming-uniform)."""
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        for name, v in self.named_views():
            if name.split(".")[1].startswith("w"):
                w = torch.empty(v.shape)
                if init == "orthogonal":
                    if gen is not None:
                        torch.manual_seed(int(torch.randint(0, 2**31 - 1, (1,), generator=gen)))
                    nn.init.orthogonal_(w)
                else:
                    bound = 1.0 / math.sqrt(v.shape[1])
                    w.uniform_(-bound, bound, generator=gen)
                v.copy_(w)
            elif init == "orthogonal":
                v.zero_()
            else:
                fan_in = self.view(name.replace(".b", ".w")).shape[1]
                b = torch.empty(v.shape).uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in), generator=gen)
                v.copy_(b)
        self.sync_image()

    # ---- reference checkpoint compatibility --------------------------------------
    _REF_KEYS = {
        "w0": "preprocess.model.model.0.weight", "b0": "preprocess.model.model.0.bias",
        "w1": "preprocess.model.model.2.weight", "b1": "preprocess.model.model.2.bias",
        "w2": "last.model.0.weight", "b2": "last.model.0.bias",
    }

    def reference_named_views(self) -> list[tuple[str, torch.Tensor]]:
        """[(key, view)] in ActorCritic.parameters() order with the key names `Algorithm.state_dict()` gives the
        reference's PPO over DiscreteActor/DiscreteCritic(Net(hidden_sizes=[H, H])): `policy.actor.<...>`, `critic.<...>`
        (algorithm_base.py:521-541; verified by tests/golden/checkpoint.npz)."""
        keys = getattr(self, "_ref_keys", None)
        out = []
        for i, (name, v) in enumerate(self.named_views()):
            net, layer = name.split(".")
            if keys is not None:
                k = keys[net][int(layer[1])][0 if layer[0] == "w" else 1]
            else:
                k = self._REF_KEYS[layer]
            out.append((("policy.actor." if net == "actor" else "critic.") + k, v))
        return out

    def to_reference_state_dict(self) -> dict[str, OrderedDict]:
        """{'actor': ..., 'critic': ...} with the key names of DiscreteActor/DiscreteCritic over Net."""
        out = {"actor": OrderedDict(), "critic": OrderedDict()}
        for name, v in self.named_views():
            net, layer = name.split(".")
            out[net][self._REF_KEYS[layer]] = v.detach().clone().cpu()
        return out

    @torch.no_grad()
    def load_reference_state_dict(self, sd: dict) -> None:
        for name, v in self.named_views():
            net, layer = name.split(".")
            v.copy_(torch.as_tensor(sd[net][self._REF_KEYS[layer]]).to(v.device, v.dtype).reshape(v.shape))
        self.sync_image()

    @torch.no_grad()
    def load_layers(self, actor, critic) -> None:
        """actor / critic: [(W, b)] * 3 (numpy or tensors) in torch nn.Linear layout."""
        for net, layers in (("actor", actor), ("critic", critic)):
            for i, (W, b) in enumerate(layers):
                self.view(f"{net}.w{i}").copy_(torch.as_tensor(np.asarray(W)).to(self.flat.device, torch.float32))
                self.view(f"{net}.b{i}").copy_(torch.as_tensor(np.asarray(b)).to(self.flat.device, torch.float32))
        self.sync_image()


class FlatMLP(nn.Module):
    """Fully-connected net dims[0] -> ... -> dims[-1] (hidden activation relu | tanh, linear output) whose
    parameters are ONE flat HBM vector in torch `parameters()` order; forward/backward run in csrc/dense.hip.

    `forward(x)` keeps the activations of the last call so that `backward(d_out)` can produce the gradient slabs
    (the pair replaces autograd for this module).  Layer views (`weight(i)`, `bias(i)`) alias the flat vector."""

    def __init__(self, dims, act: str = "relu", device: str | torch.device = "cuda", seed: int | None = None,
                 storage: torch.Tensor | None = None) -> None:
        super().__init__()
        self.dims = [int(d) for d in dims]
        self.act = act
        self.desc = ops.mlp_desc(self.dims, act)
        n = ops.mlp_param_count(self.desc)
        if storage is None:
            storage = torch.zeros(n, dtype=torch.float32, device=device)
        elif storage.numel() != n or storage.dtype != torch.float32 or not storage.is_contiguous():
            raise ValueError(f"FlatMLP: storage must be a contiguous f32 vector of {n} elements")
        # `storage` may be a slice of a larger joint parameter vector (actor + critic under one optimizer)
        self.flat = nn.Parameter(storage, requires_grad=False)
        self._offsets = []
        o = 0
        for i in range(len(self.dims) - 1):
            k, n = self.dims[i], self.dims[i + 1]
            self._offsets.append((o, o + n * k))
            o += n * k + n
        self._saved = None
        self.reset_parameters(seed)

    @property
    def n_layers(self) -> int:
        return len(self.dims) - 1

    def weight(self, i: int) -> torch.Tensor:
        o, ob = self._offsets[i]
        return self.flat.data[o:ob].view(self.dims[i + 1], self.dims[i])

    def bias(self, i: int) -> torch.Tensor:
        _, ob = self._offsets[i]
        return self.flat.data[ob:ob + self.dims[i + 1]]

    @torch.no_grad()
    def reset_parameters(self, seed: int | None = None) -> None:
        """torch nn.Linear default init (kaiming-uniform weights, uniform bias)."""
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        for i in range(self.n_layers):
            bound = 1.0 / math.sqrt(self.dims[i])
            self.weight(i).copy_(torch.empty(self.weight(i).shape).uniform_(-bound, bound, generator=gen))
            self.bias(i).copy_(torch.empty(self.bias(i).shape).uniform_(-bound, bound, generator=gen))

    @torch.no_grad()
    def load_layers(self, layers) -> None:
        """layers: [(W [out, in], b [out])] per layer (numpy or tensors), torch nn.Linear layout."""
        for i, (W, b) in enumerate(layers):
            self.weight(i).copy_(torch.as_tensor(np.asarray(W)).to(self.flat.device, torch.float32))
            self.bias(i).copy_(torch.as_tensor(np.asarray(b)).to(self.flat.device, torch.float32))

    def layer_views(self, flat: torch.Tensor):
        """[(W, b)] views of any flat vector with this net's layout (e.g. a summed gradient)."""
        out = []
        for i, (o, ob) in enumerate(self._offsets):
            out.append((flat[o:ob].view(self.dims[i + 1], self.dims[i]), flat[ob:ob + self.dims[i + 1]]))
        return out

    def forward(self, x: torch.Tensor, save: bool = True) -> torch.Tensor:
        x = x.to(self.flat.device, torch.float32).contiguous()
        lead = x.shape[:-1]
        if x.shape[-1] != self.dims[0]:
            raise ValueError(f"FlatMLP: input width {x.shape[-1]} != {self.dims[0]}")
        x2 = x.reshape(-1, self.dims[0])
        out, acts = ops.mlp_forward(self.desc, self.flat.data, x2)
        if save:
            self._saved = (x2, acts)
        return out.view(*lead, self.dims[-1])

    def backward(self, d_out: torch.Tensor, n_split: int = 0, slabs: torch.Tensor | None = None,
                 slab_stride: int = 0) -> torch.Tensor:
        """Gradient slabs [n_split, n_param] for the inputs of the last `forward(save=True)`; `slabs` / `slab_stride`
        direct them into the slabs of a joint parameter vector."""
        if self._saved is None:
            raise RuntimeError("FlatMLP.backward called before forward")
        x2, acts = self._saved
        return ops.mlp_backward(self.desc, self.flat.data, x2, acts, d_out.reshape(x2.shape[0], self.dims[-1]).contiguous(),
                                n_split, slabs=slabs, slab_stride=slab_stride)

    # reference module key names: fc1/fc2/fc3 (ctde.py:362-364, 398-400)
    def to_reference_state_dict(self) -> OrderedDict:
        sd = OrderedDict()
        for i in range(self.n_layers):
            sd[f"fc{i + 1}.weight"] = self.weight(i).detach().clone().cpu()
            sd[f"fc{i + 1}.bias"] = self.bias(i).detach().clone().cpu()
        return sd

    @torch.no_grad()
    def load_reference_state_dict(self, sd) -> None:
        self.load_layers([(sd[f"fc{i + 1}.weight"], sd[f"fc{i + 1}.bias"]) for i in range(self.n_layers)])


class MLPActorCritic(nn.Module):
    """Actor and critic MLPs of ARBITRARY widths under one flat parameter vector (`ActorCritic(actor, critic)`,
    common.py:461-474: one optimizer, one global gradient-norm clip).  Layout = actor parameters then critic parameters,
    each in torch `parameters()` order; `actor` / `critic` are `FlatMLP` views into it (csrc/dense.hip).  Use with
    `tianshou_marl_amd.algorithm.ppo_generic.GenericPPO` when the 64-wide fused kernels do not apply (hidden sizes
    other than 64, more layers, tanh, or a centralized critic: `critic_obs_dim = n_agent * obs_dim`)."""

    def __init__(self, obs_dim: int, n_act: int, hidden_sizes=(128, 128), act: str = "relu",
                 critic_obs_dim: int | None = None, device: str | torch.device = "cuda", seed: int | None = None,
                 init: str = "orthogonal") -> None:
        super().__init__()
        self.obs_dim, self.n_act, self.hidden = int(obs_dim), int(n_act), int(hidden_sizes[0])
        self.critic_obs_dim = int(critic_obs_dim or obs_dim)
        dims_a = [self.obs_dim, *hidden_sizes, self.n_act]
        dims_c = [self.critic_obs_dim, *hidden_sizes, 1]
        na = ops.mlp_param_count(ops.mlp_desc(dims_a, act))
        nc = ops.mlp_param_count(ops.mlp_desc(dims_c, act))
        self.flat = nn.Parameter(torch.zeros(na + nc, dtype=torch.float32, device=device), requires_grad=False)
        self.n_actor, self.n_critic = na, nc
        self.actor = FlatMLP(dims_a, act, device=device, storage=self.flat.data[:na])
        self.critic = FlatMLP(dims_c, act, device=device, storage=self.flat.data[na:])
        self.image = self.image_map = None  # no LDS image: the dense kernels stream the weights
        self.reset_parameters(init, seed)

    def sync_image(self) -> None:
        return None

    def reference_named_views(self) -> list[tuple[str, torch.Tensor]]:
        """As DiscreteActorCritic.reference_named_views, for Net(hidden_sizes=[...]) of any depth: hidden layer i is
        `preprocess.model.model.{2 i}` (Linear, activation, Linear, ...), the output layer `last.model.0`."""
        keys = getattr(self, "_ref_keys", None)
        out = []
        for net_name, net, prefix in (("actor", self.actor, "policy.actor."), ("critic", self.critic, "critic.")):
            for i in range(net.n_layers):
                if keys is not None:
                    kw, kb = keys[net_name][i]
                else:
                    stem = f"preprocess.model.model.{2 * i}" if i < net.n_layers - 1 else "last.model.0"
                    kw, kb = stem + ".weight", stem + ".bias"
                out += [(prefix + kw, net.weight(i)), (prefix + kb, net.bias(i))]
        return out

    @torch.no_grad()
    def reset_parameters(self, init: str = "orthogonal", seed: int | None = None) -> None:
        """orthogonal weights + zero bias (the reference scripts' init) or nn.Linear's default."""
        if seed is not None:
            torch.manual_seed(seed)
        for net in (self.actor, self.critic):
            if init != "orthogonal":
                net.reset_parameters(seed)
                continue
            for i in range(net.n_layers):
                w = torch.empty(net.weight(i).shape)
                nn.init.orthogonal_(w)
                net.weight(i).copy_(w)
                net.bias(i).zero_()


class FlatAdam:
    """torch.optim.Adam semantics (algorithm_base.py:485-498 wraps it) on one flat parameter vector, executed by
    `tsm_adam_step` (slab reduction + optional grad-norm clip + Adam in one launch).  Stands where
    `optim.Adam(module.parameters(), lr=...)` stands in the reference's CTDE constructors (ctde.py:36-37)."""

    def __init__(self, module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_grad_norm: float | None = None) -> None:
        self.param = module.flat.data if hasattr(module, "flat") else module
        self._module = module
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.exp_avg = torch.zeros_like(self.param)
        self.exp_avg_sq = torch.zeros_like(self.param)
        self.step_count = 0
        self._work = torch.zeros(ops.call("tsm_adam_work_elems", self.param.numel()), dtype=torch.float32,
                                 device=self.param.device)

    def zero_grad(self) -> None:  # gradients are produced fresh per step as slabs; nothing accumulates
        return None

    def step(self, grad_slabs: torch.Tensor) -> None:
        self.step_count += 1
        ops.adam_step(self.param, grad_slabs, self.exp_avg, self.exp_avg_sq, self.step_count, lr=self.lr,
                      betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, max_grad_norm=self.max_grad_norm,
                      work=self._work)

    def step_segs(self, segs: list, step_dev: torch.Tensor | None = None) -> None:
        """`step` for gradients that arrive as several slab arrays (ops.adam_step_segs: segments tiling the vector, each
        optionally scaled by a device scalar).  step_dev: device-resident step count (captured graphs)."""
        if step_dev is None:
            self.step_count += 1
        ops.adam_step_segs(self.param, segs, self.exp_avg, self.exp_avg_sq, self.step_count if step_dev is None else 1,
                           lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay,
                           max_grad_norm=self.max_grad_norm, work=self._work, step_dev=step_dev)

    def _shapes(self) -> list[tuple[int, ...]]:
        m = self._module
        if isinstance(m, FlatMLP):
            return [s for i in range(m.n_layers) for s in ((m.dims[i + 1], m.dims[i]), (m.dims[i + 1],))]
        return [tuple(self.param.shape)]

    def state_dict(self) -> dict:
        """torch.optim.Adam.state_dict() layout (per-parameter `state`, one `param_groups` entry), parameters in the
        module's `parameters()` order -- what `optim_actor.state_dict()` gives in the reference (ctde.py:36-37)."""
        state, o = {}, 0
        shapes = self._shapes()
        for i, shp in enumerate(shapes):
            n = int(np.prod(shp))
            if self.step_count > 0:
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[o:o + n].view(shp).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(shp).clone()}
            o += n
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False, "params": list(range(len(shapes)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        o, step = 0, 0
        for i, shp in enumerate(self._shapes()):
            n = int(np.prod(shp))
            e = sd["state"].get(i, sd["state"].get(str(i)))
            if e is not None:
                self.exp_avg[o:o + n].copy_(torch.as_tensor(e["exp_avg"]).reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(torch.as_tensor(e["exp_avg_sq"]).reshape(-1))
                step = int(float(e["step"]))
            o += n
        g = sd["param_groups"][0]
        self.step_count, self.lr = step, float(g["lr"])
        self.betas, self.eps, self.weight_decay = tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])


class RunningMeanStd:
    """tianshou.utils.statistics.RunningMeanStd (statistics.py:68-114), scalar statistics, host f64.
    The batch moments come from a device reduction; only three scalars live on the host."""

    def __init__(self, mean: float = 0.0, std: float = 1.0, clip_max: float | None = 10.0,
                 epsilon: float = float(np.finfo(np.float32).eps)) -> None:
        self.mean, self.var = mean, std  # NB: the reference stores `std` into var (statistics.py:92)
        self.clip_max, self.count, self.eps = clip_max, 0, epsilon

    def update_from_moments(self, batch_mean: float, batch_var: float, batch_count: int) -> None:
        delta = batch_mean - self.mean
        total = self.count + batch_count
        new_mean = self.mean + delta * batch_count / total
        m_2 = self.var * self.count + batch_var * batch_count + delta**2 * self.count * batch_count / total
        self.mean, self.var, self.count = new_mean, m_2 / total, total

    def update(self, x) -> None:
        if isinstance(x, torch.Tensor):
            xd = x.double()
            self.update_from_moments(float(xd.mean()), float(xd.var(unbiased=False)), x.numel())
        else:
            a = np.asarray(x, np.float64)
            self.update_from_moments(float(a.mean()), float(a.var()), a.size)


class DeviceRunningMeanStd(RunningMeanStd):
    """The same statistics with their state in HBM: `dev` = f64 {mean, var, count}.  The GAE kernel reads the scale
    sqrt(var + eps) from it and `tsm_rms_update` merges a batch of unnormalised returns into it (a2c.py:132-146), so
    `return_scaling` needs no host round trip and survives hipGraph replays.  `mean` / `var` / `count` read (and
    write) the device state; reading synchronises."""

    def __init__(self, device, mean: float = 0.0, std: float = 1.0, clip_max: float | None = 10.0,
                 epsilon: float = float(np.finfo(np.float32).eps)) -> None:
        self.dev = torch.tensor([mean, std, 0.0], dtype=torch.float64, device=device)
        self.clip_max, self.eps = clip_max, epsilon

    def _get(self, i: int) -> float:
        return float(self.dev[i].item())

    mean = property(lambda s: s._get(0), lambda s, v: s.dev[0:1].fill_(float(v)))
    var = property(lambda s: s._get(1), lambda s, v: s.dev[1:2].fill_(float(v)))
    count = property(lambda s: int(s._get(2)), lambda s, v: s.dev[2:3].fill_(float(v)))

    def update_scaled_returns(self, returns: torch.Tensor, rms_eps: float, ids: torch.Tensor | None = None) -> None:
        """update(returns * sqrt(var + rms_eps)) entirely on device (a2c.py:144-146)."""
        ops.rms_update(returns, self.dev, rms_eps, ids=ids)


def net_from_reference_modules(policy, critic, device="cuda"):
    """Build the flat HBM network from reference-shaped torch modules (`utils/ref_nets.py` mirrors or the reference's
    own `DiscreteActorPolicy` / `DiscreteCritic`): parameters are copied once, the key names of the modules'
    `state_dict()` are remembered for checkpoints.  64-64 ReLU nets on observations of at most 64 floats run on the
    fused kernels (DiscreteActorCritic); everything else on the general MLP kernels (MLPActorCritic).
    critic=None (Reinforce): the fused layout always carries a critic trunk; it stays untouched."""
    from .ref_nets import activation_of, linear_layers, reference_key_names

    cached = getattr(policy, "_tsm_net", None)
    if cached is not None and cached[0] is critic and str(cached[1].flat.device).startswith(str(device).split(":")[0]):
        return cached[1]
    actor_mod = getattr(policy, "actor", policy)
    la = linear_layers(actor_mod)
    act = activation_of(actor_mod)
    obs_dim, n_act = la[0][0].shape[1], la[-1][0].shape[0]
    hidden = [w.shape[0] for w, _ in la[:-1]]
    if critic is not None:
        lc = linear_layers(critic)
        if activation_of(critic) != act and len(lc) > 1:
            raise ValueError("actor and critic must use the same activation")
        if [w.shape[0] for w, _ in lc[:-1]] != hidden or lc[-1][0].shape[0] != 1:
            raise ValueError("actor and critic must have the same hidden sizes and the critic a single output")
        critic_obs = lc[0][0].shape[1]
    else:
        lc, critic_obs = None, obs_dim
    if len(la) == 3 and hidden == [64, 64] and obs_dim <= 64 and n_act <= 16 and act == "relu" and critic_obs == obs_dim:
        net = DiscreteActorCritic(obs_dim, n_act, 64, device=device)
        if lc is None:
            lc = [(np.zeros_like(net.view(f"critic.w{i}").cpu().numpy()), np.zeros_like(net.view(f"critic.b{i}").cpu().numpy()))
                  for i in range(3)]
        net.load_layers(la, lc)
    else:
        if lc is None:
            raise ValueError("Reinforce without a critic runs on the fused 64-64 layout only")
        net = MLPActorCritic(obs_dim, n_act, tuple(hidden), act=act, critic_obs_dim=critic_obs, device=device)
        net.actor.load_layers(la)
        net.critic.load_layers(lc)
    net._ref_keys = {"actor": reference_key_names(actor_mod),
                     "critic": reference_key_names(critic) if critic is not None else None}
    if net._ref_keys["critic"] is None:
        net._ref_keys = None
    try:
        policy._tsm_net = (critic, net)
    except Exception:  # noqa: BLE001  (objects without attribute assignment: no cache)
        pass
    return net
