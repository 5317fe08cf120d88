"""The reference's network / policy constructors, so that a reference script changes its imports only.

    Net, MLP                 /root/reference/tianshou/utils/net/common.py:90-181, 246-369
    DiscreteActor / Critic   /root/reference/tianshou/utils/net/discrete.py:27-124
    ActorCritic              /root/reference/tianshou/utils/net/common.py:461-474
    DiscreteActorPolicy      /root/reference/tianshou/algorithm/modelfree/reinforce.py:195-243

These are small torch modules with the reference's constructor arguments, attribute names and `state_dict` key names
(`preprocess.model.model.{0,2}.*`, `last.model.0.*`).  They hold the INITIAL parameters (so the usual
`for m in actor_critic.modules(): orthogonal_(m.weight)` of the reference scripts works on them) and describe the
architecture; `PPO(policy=..., critic=..., optim=...)` copies the parameters into its flat HBM vector once and the HIP
kernels own them from then on (`algo.state_dict()` exports them under the reference's key names again).  Their torch
`forward` is a plain host evaluation kept for tests and debugging -- it is never on the training path.
"""
from __future__ import annotations

from collections.abc import Sequence

import numpy as np
import torch
from torch import nn

from ..env.spaces import Discrete


def _act_name(mod) -> str:
    if mod is None:
        return "none"
    cls = mod if isinstance(mod, type) else type(mod)
    if issubclass(cls, nn.ReLU):
        return "relu"
    if issubclass(cls, nn.Tanh):
        return "tanh"
    raise ValueError(f"unsupported activation {cls.__name__}: the HIP MLP kernels implement ReLU and Tanh")


class MLP(nn.Module):
    def __init__(self, *, input_dim: int, output_dim: int = 0, hidden_sizes: Sequence[int] = (), norm_layer=None,
                 norm_args=None, activation=nn.ReLU, act_args=None, linear_layer=nn.Linear, flatten_input: bool = True) -> None:
        super().__init__()
        if norm_layer:
            raise ValueError("norm layers are not supported by the HIP MLP kernels")
        if isinstance(activation, (list, tuple)):
            if len({_act_name(a) for a in activation}) > 1:
                raise ValueError("one activation for all hidden layers")
            activation = activation[0] if activation else None
        layers: list[nn.Module] = []
        d = int(input_dim)
        for h in hidden_sizes:
            layers.append(linear_layer(d, int(h)))
            if activation is not None:
                layers.append(activation() if isinstance(activation, type) else activation)
            d = int(h)
        if output_dim > 0:
            layers.append(linear_layer(d, int(output_dim)))
        self.output_dim = int(output_dim) or d
        self.model = nn.Sequential(*layers)
        self.flatten_input = flatten_input
        self.activation_name = _act_name(activation)

    def get_output_dim(self) -> int:
        return self.output_dim

    def forward(self, obs) -> torch.Tensor:
        x = torch.as_tensor(np.asarray(obs) if not isinstance(obs, torch.Tensor) else obs, dtype=torch.float32)
        return self.model(x.flatten(1) if self.flatten_input else x)


class Net(nn.Module):
    def __init__(self, *, state_shape, action_shape=0, hidden_sizes: Sequence[int] = (), norm_layer=None, norm_args=None,
                 activation=nn.ReLU, act_args=None, softmax: bool = False, concat: bool = False, num_atoms: int = 1,
                 dueling_param=None, linear_layer=nn.Linear) -> None:
        super().__init__()
        if softmax or concat or num_atoms != 1 or dueling_param is not None:
            raise ValueError("Net: softmax / concat / num_atoms / dueling heads are outside the on-policy hot path")
        input_dim = int(np.prod(state_shape))
        action_dim = int(np.prod(action_shape)) * num_atoms if np.prod(action_shape) else 0
        self.model = MLP(input_dim=input_dim, output_dim=action_dim, hidden_sizes=hidden_sizes, norm_layer=norm_layer,
                         norm_args=norm_args, activation=activation, act_args=act_args, linear_layer=linear_layer)
        self.output_dim = self.model.output_dim

    def get_output_dim(self) -> int:
        return self.output_dim

    def forward(self, obs, state=None, info=None):
        return self.model(obs), state


class DiscreteActor(nn.Module):
    def __init__(self, *, preprocess_net, action_shape, hidden_sizes: Sequence[int] = (), softmax_output: bool = True) -> None:
        super().__init__()
        self.output_dim = int(np.prod(action_shape))
        self.preprocess = preprocess_net
        self.last = MLP(input_dim=preprocess_net.get_output_dim(), output_dim=self.output_dim, hidden_sizes=hidden_sizes)
        self.softmax_output = softmax_output

    def get_preprocess_net(self):
        return self.preprocess

    def get_output_dim(self) -> int:
        return self.output_dim

    def forward(self, obs, state=None, info=None):
        x, h = self.preprocess(obs, state)
        x = self.last(x)
        return (torch.softmax(x, dim=-1) if self.softmax_output else x), h


class DiscreteCritic(nn.Module):
    def __init__(self, *, preprocess_net, hidden_sizes: Sequence[int] = (), last_size: int = 1) -> None:
        super().__init__()
        self.output_dim = int(last_size)
        self.preprocess = preprocess_net
        self.last = MLP(input_dim=preprocess_net.get_output_dim(), output_dim=last_size, hidden_sizes=hidden_sizes)

    def get_output_dim(self) -> int:
        return self.output_dim

    def forward(self, obs, state=None, info=None) -> torch.Tensor:
        x, _ = self.preprocess(obs, state)
        return self.last(x)


class ActorCritic(nn.Module):
    """common.py:461-474: lets one optimizer (and one init loop) see actor + critic parameters."""

    def __init__(self, actor: nn.Module, critic: nn.Module) -> None:
        super().__init__()
        self.actor, self.critic = actor, critic


def dist_fn_categorical_from_logits(logits: torch.Tensor) -> torch.distributions.Categorical:
    """reinforce.py (discrete default): Categorical(logits=...)."""
    return torch.distributions.Categorical(logits=logits)


class DiscreteActorPolicy(nn.Module):
    """reinforce.py:195-243 -- the description of a discrete stochastic policy: actor net + distribution + spaces."""

    def __init__(self, *, actor, dist_fn=dist_fn_categorical_from_logits, deterministic_eval: bool = False, action_space,
                 observation_space=None) -> None:
        super().__init__()
        n = getattr(action_space, "n", None)
        if n is None or type(action_space).__name__ != "Discrete":
            raise ValueError(f"Action space must be an instance of Discrete; got {action_space}")  # reinforce.py:233-234
        self.actor = actor
        self.dist_fn = dist_fn
        self.deterministic_eval = deterministic_eval
        self.action_space = action_space if isinstance(action_space, Discrete) else Discrete(int(n))
        self.observation_space = observation_space
        self.is_within_training_step = False


def linear_layers(module: nn.Module) -> list[tuple[np.ndarray, np.ndarray]]:
    """[(W [out, in], b [out])] of every nn.Linear of `module` in `modules()` order (works for these mirrors and for the
    reference's own torch modules alike)."""
    out = []
    for m in module.modules():
        if isinstance(m, nn.Linear):
            if m.bias is None:
                raise ValueError("linear layers without bias are not supported")
            out.append((m.weight.detach().cpu().numpy().astype(np.float32), m.bias.detach().cpu().numpy().astype(np.float32)))
    if not out:
        raise ValueError(f"{type(module).__name__} holds no nn.Linear layers")
    for (w0, _), (w1, _) in zip(out, out[1:]):
        if w1.shape[1] != w0.shape[0]:
            raise ValueError("the module is not a plain chain of linear layers")
    return out


def activation_of(module: nn.Module) -> str:
    names = {_act_name(m) for m in module.modules() if isinstance(m, (nn.ReLU, nn.Tanh))}
    other = [type(m).__name__ for m in module.modules()
             if not isinstance(m, (nn.ReLU, nn.Tanh, nn.Linear, nn.Sequential, nn.Identity, nn.Flatten)) and not list(m.children())]
    if other:
        raise ValueError(f"unsupported layers {sorted(set(other))}: the HIP MLP kernels run Linear + ReLU/Tanh chains")
    if len(names) > 1:
        raise ValueError("mixed activations are not supported")
    return names.pop() if names else "relu"


def reference_key_names(module: nn.Module) -> list[tuple[str, str]]:
    """[(weight key, bias key)] of the module's linear layers as `module.state_dict()` names them."""
    keys = []
    for name, m in module.named_modules():
        if isinstance(m, nn.Linear):
            keys.append((f"{name}.weight", f"{name}.bias"))
    return keys
