"""GPU parity tests for the fused actor+critic MLP kernels (f32 MFMA): rollout forward + sampling,
and the fused PPO gradient step.  Floating-point kernels: checked against a plain PyTorch
reference (f64 on CPU) of the same op, the CPU oracle, and the reference's own PPO fixture."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tianshou_marl_amd import ops

DEV = "cuda"


def pack(actor, critic):
    """[(W,b)]*3 per net -> flat f32 vector in ActorCritic.parameters() order."""
    parts = []
    for net in (actor, critic):
        for W, b in net:
            parts += [np.asarray(W, np.float32).reshape(-1), np.asarray(b, np.float32).reshape(-1)]
    return np.concatenate(parts)


def rand_nets(rng, D, H, A, scale=0.3):
    mk = lambda o, i: ((rng.standard_normal((o, i)) * scale).astype(np.float32),  # noqa: E731
                       (rng.standard_normal(o) * 0.1).astype(np.float32))
    actor = [mk(H, D), mk(H, H), mk(A, H)]
    critic = [mk(H, D), mk(H, H), mk(1, H)]
    return actor, critic


def torch_ppo_loss(params64, D, H, A, obs, act, logp_old, adv, ret, v_old, cfg):
    """Plain PyTorch (f64, CPU) restatement of ppo.py:182-211 on top of an MLP built from the flat vector."""
    p = params64
    o = 0

    def take(n, shape):
        nonlocal o
        t = p[o:o + n].reshape(shape)
        o += n
        return t

    aW1, ab1, aW2, ab2, aW3, ab3 = take(H * D, (H, D)), take(H, (H,)), take(H * H, (H, H)), take(H, (H,)), take(A * H, (A, H)), take(A, (A,))
    cW1, cb1, cW2, cb2, cW3, cb3 = take(H * D, (H, D)), take(H, (H,)), take(H * H, (H, H)), take(H, (H,)), take(H, (1, H)), take(1, (1,))
    x = torch.from_numpy(obs).double()
    h = torch.relu(torch.relu(x @ aW1.T + ab1) @ aW2.T + ab2)
    logits = h @ aW3.T + ab3
    hc = torch.relu(torch.relu(x @ cW1.T + cb1) @ cW2.T + cb2)
    value = (hc @ cW3.T + cb3).flatten()
    dist = torch.distributions.Categorical(logits=logits)
    a = torch.from_numpy(adv).double()
    if cfg["adv_norm"]:
        a = (a - a.mean()) / (a.std() + 1e-8)
    if cfg.get("loss_kind", 0) == 1:  # a2c.py:260-270 (Reinforce: vf_coef = ent_coef = 0, adv = returns)
        pg_loss = -(dist.log_prob(torch.from_numpy(act)) * a).mean()
        vf_loss = (torch.from_numpy(ret).double() - value).pow(2).mean()
        ent = dist.entropy().mean()
        return pg_loss + cfg["vf_coef"] * vf_loss - cfg["ent_coef"] * ent, pg_loss, vf_loss, ent, logits, value
    ratio = (dist.log_prob(torch.from_numpy(act)) - torch.from_numpy(logp_old).double()).exp()
    s1 = ratio * a
    s2 = ratio.clamp(1 - cfg["eps_clip"], 1 + cfg["eps_clip"]) * a
    if cfg["dual_clip"]:
        c1 = torch.min(s1, s2)
        c2 = torch.max(c1, cfg["dual_clip"] * a)
        clip_loss = -torch.where(a < 0, c2, c1).mean()
    else:
        clip_loss = -torch.min(s1, s2).mean()
    r = torch.from_numpy(ret).double()
    if cfg["value_clip"]:
        vs = torch.from_numpy(v_old).double()
        vclip = vs + (value - vs).clamp(-cfg["eps_clip"], cfg["eps_clip"])
        vf_loss = torch.max((r - value).pow(2), (r - vclip).pow(2)).mean()
    else:
        vf_loss = (r - value).pow(2).mean()
    ent = dist.entropy().mean()
    loss = clip_loss + cfg["vf_coef"] * vf_loss - cfg["ent_coef"] * ent
    return loss, clip_loss, vf_loss, ent, logits, value


@pytest.mark.parametrize("B,D,A", [(1, 18, 5), (16, 18, 5), (17, 18, 5), (3072, 18, 5), (1000, 48, 5), (333, 7, 3),
                                   (64, 64, 16), (100, 1, 2)])
def test_policy_forward_matches_oracle(oracle, B, D, A):
    rng = np.random.default_rng(B + D + A)
    H = 64
    actor, critic = rand_nets(rng, D, H, A)
    P = torch.from_numpy(pack(actor, critic)).to(DEV)
    assert P.numel() == ops.policy_param_count(D, H, A)
    obs = rng.standard_normal((B, D)).astype(np.float32)
    out = ops.policy_forward(P, torch.from_numpy(obs).to(DEV), A, H, mode="none")
    lg_o = oracle.mlp_forward(obs, [w for w, _ in actor], [b for _, b in actor])
    v_o = oracle.mlp_forward(obs, [w for w, _ in critic], [b for _, b in critic])[:, 0]
    # f32 FMA chains vs f64: 1e-5 relative to the activation scale
    np.testing.assert_allclose(out["logits"].cpu().numpy(), lg_o, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(out["value"].cpu().numpy(), v_o, rtol=1e-5, atol=2e-5)
    # given actions -> log-prob (ppo.py:157-161)
    act = rng.integers(0, A, B).astype(np.int32)
    out = ops.policy_forward(P, torch.from_numpy(obs).to(DEV), A, H, mode="given", act=torch.from_numpy(act).to(DEV))
    lp_o, _ = oracle.categorical_logp_entropy(lg_o.astype(np.float32), act)
    np.testing.assert_allclose(out["logp"].cpu().numpy(), lp_o, rtol=1e-4, atol=2e-5)
    # mode == argmax of the logits
    out = ops.policy_forward(P, torch.from_numpy(obs).to(DEV), A, H, mode="mode")
    lg = out["logits"].cpu().numpy()
    assert np.array_equal(out["act"].cpu().numpy(), lg.argmax(-1))


def test_policy_forward_sampling(oracle):
    rng = np.random.default_rng(0)
    D, H, A, B = 18, 64, 5, 100000
    actor, critic = rand_nets(rng, D, H, A)
    P = torch.from_numpy(pack(actor, critic)).to(DEV)
    obs = np.tile(rng.standard_normal((1, D)).astype(np.float32), (B, 1))
    o = torch.from_numpy(obs).to(DEV)
    a1 = ops.policy_forward(P, o, A, H, mode="sample", seed=7, offset=100)
    a2 = ops.policy_forward(P, o, A, H, mode="sample", seed=7, offset=100)
    a3 = ops.policy_forward(P, o, A, H, mode="sample", seed=8, offset=100)
    assert torch.equal(a1["act"], a2["act"]) and not torch.equal(a1["act"], a3["act"])
    # same stream definition as tsm_categorical_sample
    s_act, s_logp = ops.categorical_sample(a1["logits"], seed=7, offset=100)
    assert torch.equal(s_act, a1["act"])
    assert torch.allclose(s_logp, a1["logp"], rtol=1e-5, atol=1e-6)
    lg = a1["logits"][0].cpu().numpy().astype(np.float64)
    p = np.exp(lg - lg.max())
    p /= p.sum()
    counts = np.bincount(a1["act"].cpu().numpy(), minlength=A)
    chi2 = ((counts - B * p) ** 2 / (B * p)).sum()
    assert chi2 < 30.0, chi2


CFGS = {
    "default": dict(eps_clip=0.2, dual_clip=None, value_clip=False, adv_norm=True, vf_coef=0.5, ent_coef=0.01),
    "dual_vclip": dict(eps_clip=0.1, dual_clip=2.0, value_clip=True, adv_norm=True, vf_coef=0.5, ent_coef=0.01),
    "nonorm": dict(eps_clip=0.2, dual_clip=None, value_clip=False, adv_norm=False, vf_coef=0.25, ent_coef=0.02),
}


CFGS["a2c"] = dict(adv_norm=False, vf_coef=0.5, ent_coef=0.01, loss_kind=1)
CFGS["reinforce"] = dict(adv_norm=False, vf_coef=0.0, ent_coef=0.0, loss_kind=1)


@pytest.mark.parametrize("M,D,A,n_blocks", [(16, 18, 5, 1), (64, 18, 5, 4), (100, 18, 5, 3), (4096, 18, 5, None),
                                            (1000, 48, 5, 7), (257, 33, 9, 2)])
@pytest.mark.parametrize("variant", ["default", "dual_vclip", "nonorm", "a2c", "reinforce"])
def test_ppo_update_fused_gradients(M, D, A, n_blocks, variant):
    rng = np.random.default_rng(M + D)
    H = 64
    cfg = CFGS[variant]
    actor, critic = rand_nets(rng, D, H, A)
    Pn = pack(actor, critic)
    n = M + 50
    obs = rng.standard_normal((n, D)).astype(np.float32)
    act = rng.integers(0, A, n)
    logp_old = (rng.standard_normal(n) * 0.3 - 1.5).astype(np.float32)
    adv = (rng.standard_normal(n) * 2 + 0.3).astype(np.float32)
    ret = rng.standard_normal(n).astype(np.float32)
    v_old = rng.standard_normal(n).astype(np.float32)
    perm = rng.permutation(n)[:M]
    # reference: torch autograd in f64
    p64 = torch.from_numpy(Pn).double().requires_grad_(True)
    loss, clip_l, vf_l, ent, _, _ = torch_ppo_loss(p64, D, H, A, obs[perm], act[perm], logp_old[perm], adv[perm],
                                                   ret[perm], v_old[perm], cfg)
    loss.backward()
    g_ref = p64.grad.numpy()
    # device
    d = lambda x, dt=None: torch.from_numpy(np.ascontiguousarray(x)).to(DEV, dt)  # noqa: E731
    adv_d, perm_d = d(adv), d(perm)
    stats = ops.ppo_adv_stats(adv_d, d(np.array([0, M], np.int64)), perm=perm_d)
    slabs, sc = ops.ppo_update_fused(d(Pn), d(obs), d(act, torch.int32), d(logp_old), adv_d, d(ret),
                                     ops.make_ppo_cfg(**cfg), A, H, adv_stats=stats[0], v_s_old=d(v_old),
                                     perm=perm_d, n_blocks=n_blocks)
    g = slabs.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(sc.cpu().numpy(), [loss.item(), clip_l.item(), vf_l.item(), ent.item()],
                               rtol=2e-5, atol=2e-6)
    # gradients: f32 MFMA chains vs f64 autograd -- 1e-4 of the gradient scale
    scale = np.abs(g_ref).max()
    np.testing.assert_allclose(g, g_ref, rtol=1e-3, atol=1e-4 * scale)
    assert np.linalg.norm(g - g_ref) / np.linalg.norm(g_ref) < 1e-5


def test_ppo_update_fused_reference_fixture(golden_dir):
    """Full-batch gradient step of the reference PPO (tests/golden/ppo_update.npz, variant `default`):
    gradients after backward and parameters after Adam must match the reference's own tensors."""
    g = np.load(os.path.join(golden_dir, "ppo_update.npz"), allow_pickle=True)
    p = "default_"
    actor = [(g[p + f"actor_w{i}"], g[p + f"actor_b{i}"]) for i in range(3)]
    critic = [(g[p + f"critic_w{i}"], g[p + f"critic_b{i}"]) for i in range(3)]
    Pn = pack(actor, critic)
    D, H, A = 18, 64, 5
    n = len(g[p + "act"])
    d = lambda x, dt=None: torch.from_numpy(np.ascontiguousarray(x)).to(DEV, dt)  # noqa: E731
    P = d(Pn)
    obs = d(g[p + "obs"])
    # a13: logp_old and a11: v_s recomputed by the fused forward == the reference's tensors
    out = ops.policy_forward(P, obs, A, H, mode="given", act=d(g[p + "act"], torch.int32))
    np.testing.assert_allclose(out["logp"].cpu().numpy(), g[p + "logp_old"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["value"].cpu().numpy(), g[p + "v_s"], rtol=1e-4, atol=1e-5)
    stats = ops.ppo_adv_stats(d(g[p + "adv"]), d(np.array([0, n], np.int64)))
    slabs, sc = ops.ppo_update_fused(P, obs, d(g[p + "act"], torch.int32), d(g[p + "logp_old"]), d(g[p + "adv"]),
                                     d(g[p + "returns"]), ops.make_ppo_cfg(), A, H, adv_stats=stats[0])
    ref_sc = [g[p + "stat_loss"][0], g[p + "stat_actor_loss"][0], g[p + "stat_vf_loss"][0], g[p + "stat_ent_loss"][0]]
    np.testing.assert_allclose(sc.cpu().numpy(), ref_sc, rtol=1e-5, atol=1e-6)
    grads_ref = pack([(g[p + f"last_actor_gw{i}"], g[p + f"last_actor_gb{i}"]) for i in range(3)],
                     [(g[p + f"last_critic_gw{i}"], g[p + f"last_critic_gb{i}"]) for i in range(3)])
    grad = slabs.sum(0).cpu().numpy()
    assert np.linalg.norm(grad - grads_ref) / np.linalg.norm(grads_ref) < 1e-5
    # a15: Adam(lr=3e-4) step 1 -> the reference's post-update weights
    m, v = torch.zeros_like(P), torch.zeros_like(P)
    ops.adam_step(P, slabs, m, v, 1, lr=3e-4)
    after = pack([(g[p + f"after_actor_w{i}"], g[p + f"after_actor_b{i}"]) for i in range(3)],
                 [(g[p + f"after_critic_w{i}"], g[p + f"after_critic_b{i}"]) for i in range(3)])
    np.testing.assert_allclose(P.cpu().numpy(), after, rtol=1e-5, atol=5e-6)


def test_fused_mlp_rejects_unsupported_dims():
    P = torch.zeros(10, device=DEV)
    with pytest.raises(ValueError):
        ops.policy_forward(P, torch.zeros(4, 100, device=DEV), 5, 64)
    with pytest.raises(ValueError):
        ops.policy_param_count(18, 128, 5)


@pytest.mark.parametrize("M,D,A,kw", [(4096, 18, 5, {}), (1000, 48, 5, dict(dual_clip=2.0, value_clip=True)),
                                      (257, 33, 9, dict(adv_norm=False)), (300, 18, 5, dict(adv_norm=False, loss_kind=1))])
def test_one_net_per_workgroup_update_is_bit_identical_to_the_joint_kernel(M, D, A, kw):
    """`ppo_update_split_kernel` (grid.y = actor | critic, the default) runs the same per-net instruction sequence as
    `ppo_update_kernel` (both nets in one workgroup): slabs and loss statistics must agree bit for bit."""
    import ctypes

    from tianshou_marl_amd import _abi

    lib = _abi.load()
    lib.tsm_debug_set_update_variant.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(M)
    actor, critic = rand_nets(rng, D, 64, A)
    P = torch.from_numpy(pack(actor, critic)).to(DEV)
    n = M + 77
    d = lambda x, dt=None: torch.from_numpy(np.ascontiguousarray(x)).to(DEV, dt)  # noqa: E731
    obs, act = d(rng.standard_normal((n, D)).astype(np.float32)), d(rng.integers(0, A, n), torch.int32)
    lp, adv, ret, vo = (d(rng.standard_normal(n).astype(np.float32)) for _ in range(4))
    perm = d(rng.permutation(n)[:M])
    stats = ops.ppo_adv_stats(adv, d(np.array([0, M], np.int64)), perm=perm)
    cfg = ops.make_ppo_cfg(**kw)
    out = []
    try:
        for variant in (0, 1):
            lib.tsm_debug_set_update_variant(variant)
            slabs, sc = ops.ppo_update_fused(P, obs, act, lp, adv, ret, cfg, A, 64, adv_stats=stats[0], perm=perm,
                                             v_s_old=vo if kw.get("value_clip") else None)
            out.append((slabs.clone(), sc.clone()))
    finally:
        lib.tsm_debug_set_update_variant(0)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
