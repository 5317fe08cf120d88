"""GPU parity of the 128-wide kernels against the REFERENCE's own tensors.

`tests/golden/ppo_update_wide.npz` (make_fixtures.py::make_ppo_update_wide: the reference's `PPO` with
`Net(hidden_sizes=[128, 128])` actor and critic on 48-wide observations -- modelfree/ppo.py:164-224,
utils/net/common.py:246-369 -- run in the build container) holds the buffer rows, the `np.random.permutation` draws of
`Batch.split`, the `_preprocess_batch` outputs, the loss statistics of every gradient step and the post-update weights of
    w128_mb64            batch 64, repeat 2 (200 rows -> 64, 64, 72)
    w128_vclip_gn        value clipping + max_grad_norm 0.5, batch 64, repeat 2
    w128_dualclip_full   dual clip 2.0 + value clipping, the whole batch in one step
    w128_recompute       recompute_advantage (ppo.py:174-178), batch 64, repeat 2
They are replayed through `GenericPPO` with the one-launch actor / critic steps, the one-launch critic forward
(csrc/critic_rows.hip) and the segmented Adam, captured (hipGraph) and eager: loss statistics 2e-5, weights 1e-5.
Every replay runs once per ACTOR KERNEL behind `tsm_ppo_actor_rows_update`: the 32-sample tiles of csrc/ppo_rows.hip (what the
size rule picks for these 64..200-sample minibatches) and the 64-sample tiles of csrc/actor_rows64.hip (what it picks at the
bench's 65 536-sample minibatches), forced through the `actor_tile` kernel option -- so the kernel behind the roofline figure
meets the reference's tensors, partial tiles included (64 -> one full tile, 72 -> 64 + 8, 200 -> 3 x 64 + 8)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tianshou_marl_amd import ops
    from tianshou_marl_amd.algorithm import GenericPPO, policy_within_training_step
    from tianshou_marl_amd.data.batch import Batch
    from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
    from tianshou_marl_amd.utils.net import MLPActorCritic

DEV = "cuda"
N_ENV, T, D, H, A = 8, 25, 48, 128, 5
VARIANTS = ["w128_mb64", "w128_vclip_gn", "w128_dualclip_full", "w128_recompute"]
TILES = [32, 64]


def _layers(g, p, kind, net):
    return [(g[p + f"{kind}{net}_w{i}"], g[p + f"{kind}{net}_b{i}"]) for i in range(3)]


def _flat(g, p, kind):
    return np.concatenate([np.concatenate([W.reshape(-1), b.reshape(-1)]) for net in ("actor", "critic")
                           for W, b in _layers(g, p, kind, net)]).astype(np.float32)


def _assert_weights(got, g, p):
    """Post-update weights vs the reference's: rtol 1e-5 + atol 5e-6.  A run of ONE Adam step from zero moments moves a
    parameter by u = lr g / (|g| + eps), and du / dg = lr eps / (|g| + eps)^2 blows up where |g| ~ eps = 1e-8 (a handful of
    parameters whose gradient all but cancels): there an f32 summation-order difference of 1e-9 in g is allowed its
    lr eps 1e-9 / (|g| + eps)^2 <= 3e-5 on the weight (the reference's own gradient of that step is in the fixture)."""
    ref = _flat(g, p, "after_")
    tol = 5e-6 + 1e-5 * np.abs(ref)
    if int(g[p + "gradient_steps"]) == 1:
        gr = np.concatenate([np.concatenate([g[p + f"last_{net}_gw{i}"].reshape(-1), g[p + f"last_{net}_gb{i}"].reshape(-1)])
                             for net in ("actor", "critic") for i in range(3)]).astype(np.float64)
        lr, eps = float(g[p + "ppo_cfg"][9]), 1e-8
        tol = tol + lr * eps * 1e-9 / (np.abs(gr) + eps) ** 2
    bad = np.abs(got.astype(np.float64) - ref) > tol
    assert not bad.any(), (int(bad.sum()), float(np.abs(got - ref)[bad].max()))


def _job(g, name, graph, **kw):
    p = name + "_"
    eps_clip, dual_clip, value_clip, adv_norm, vf_coef, ent_coef, gamma, lam, max_gn, lr = g[p + "ppo_cfg"]
    net = MLPActorCritic(D, A, (H, H), device=DEV)
    net.actor.load_layers(_layers(g, p, "", "actor"))
    net.critic.load_layers(_layers(g, p, "", "critic"))
    algo = GenericPPO(net=net, lr=float(lr), eps_clip=float(eps_clip), dual_clip=float(dual_clip) or None,
                      value_clip=bool(value_clip), advantage_normalization=bool(adv_norm), vf_coef=float(vf_coef),
                      ent_coef=float(ent_coef), gamma=float(gamma), gae_lambda=float(lam), max_grad_norm=float(max_gn) or None,
                      recompute_advantage=bool(g[p + "recompute_advantage"]), dispatch="pooled", shuffle="numpy", graph=graph, **kw)
    assert np.array_equal(g[p + "indices"], np.arange(N_ENV * T))  # env-major, time-ordered rows (sample(0))
    buf = DeviceVectorReplayBuffer(N_ENV * T, N_ENV, 1, D, device=DEV)
    rows = lambda k, t: g[p + k].reshape(N_ENV, T, *g[p + k].shape[1:])[:, t]  # noqa: E731
    for t in range(T):
        buf.add(Batch(obs=rows("obs", t), act=rows("act", t), rew=rows("rew", t), terminated=rows("terminated", t),
                      truncated=rows("truncated", t), obs_next=rows("obs_next", t)))
    return algo, net, buf


@pytest.mark.parametrize("name", VARIANTS)
def test_preprocess_batch_of_the_wide_nets_matches_reference(golden_dir, name):
    """a2c.py:113-151 + ppo.py:146-162 through the one-launch critic forward: v_s, returns, advantages, logp_old."""
    g = np.load(os.path.join(golden_dir, "ppo_update_wide.npz"), allow_pickle=True)
    p = name + "_"
    algo, net, buf = _job(g, name, graph=False)
    assert algo.fused_actor and algo.fused_critic and algo.fused_values
    pb = algo._preprocess_batch(buf)
    em = lambda x: x.view(T, N_ENV).t().reshape(-1).cpu().numpy()  # noqa: E731  time-major lanes -> env-major rows
    np.testing.assert_allclose(em(pb["v_s"]), g[p + "v_s"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(em(pb["logp_old"]), g[p + "logp_old"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(em(pb["ret"]), g[p + "returns"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(em(pb["adv"]), g[p + "adv"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tile", TILES, ids=["tile32", "tile64"])
@pytest.mark.parametrize("graph", [True, False], ids=["graph", "eager"])
@pytest.mark.parametrize("name", VARIANTS)
def test_update_of_the_wide_nets_replays_reference_minibatch_loop(golden_dir, name, graph, tile):
    """ppo.py:164-224 with the reference's permutations on the row kernels: loss statistics of every gradient step and
    the weights after the last Adam step.  The captured form needs a warm-up update of the same shape (the first one of a
    shape runs eagerly, the second captures): it runs on a twin and the measured object replays a graph of its own."""
    g = np.load(os.path.join(golden_dir, "ppo_update_wide.npz"), allow_pickle=True)
    p = name + "_"
    with ops.kernel_override(actor_tile=tile):
        # the slab count says which kernel serves a minibatch: one slab per tile of the forced size
        assert ops.ppo_actor_rows_grid(72) == -(-72 // tile) and ops.ppo_actor_rows_grid(200) == -(-200 // tile)
        algo, net, buf = _job(g, name, graph)
        bs, rep = int(g[p + "batch_size"]), int(g[p + "repeat"])
        bs = None if bs == -1 else bs
        captured = lambda: any(isinstance(k, tuple) and k and k[0] == "ggraph" and "graph" in v  # noqa: E731
                               for k, v in algo._ws.items())
        if graph:  # eager update, then the capturing one, on throw-away parameters; then restore and replay
            p0 = net.flat.data.clone()
            for _ in range(2):
                with policy_within_training_step(algo):
                    algo.update(buf, bs, rep)
            # recompute_advantage re-runs the critic passes between repeats from the host: the update stays on eager
            # launches by design (GenericPPO._update), whatever `graph` says
            assert captured() == (not algo.recompute_adv)
            net.flat.data.copy_(p0)
            algo.exp_avg.zero_()
            algo.exp_avg_sq.zero_()
            algo.opt_step = 0
            algo.param_version += 1
        np.random.seed(11)  # the state make_ppo_update_wide drew `perms` from
        with policy_within_training_step(algo):
            stats = algo.update(buf, bs, rep)
    assert ops.kernel_option("actor_tile") == 0
    assert stats.gradient_steps == int(g[p + "gradient_steps"])
    for k in ("loss", "actor_loss", "vf_loss", "ent_loss"):
        s = getattr(stats, k)
        np.testing.assert_allclose([s.mean, s.std, s.max, s.min], g[p + "stat_" + k], rtol=2e-5, atol=2e-6, err_msg=k)
    _assert_weights(net.flat.data.cpu().numpy(), g, p)


def test_dense_and_row_kernel_paths_agree_on_the_reference_run(golden_dir):
    """The same reference run through the dense GEMM composition (fused_actor=False): both ways to the reference's
    weights, so the one-launch kernels are checked against the reference AND against the builder's other path."""
    g = np.load(os.path.join(golden_dir, "ppo_update_wide.npz"), allow_pickle=True)
    p = "w128_vclip_gn_"
    algo, net, buf = _job(g, "w128_vclip_gn", graph=False, fused_actor=False)
    assert not algo.fused_actor and not algo.fused_values
    np.random.seed(11)
    with policy_within_training_step(algo):
        stats = algo.update(buf, 64, 2)
    assert stats.gradient_steps == int(g[p + "gradient_steps"])
    np.testing.assert_allclose(net.flat.data.cpu().numpy(), _flat(g, p, "after_"), rtol=1e-5, atol=5e-6)
