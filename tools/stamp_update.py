import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import _abi, ops
from tianshou_marl_amd.utils.net import DiscreteActorCritic
dev = "cuda"
D, A, H, n, M = 18, 5, 64, 76800, 4096
net = DiscreteActorCritic(D, A, H, device=dev, seed=0)
P = net.flat.data
obs = torch.randn(n, D, device=dev); act = torch.randint(0, A, (n,), dtype=torch.int32, device=dev)
logp, adv, ret = (torch.randn(n, device=dev) for _ in range(3))
perm = torch.randperm(n, device=dev)[:M].contiguous()
stats = ops.ppo_adv_stats(adv, torch.tensor([0, M], device=dev), perm=perm)
cfg = ops.make_ppo_cfg()
st = torch.zeros(2048, dtype=torch.int64, device=dev)
lib = _abi.load(); lib.tsm_debug_set_stamps.argtypes = [ctypes.c_void_p]; lib.tsm_debug_set_stamps(st.data_ptr())
import numpy as np
names = ["prologue (ids + image, gathers, image -> LDS)", "commit X", "L1", "L2", "L3", "loss head", "dW3 / dH2", "dW2 / dH1", "dW1", "slab stores", "loss sums"]
for label, pm in (("perm gather", perm),):
    for _ in range(3000):   # (the stamps kept are the last launch's: the clock the part settles to under this load)
        ops.ppo_update_fused(P, obs, act, logp, adv, ret, cfg, A, H, adv_stats=stats[0], perm=pm, M=M, image=net.image)
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    for net_i, nm in ((0, "actor"), (1, "critic")):
        b = s[16 + 24 * net_i: 16 + 24 * net_i + 12]
        print(f"workgroup (0, {nm}): " + ", ".join(f"{names[k]} {(b[k + 1] - b[k]) / 100.0:.2f}" for k in range(11)) + f"; total {(b[11] - b[0]) / 100.0:.2f} us")
    wg = s[1024:1024 + 1024].reshape(512, 2).astype(np.float64) / 100.0
    t0 = wg[:, 0].min()
    life = wg[:, 1] - wg[:, 0]
    print(f"workgroups (256 x 2 nets): starts spread over {wg[:, 0].max() - t0:.2f} us; first start -> last end {wg[:, 1].max() - t0:.2f} us; lifetime min / median / max "
          f"{life.min():.2f} / {np.median(life):.2f} / {life.max():.2f} us (actor median {np.median(life[:256]):.2f}, critic {np.median(life[256:]):.2f})")
