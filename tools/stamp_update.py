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
st = torch.zeros(1024, dtype=torch.int64, device=dev)
lib = _abi.load(); lib.tsm_debug_set_stamps.argtypes = [ctypes.c_void_p]; lib.tsm_debug_set_stamps(st.data_ptr())
names = ["stage", "gather X", "forward", "loss head", "backward", "slab write"]
for label, pm in (("perm gather", perm), ("contiguous rows", None)):
    for _ in range(5):
        ops.ppo_update_fused(P, obs, act, logp, adv, ret, cfg, A, H, adv_stats=stats[0], perm=pm, M=M, image=net.image)
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    print(label, {names[k]: round((s[k + 1] - s[k]) / 100.0, 2) for k in range(6)}, "total", round((s[6] - s[0]) / 100.0, 2))
import numpy as np
b = s[64:64 + 512].reshape(256, 2).astype(np.float64) / 100.0
t0 = b[:, 0].min()
print("workgroup start (us after first): min/med/max", np.round(np.percentile(b[:, 0] - t0, [0, 50, 100]), 2))
print("workgroup end   (us after first start): min/med/max", np.round(np.percentile(b[:, 1] - t0, [0, 50, 100]), 2))
print("workgroup duration: min/med/max", np.round(np.percentile(b[:, 1] - b[:, 0], [0, 50, 100]), 2))
