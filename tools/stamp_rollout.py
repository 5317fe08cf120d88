import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_marl_amd import _abi
from tianshou_marl_amd.algorithm.ppo import PPO, policy_within_training_step
from tianshou_marl_amd.data.buffer import DeviceVectorReplayBuffer
from tianshou_marl_amd.data.collector import Collector
from tianshou_marl_amd.env.mpe import DeviceSimpleSpreadVectorEnv
from tianshou_marl_amd.utils.net import DiscreteActorCritic
dev = "cuda"
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024   # 4096 = BASELINE configs[1]
env = DeviceSimpleSpreadVectorEnv(E, 3, device=dev); net = DiscreteActorCritic(18, 5, 64, device=dev, seed=0)
algo = PPO(net=net); buf = DeviceVectorReplayBuffer(E * 25, E, 3, 18, device=dev); col = Collector(algo, env, buf); col.reset()
st = torch.zeros(1024, dtype=torch.int64, device=dev)
lib = _abi.load(); lib.tsm_debug_set_stamps.argtypes = [ctypes.c_void_p]; lib.tsm_debug_set_stamps(st.data_ptr())
with policy_within_training_step(algo):
    for _ in range(300):   # (the stamps kept are those of the LAST launch: the clock the part settles to under this load, not a cold one)
        col.collect(n_step=E * 25); col.reset_buffer(keep_statistics=True)
torch.cuda.synchronize()
full = st.cpu().numpy()
s = full[:64].reshape(8, 8)
names = ["A obs", "B fwd", "C head", "D env+idx", "E scatter", "F done", "next"]
for t in range(1, 4):
    d = [(s[t][k + 1] - s[t][k]) / 100.0 for k in range(6)]  # 100 MHz wall clock -> us
    print("step", t, {names[k]: round(d[k], 2) for k in range(6)}, "total", round((s[t][6] - s[t][0]) / 100.0, 2))

import numpy as np
blk = full[64:64 + 2 * 205].reshape(-1, 2)
blk = blk[blk[:, 0] > 0]
t0 = blk[:, 0].min()
print("workgroup start offsets us: min %.2f max %.2f" % ((blk[:, 0].min() - t0) / 100, (blk[:, 0].max() - t0) / 100))
dur = (blk[:, 1] - blk[:, 0]) / 100.0
print("workgroup durations us: min %.1f median %.1f max %.1f; kernel span %.1f" % (dur.min(), np.median(dur), dur.max(), (blk[:, 1].max() - t0) / 100))
print("  by blockIdx %% 8 (XCD): " + ", ".join("%.1f" % dur[x::8].mean() for x in range(8)))
print("  sorted: " + " ".join("%.0f" % v for v in sorted(dur)))
order = np.argsort(dur)
print("  slowest workgroups (id: us): " + ", ".join("%d: %.0f" % (i, dur[i]) for i in order[-12:]) + "; fastest: " + ", ".join("%d: %.0f" % (i, dur[i]) for i in order[:8]))
steps = full[640:640 + 27]
print("workgroup 0 step durations us:", [round((steps[i + 1] - steps[i]) / 100.0, 1) for i in range(25)])
x = full[900:964].reshape(8, 8)
for t in range(1, 3):
    seq = [s[t][3]] + [x[t][k] for k in range(5)] + [s[t][4]]
    if x[t][5] > 0:
        print("  move: wave 0's own part %.2f us, then %.2f us at the barrier (the other wave's pair tasks)" % ((x[t][5] - s[t][3]) / 100.0, (x[t][0] - x[t][5]) / 100.0))
    print("D sub-phases", [round((seq[i + 1] - seq[i]) / 100.0, 2) for i in range(6)], "(move, publish, min-dist+penalty, obs_next, reward, index algebra)")
