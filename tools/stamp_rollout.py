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
env = DeviceSimpleSpreadVectorEnv(1024, 3, device=dev); net = DiscreteActorCritic(18, 5, 64, device=dev, seed=0)
algo = PPO(net=net); buf = DeviceVectorReplayBuffer(1024 * 25, 1024, 3, 18, device=dev); col = Collector(algo, env, buf); col.reset()
st = torch.zeros(64, dtype=torch.int64, device=dev)
lib = _abi.load(); lib.tsm_debug_set_stamps.argtypes = [ctypes.c_void_p]; lib.tsm_debug_set_stamps(st.data_ptr())
with policy_within_training_step(algo):
    for _ in range(3):
        col.collect(n_step=1024 * 25); col.reset_buffer(keep_statistics=True)
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(8, 8)
names = ["A obs", "B fwd", "C head", "D env+idx", "E scatter", "F done", "next"]
for t in range(1, 4):
    d = [(s[t][k + 1] - s[t][k]) / 100.0 for k in range(6)]  # 100 MHz wall clock -> us
    print("step", t, {names[k]: round(d[k], 2) for k in range(6)}, "total", round((s[t][6] - s[t][0]) / 100.0, 2))
