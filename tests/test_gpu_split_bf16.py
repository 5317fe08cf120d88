"""EXPERIMENTAL path, opt-in (TSM_SPLIT_BF16=1): layer 1 of the one-launch critic forward (csrc/critic_rows.hip,
`critic_rows_forward_bf16x6_kernel`) on the bf16 matrix pipe with three-way split operands -- six bf16 MFMAs per 32 k in place
of eight f32 ones.  Replaces nothing by default; the switch is read once per process, so the path runs in a child process.
Checked here: the values agree with float64 as tightly as the f32 path's do (1e-5 of the value scale is the bar of
/root/reference/tianshou/algorithm/modelfree/a2c.py:121-127's critic passes), and they are NOT the f32 path's bits (the path ran)."""
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tianshou_marl_amd import ops
from tianshou_marl_amd.utils.net import FlatMLP
out = {}
for K1, Mr, n_out in ((384, 5000, 1), (48, 333, 1), (200, 1000, 8), (20, 70, 3)):
    torch.manual_seed(K1)
    f = FlatMLP([K1, 128, 128, n_out], device="cuda", seed=K1)
    x = torch.randn(Mr + 5, K1, device="cuda") * 1.7
    rows = torch.randperm(Mr + 5, device="cuda")[:Mr].contiguous()
    v = ops.critic_rows_forward(f.flat.data, x, rows=rows, n_out=n_out)
    w = [t.double() for t in (f.weight(0), f.bias(0), f.weight(1), f.bias(1), f.weight(2), f.bias(2))]
    h = torch.relu(x[rows].double() @ w[0].T + w[1]); h = torch.relu(h @ w[2].T + w[3])
    ref = (h @ w[4].T + w[5]).mean(1)
    out[f"v_{K1}"] = v.cpu().numpy(); out[f"ref_{K1}"] = ref.cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def _run(tmp_path, flag):
    env = dict(os.environ, TSM_SPLIT_BF16=flag)
    path = str(tmp_path / f"out{flag}.npz")
    r = subprocess.run([sys.executable, "-c", CHILD, path], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)


def test_split_bf16_layer_1_matches_float64_like_the_f32_path(tmp_path):
    a, b = _run(tmp_path, "0"), _run(tmp_path, "1")
    for K1 in (384, 48, 200, 20):
        ref = a[f"ref_{K1}"]
        scale = np.abs(ref).max()
        e32 = np.abs(a[f"v_{K1}"] - ref).max() / scale
        e16 = np.abs(b[f"v_{K1}"] - ref).max() / scale
        assert e32 < 1e-5 and e16 < 1e-5, (K1, e32, e16)
        assert e16 < 4 * e32 + 2e-7, (K1, e32, e16)          # as tight as the f32 matrix pipe's own rounding
        assert not np.array_equal(a[f"v_{K1}"], b[f"v_{K1}"]), K1   # (the experimental kernel did run)
