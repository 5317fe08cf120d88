"""CPU checks of the drop-in boundary: the C-ABI library loads here (no GPU) and exports every
symbol declared in include/tsmarl.h; argument validation that needs no device works."""
import ctypes as C
import os

import pytest

from tianshou_marl_amd import _abi, _build


@pytest.fixture(scope="module")
def lib():
    _build.build()
    return _abi.load()


def test_every_header_symbol_is_exported_and_bound(lib):
    declared = _abi.header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in tsmarl.h but not exported"
        assert name in _abi.SIGNATURES, f"{name} declared in tsmarl.h but not bound in _abi.SIGNATURES"
    for name in _abi.SIGNATURES:
        assert name in declared, f"{name} bound but not declared in tsmarl.h"


def test_abi_version_and_sizes(lib):
    # one number in three places: the header, the library, the binding (bumped with every signature / layout change)
    import re

    hdr = int(re.search(r"#define\s+TSM_ABI_VERSION\s+(\d+)", open(_abi.HEADER_PATH).read()).group(1))
    assert lib.tsm_abi_version() == hdr == _abi.ABI_VERSION == 4
    # ctypes mirrors of the header's structs keep their layout
    import ctypes

    assert ctypes.sizeof(_abi.tsm_slab_seg) == 64 and ctypes.sizeof(_abi.tsm_ppo_cfg) == 48
    assert lib.tsm_vrb_state_bytes(4, 1) == (6 * 4 + 4 + 1) * 8
    assert lib.tsm_vrb_state_bytes(8, 3) == (6 * 8 + 24 + 1) * 8
    assert lib.tsm_ppo_loss_partial_elems(0) == 0
    assert lib.tsm_ppo_loss_partial_elems(257) == 8


def test_argument_validation_maps_to_python_errors(lib):
    # invalid sizes are rejected on the host before any device work (no GPU needed)
    with pytest.raises(ValueError):
        _abi.call("tsm_gae_lanes", None, None, None, None, None, 1, -1, 4, 1, None, None, 0.99, 0.95, 1.0,
                  None, None, None)
    with pytest.raises(ValueError):  # n_lane not a multiple of lanes_per_env
        _abi.call("tsm_gae_lanes", 1, 1, 1, 1, 1, 1, 5, 7, 3, None, None, 0.99, 0.95, 1.0, 1, 1, None)
    with pytest.raises(ValueError):
        _abi.call("tsm_vrb_init", None, 0, 5, 1, None)
    cfg = _abi.tsm_ppo_cfg(0.2, 0.5, 0.5, 0.01, 0, 1)  # dual_clip must be > 1 (ppo.py:124-126)
    with pytest.raises(ValueError, match="Dual-clip"):
        _abi.call("tsm_ppo_loss_fwd_bwd", 1, 1, 1, 1, 1, 1, None, None, 0, 8, 5, 1, C.byref(cfg), 1, 1, 1, None)


def test_kernel_options_are_host_state_with_validated_values(lib):
    """tsm_kernel_option_get / _set need no device: every option ops.KERNEL_OPTIONS names exists, starts at its rule value 0
    (no TSM_* override in the test environment), takes its documented values and refuses others and unknown names."""
    from tianshou_marl_amd import ops

    assert set(ops.KERNEL_OPTIONS) == {"actor_tile", "split_bf16", "generic_kernels", "rollout_form"}
    for name, good, bad in [("actor_tile", (32, 64, 0), 48), ("split_bf16", (1, 0), 2), ("generic_kernels", (1, 0), 128),
                            ("rollout_form", (1, 2, 0), 3)]:
        if os.environ.get("TSM_" + name.upper()) is None:
            assert ops.kernel_option(name) == 0
        for v in good:
            ops.set_kernel_option(name, v)
            assert ops.kernel_option(name) == v
        with pytest.raises(ValueError):
            ops.set_kernel_option(name, bad)
        assert ops.kernel_option(name) == 0
    with pytest.raises(ValueError, match="unknown option"):
        ops.kernel_option("no_such_option")
    with ops.kernel_override(rollout_form=2, actor_tile=64):
        assert ops.kernel_options() == (64, 0, 0, 2)
    assert ops.kernel_options() == (0, 0, 0, 0)


def test_new_host_entry_points_validate_without_a_device(lib):
    """tsm_gather_fields / tsm_gae_set_scan_workspace check their arguments on the host; the Python wrappers refuse CPU tensors
    (no CPU fallback) and do not register a scan workspace without a GPU."""
    import torch

    from tianshou_marl_amd import ops

    assert lib.tsm_gae_scan_workspace_bytes() >= 8 + 128 * 4 + 128 * 16
    with pytest.raises(ValueError):
        _abi.call("tsm_gae_set_scan_workspace", 1, 8)           # too small
    _abi.call("tsm_gae_set_scan_workspace", None, 0)            # withdrawing is always allowed
    fields = (_abi.tsm_gather_field * 1)(_abi.tsm_gather_field(None, None, 4, 0, 0, 1, 0, 1, 0, 0, 0))
    with pytest.raises(ValueError, match="null pointer"):
        _abi.call("tsm_gather_fields", fields, 1, None)
    with pytest.raises(ValueError, match="1..12 fields"):
        _abi.call("tsm_gather_fields", fields, 13, None)
    fields[0] = _abi.tsm_gather_field(1, 1, 6, 2, 2, 1, 0, 1, 0, 0, 0)   # n_rows != T * E
    with pytest.raises(ValueError, match="T \\* E"):
        _abi.call("tsm_gather_fields", fields, 1, None)
    fields[0] = _abi.tsm_gather_field(1, 1, 4, 0, 0, 1, 0, 1, 0, 1, 0)   # f32 -> i32 is not a conversion it does
    with pytest.raises(ValueError, match="f32 sources"):
        _abi.call("tsm_gather_fields", fields, 1, None)
    with pytest.raises(ValueError, match="device tensors"):
        ops.gather_fields([(torch.zeros(4), torch.zeros(4))])
    if not torch.cuda.is_available():
        assert ops.ensure_scan_workspace("cpu") is False


def test_product_has_no_oracle_import():
    """The product package must never import the oracle (test infrastructure)."""
    root = os.path.dirname(os.path.abspath(_abi.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "ref_shim" not in src, f


def test_ops_refuse_cpu_tensors():
    import torch

    from tianshou_marl_amd import ops

    x = torch.zeros(4, 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.gae_lanes(x, x, x, x.to(torch.uint8), x.to(torch.uint8))
