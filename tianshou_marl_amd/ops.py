"""Tensor-level front-end of the C-ABI (one thin function per entry point of include/tsmarl.h).

Inputs/outputs are torch tensors that live in HBM; every function launches hand-written HIP
kernels through `_abi.call` on torch's current stream.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _abi
from ._abi import call, ptr, stream_ptr, tsm_field, tsm_ppo_cfg


def _chk(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t.contiguous()


# --------------------------------------------------------------------------------------------
# GAE  (algorithm_base.py:651-717,1079-1134; a2c.py:132-146)
# --------------------------------------------------------------------------------------------
def gae_lanes(v_s, v_s_next, rew, terminated, truncated, gamma=0.99, gae_lambda=0.95, v_scale=1.0,
              lanes_per_env=1, env_start=None, env_len=None, out=None, rms=None, rms_eps=1e-8):
    """Per-lane GAE on time-major tensors [T, ...lanes]; returns (returns, adv) f32 tensors.

    terminated/truncated: u8/bool, either the lane shape [T, n_lane] or env-level [T, n_env].
    rms: device f64[3] {mean, var, count} of the return statistics -> v_scale = sqrt(var + rms_eps) read on device.
    """
    v_s = _chk(v_s, torch.float32, "v_s")
    T = v_s.shape[0]
    L = v_s[0].numel() if T > 0 else 0
    if T > 4096 and L <= 64:
        ensure_scan_workspace(v_s.device)
    v_s_next = _chk(v_s_next, torch.float32, "v_s_next")
    rew = _chk(rew, torch.float32, "rew")
    term = terminated.contiguous().view(torch.uint8) if terminated.dtype == torch.bool else _chk(terminated, torch.uint8, "terminated")
    trunc = truncated.contiguous().view(torch.uint8) if truncated.dtype == torch.bool else _chk(truncated, torch.uint8, "truncated")
    if v_s_next.shape != v_s.shape or rew.shape != v_s.shape:
        raise ValueError("gae_lanes: v_s, v_s_next, rew must have identical shapes")
    flags_per_lane = 1 if term.numel() == v_s.numel() else 0
    if not flags_per_lane and term.numel() * lanes_per_env != v_s.numel():
        raise ValueError("gae_lanes: flag tensors must be lane-shaped or env-shaped")
    if trunc.numel() != term.numel():
        raise ValueError("gae_lanes: terminated/truncated shape mismatch")
    if out is None:
        ret, adv = torch.empty_like(v_s), torch.empty_like(v_s)
    else:
        ret, adv = out
    if rms is not None:
        call("tsm_gae_lanes_rms", ptr(v_s), ptr(v_s_next), ptr(rew), ptr(term), ptr(trunc), flags_per_lane, T, L,
             lanes_per_env, ptr(env_start), ptr(env_len), float(gamma), float(gae_lambda),
             ptr(_chk(rms, torch.float64, "rms")), float(rms_eps), ptr(ret), ptr(adv), stream_ptr())
        return ret, adv
    call("tsm_gae_lanes", ptr(v_s), ptr(v_s_next), ptr(rew), ptr(term), ptr(trunc), flags_per_lane, T, L,
         lanes_per_env, ptr(env_start), ptr(env_len), float(gamma), float(gae_lambda), float(v_scale),
         ptr(ret), ptr(adv), stream_ptr())
    return ret, adv


_scan_ws: dict = {}   # device index -> (workspace, pinned error word)


def ensure_scan_workspace(device) -> bool:
    """Register the workspace of the parallel long-series GAE scan (include/tsmarl.h: tsm_gae_set_scan_workspace) once per DEVICE;
    not while a stream is capturing (the allocation would belong to that graph's pool).  Callers that capture graphs call it
    beforehand (PPO._warm_kernels)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    if idx not in _scan_ws and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        n = int(call("tsm_gae_scan_workspace_bytes"))
        with torch.cuda.device(idx):
            ws = torch.zeros(n, dtype=torch.uint8, device=torch.device("cuda", idx))
            err = torch.zeros(1, dtype=torch.int32, pin_memory=True)
            call("tsm_gae_set_scan_workspace", ws.data_ptr(), n)
            call("tsm_gae_set_scan_error_word", err.data_ptr())
        _scan_ws[idx] = (ws, err)
    return idx in _scan_ws


def gae_scan_failed() -> bool:
    """True once a parallel long-series scan gave up waiting for another workgroup's map (its returns are NaN from there on).
    A plain read of pinned host words: meaningful after the host has waited for the launch (the statistics' event)."""
    return any(int(err[0]) != 0 for _, err in _scan_ws.values())


def rms_update(returns, rms, rms_eps=1e-8, ids=None, work=None):
    """RunningMeanStd.update(returns * sqrt(var + eps)) on device (a2c.py:144-146); rms f64[3] is updated in place."""
    returns = _chk(returns, torch.float32, "returns").reshape(-1)
    n = returns.numel() if ids is None else ids.numel()
    if work is None:
        work = torch.empty(call("tsm_rms_update_work_elems", n), dtype=torch.float64, device=returns.device)
    call("tsm_rms_update", ptr(returns), ptr(ids), n, ptr(_chk(rms, torch.float64, "rms")), float(rms_eps), ptr(work),
         stream_ptr())
    return rms


def mc_return_to_go_lanes(rew, gamma=0.99):
    rew = _chk(rew, torch.float32, "rew")
    out = torch.empty_like(rew)
    T = rew.shape[0]
    call("tsm_mc_return_to_go_lanes", ptr(rew), T, rew[0].numel() if T else 0, float(gamma), ptr(out), stream_ptr())
    return out


# --------------------------------------------------------------------------------------------
# VectorReplayBuffer index state (manager.py / buffer_base.py)
# --------------------------------------------------------------------------------------------
class VrbState:
    """Device-resident episode/index bookkeeping of a VectorReplayBuffer(total_size, buffer_num)."""

    def __init__(self, total_size: int, buffer_num: int, rew_dim: int = 1, device="cuda"):
        self.buffer_num = int(buffer_num)
        self.sub_size = -(-int(total_size) // self.buffer_num)  # ceil, vecbuf.py:35
        self.maxsize = self.sub_size * self.buffer_num
        self.rew_dim = max(1, int(rew_dim))
        self.device = torch.device(device)
        nbytes = call("tsm_vrb_state_bytes", self.buffer_num, self.rew_dim)
        self.state = torch.zeros(nbytes // 8, dtype=torch.int64, device=self.device)
        self.done_store = torch.zeros(self.sub_size, self.buffer_num, dtype=torch.uint8, device=self.device)
        self._scratch = torch.zeros(self.buffer_num + 1, dtype=torch.int64, device=self.device)
        self._n_out = torch.zeros(1, dtype=torch.int64, device=self.device)
        call("tsm_vrb_init", ptr(self.state), self.buffer_num, self.sub_size, self.rew_dim, stream_ptr())

    # views into the packed state (i64 [6][B] | f64 [B][D] | i64 flag)
    def _i64(self, k):
        B = self.buffer_num
        return self.state[k * B:(k + 1) * B]

    insertion_idx = property(lambda s: s._i64(0))
    size = property(lambda s: s._i64(1))
    last_index = property(lambda s: s._i64(4))
    lengths = property(lambda s: s._i64(5))

    def __len__(self) -> int:
        return int(self.lengths.sum().item())

    def reset(self, keep_statistics: bool = False) -> None:
        call("tsm_vrb_reset", ptr(self.state), self.buffer_num, self.sub_size, self.rew_dim,
             int(keep_statistics), stream_ptr())

    def add(self, rew, done, buffer_ids=None, fields=(), outs=None):
        """manager.py:131-193.  fields: iterable of (src[R, ...], dst[sub_size, buffer_num, ...]).
        outs: optional preallocated (ptr i64[R], ep_rew f64[R,D], ep_len i64[R], ep_idx i64[R])."""
        rew = _chk(rew, torch.float32, "rew").reshape(rew.shape[0], -1)
        R = rew.shape[0]
        if rew.shape[1] != self.rew_dim:
            raise ValueError(f"rew has {rew.shape[1]} columns, buffer was created with rew_dim={self.rew_dim}")
        done = done.contiguous().view(torch.uint8) if done.dtype == torch.bool else _chk(done, torch.uint8, "done")
        ids = None if buffer_ids is None else _chk(buffer_ids, torch.int64, "buffer_ids")
        dev = self.device
        if outs is not None:
            ptr_out, ep_rew, ep_len, ep_idx = outs
        else:
            ptr_out = torch.empty(R, dtype=torch.int64, device=dev)
            ep_rew = torch.empty(R, self.rew_dim, dtype=torch.float64, device=dev)
            ep_len = torch.empty(R, dtype=torch.int64, device=dev)
            ep_idx = torch.empty(R, dtype=torch.int64, device=dev)
        farr = (tsm_field * max(1, len(fields)))()
        keep = []
        for i, (src, dst) in enumerate(fields):
            src = src.contiguous()
            keep.append(src)
            rb = src[0].numel() * src.element_size() if R else dst[0, 0].numel() * dst.element_size()
            if dst[0, 0].numel() * dst.element_size() != rb:
                raise ValueError(f"field {i}: row size mismatch between source and store")
            farr[i] = tsm_field(ptr(src), ptr(dst), rb)
        call("tsm_vrb_add", ptr(self.state), self.buffer_num, self.sub_size, self.rew_dim, ptr(ids), R,
             ptr(rew), ptr(done), ptr(self.done_store), farr, len(fields), ptr(ptr_out), ptr(ep_rew),
             ptr(ep_len), ptr(ep_idx), stream_ptr())
        return ptr_out, ep_rew, ep_len, ep_idx

    def check(self) -> None:
        call("tsm_vrb_check", ptr(self.state), self.buffer_num, self.rew_dim, stream_ptr())

    def sample_indices_all(self) -> torch.Tensor:
        out = torch.empty(self.maxsize, dtype=torch.int64, device=self.device)
        call("tsm_vrb_sample_indices_all", ptr(self.state), self.buffer_num, self.sub_size, ptr(out),
             ptr(self._n_out), ptr(self._scratch), stream_ptr())
        return out[: int(self._n_out.item())]

    def unfinished_index(self) -> torch.Tensor:
        out = torch.empty(self.buffer_num, dtype=torch.int64, device=self.device)
        call("tsm_vrb_unfinished_index", ptr(self.state), self.buffer_num, self.sub_size,
             ptr(self.done_store), ptr(out), ptr(self._n_out), stream_ptr())
        return out[: int(self._n_out.item())]

    def _pn(self, name, index):
        index = _chk(torch.as_tensor(index, device=self.device), torch.int64, "index").reshape(-1)
        out = torch.empty_like(index)
        call(name, ptr(self.state), self.buffer_num, self.sub_size, ptr(self.done_store), ptr(index),
             index.numel(), ptr(out), stream_ptr())
        return out

    def prev(self, index):
        return self._pn("tsm_vrb_prev", index)

    def next(self, index):
        return self._pn("tsm_vrb_next", index)

    def gather(self, store: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
        """ReplayBuffer.__getitem__ for one field: rows of a [sub_size, buffer_num, ...] store by flat index."""
        index = _chk(index, torch.int64, "index").reshape(-1)
        row_shape = store.shape[2:]
        out = torch.empty((index.numel(), *row_shape), dtype=store.dtype, device=store.device)
        rb = store[0, 0].numel() * store.element_size()
        call("tsm_vrb_gather", ptr(store), self.buffer_num, self.sub_size, rb, ptr(index), index.numel(),
             ptr(out), stream_ptr())
        return out


# --------------------------------------------------------------------------------------------
# agent dispatch (marl.py:148,170-180,233)
# --------------------------------------------------------------------------------------------
def agent_index(agent_id: torch.Tensor, n_agent: int):
    """Stable partition of row numbers by agent -> (index[B] i64, offsets[n_agent+1] i64)."""
    agent_id = _chk(agent_id, torch.int32, "agent_id").reshape(-1)
    B = agent_id.numel()
    dev = agent_id.device
    index = torch.empty(B, dtype=torch.int64, device=dev)
    offsets = torch.empty(n_agent + 1, dtype=torch.int64, device=dev)
    n_blocks = max(1, -(-B // 1024))
    scratch = torch.empty(n_agent * n_blocks + 1, dtype=torch.int64, device=dev)
    call("tsm_agent_index", ptr(agent_id), B, n_agent, ptr(index), ptr(offsets), ptr(scratch), stream_ptr())
    return index, offsets


def scatter_rows(src: torch.Tensor, index: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst[index[i]] = src[i]"""
    src = src.contiguous()
    n = index.numel()
    rb = (src[0].numel() if n else 1) * src.element_size()
    call("tsm_scatter_rows", ptr(src), ptr(_chk(index, torch.int64, "index")), n, rb, ptr(dst), stream_ptr())
    return dst


def gather_rows(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """out[i] = src[index[i]]"""
    src = src.contiguous()
    n = index.numel()
    out = torch.empty((n, *src.shape[1:]), dtype=src.dtype, device=src.device)
    rb = src[0].numel() * src.element_size()
    call("tsm_gather_rows", ptr(src), ptr(_chk(index, torch.int64, "index")), n, rb, ptr(out), stream_ptr())
    return out


# --------------------------------------------------------------------------------------------
# categorical head (discrete.py:22-24; reinforce.py:183-189)
# --------------------------------------------------------------------------------------------
def categorical_sample(logits, seed: int, offset: int = 0, deterministic: bool = False, want_logp: bool = True,
                       offset_dev=None, out=None):
    """Categorical(logits).sample() / .mode + log-prob; `out` = optional preallocated (act i32[B], logp f32[B])."""
    logits = _chk(logits, torch.float32, "logits")
    B, A = logits.shape
    if out is not None:
        act, logp = out
    else:
        act = torch.empty(B, dtype=torch.int32, device=logits.device)
        logp = torch.empty(B, dtype=torch.float32, device=logits.device) if want_logp else None
    call("tsm_categorical_sample", ptr(logits), B, A, seed & (2**64 - 1), offset & (2**64 - 1), ptr(offset_dev),
         int(deterministic), ptr(act), ptr(logp), stream_ptr())
    return act, logp


def categorical_logp_entropy(logits, act):
    logits = _chk(logits, torch.float32, "logits")
    act = _chk(act, torch.int32, "act")
    B, A = logits.shape
    logp = torch.empty(B, dtype=torch.float32, device=logits.device)
    ent = torch.empty(B, dtype=torch.float32, device=logits.device)
    call("tsm_categorical_logp_entropy", ptr(logits), ptr(act), B, A, ptr(logp), ptr(ent), stream_ptr())
    return logp, ent


# --------------------------------------------------------------------------------------------
# PPO loss (ppo.py:182-211)
# --------------------------------------------------------------------------------------------
def make_ppo_cfg(eps_clip=0.2, dual_clip=None, value_clip=False, adv_norm=True, vf_coef=0.5, ent_coef=0.01,
                 loss_kind=0, value_group=1):
    """loss_kind 0: PPO clip objective; 1: plain policy gradient -mean(logp * adv) (A2C / Reinforce); 2: the value term
    alone (ppo_value_loss).  value_group N > 1: one critic value per joint row of N agents (centralized critic)."""
    if loss_kind not in (0, 1, 2):
        raise ValueError(f"loss_kind must be 0 (PPO clip), 1 (policy gradient) or 2 (value term only), got {loss_kind}")
    return tsm_ppo_cfg(float(eps_clip), float(dual_clip or 0.0), float(vf_coef), float(ent_coef),
                       int(bool(value_clip)), int(bool(adv_norm)), int(loss_kind), int(value_group))


def ppo_adv_stats(adv, mb_start, perm=None, out=None, max_rows: int = 0, work=None):
    """Per-minibatch (mean, unbiased std) of adv[perm[mb_start[k]:mb_start[k+1]]] -> [n_mb, 2] f32.
    max_rows: host knowledge of the longest minibatch; above 8192 rows the chunks of a minibatch are reduced by separate
    workgroups (tsm_ppo_adv_stats_wide, work = f64 scratch kept alive by the caller when captured in a graph)."""
    adv = _chk(adv, torch.float32, "adv").reshape(-1)
    mb_start = _chk(mb_start, torch.int64, "mb_start")
    n_mb = mb_start.numel() - 1
    stats = out if out is not None else torch.empty(n_mb, 2, dtype=torch.float32, device=adv.device)
    if max_rows > 8192:
        if work is None:
            work = torch.empty(call("tsm_ppo_adv_stats_work_elems", n_mb, max_rows), dtype=torch.float64, device=adv.device)
        call("tsm_ppo_adv_stats_wide", ptr(adv), ptr(perm), ptr(mb_start), n_mb, int(max_rows), ptr(work), ptr(stats),
             stream_ptr())
        return stats
    call("tsm_ppo_adv_stats", ptr(adv), ptr(perm), ptr(mb_start), n_mb, ptr(stats), stream_ptr())
    return stats


def ppo_adv_stats_pack(stats, mb_start, out=None):
    """This rank's per-minibatch (mean, unbiased std) + row counts -> f64 [n_mb, 3] = (n, sum x, sum x^2): the additive
    form one all-reduce sums over data-parallel ranks (parallel.GradSync.merge_adv_stats_)."""
    stats = _chk(stats, torch.float32, "stats").reshape(-1, 2)
    mb_start = _chk(mb_start, torch.int64, "mb_start")
    n_mb = mb_start.numel() - 1
    if stats.shape[0] != n_mb:
        raise ValueError(f"ppo_adv_stats_pack: {stats.shape[0]} statistics rows for {n_mb} minibatches")
    pack = out if out is not None else torch.empty(n_mb, 3, dtype=torch.float64, device=stats.device)
    call("tsm_ppo_adv_stats_pack", ptr(stats), ptr(mb_start), n_mb, ptr(_chk(pack, torch.float64, "pack")), stream_ptr())
    return pack


def ppo_adv_stats_unpack(pack, stats):
    """Summed (n, sum x, sum x^2) -> (mean, unbiased std) of the union, written into `stats` [n_mb, 2] f32 in place."""
    pack = _chk(pack, torch.float64, "pack").reshape(-1, 3)
    st = _chk(stats, torch.float32, "stats")
    if st.numel() != 2 * pack.shape[0]:
        raise ValueError("ppo_adv_stats_unpack: stats and pack disagree on the number of minibatches")
    call("tsm_ppo_adv_stats_unpack", ptr(pack), pack.shape[0], ptr(st), stream_ptr())
    return stats


def ppo_loss_fwd_bwd(logits, value, act, logp_old, adv, returns, cfg: tsm_ppo_cfg, adv_stats=None,
                     v_s_old=None, perm=None, first_row=0, finalize: bool = True):
    """One minibatch -> (dlogits[M,A], dvalue[M], scalars[4]={loss, clip, vf, ent}).
    finalize=False leaves the per-workgroup partial sums unfolded (returned in place of `scalars`): an epoch's
    statistics can then be folded once, as the fused path does (ppo_finalize_many)."""
    logits = _chk(logits, torch.float32, "logits")
    M, A = logits.shape
    value = _chk(value, torch.float32, "value").reshape(-1)
    dev = logits.device
    vg = max(1, int(cfg.value_group))
    if value.numel() * vg != M:
        raise ValueError(f"ppo_loss_fwd_bwd: {value.numel()} values for {M} samples with value_group={vg}")
    dlogits = torch.empty_like(logits)
    dvalue = torch.empty(M // vg, dtype=torch.float32, device=dev)
    partial = torch.empty(max(1, call("tsm_ppo_loss_partial_elems", M)), dtype=torch.float64, device=dev)
    scalars = torch.empty(4, dtype=torch.float32, device=dev)
    s = stream_ptr()
    call("tsm_ppo_loss_fwd_bwd", ptr(logits), ptr(value), ptr(_chk(act, torch.int32, "act")),
         ptr(_chk(logp_old, torch.float32, "logp_old")), ptr(_chk(adv, torch.float32, "adv")),
         ptr(_chk(returns, torch.float32, "returns")), ptr(v_s_old), ptr(perm), first_row, M, A,
         ptr(adv_stats), C.byref(cfg), ptr(dlogits), ptr(dvalue), ptr(partial), s)
    if not finalize:
        return dlogits, dvalue, partial
    if M > 0:
        call("tsm_ppo_loss_finalize", ptr(partial), M, C.byref(cfg), ptr(scalars), s)
    return dlogits, dvalue, scalars


# --------------------------------------------------------------------------------------------
# optimizer (algorithm_base.py:485-498; optim.py:91-111)
# --------------------------------------------------------------------------------------------
def adam_step(param, grad_slabs, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
              weight_decay=0.0, max_grad_norm=None, work=None, step_dev=None, image=None, image_map=None, lr_dev=None):
    """In-place Adam on a flat f32 vector; grad_slabs [n_slab, n] are summed in slab order."""
    n = param.numel()
    grad_slabs = _chk(grad_slabs, torch.float32, "grad_slabs").reshape(-1, n)
    if max_grad_norm and work is None:
        work = torch.empty(call("tsm_adam_work_elems", n), dtype=torch.float32, device=param.device)
    call("tsm_adam_step", ptr(param), ptr(grad_slabs), grad_slabs.shape[0], n, ptr(exp_avg), ptr(exp_avg_sq),
         int(step), ptr(step_dev), float(lr), ptr(lr_dev), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
         float(max_grad_norm or 0.0), ptr(work), ptr(image), ptr(image_map), stream_ptr())
    return param


def _slab_segs(segs, n: int):
    """[(slabs [n_slab, stride >= n_k], offset, n_k[, scale_dev[, frag_image[, col0]]]), ...] -> ctypes array of tsm_slab_seg.
    col0: the segment's gradients are columns [col0, col0 + n_k) of every slab row (a view into joint slabs).
    A segment's slabs tensor may be wider than its parameter count (a view into joint slabs): stride = its row pitch.
    scale_dev (optional device f32[1]): the segment's summed gradient is multiplied by it.
    frag_image (optional, `critic_w1_image` layout): the segment is a [128][n_k / 128] first-layer weight matrix whose
    updated values the optimizer also stores in fragment order."""
    arr = (_abi.tsm_slab_seg * len(segs))()
    for k, seg in enumerate(segs):
        sl, off, nk = seg[:3]
        sc = seg[3] if len(seg) > 3 else None
        img = seg[4] if len(seg) > 4 else None
        col0 = int(seg[5]) if len(seg) > 5 else 0   # the segment's gradients start at this column of every slab row
        sl = _chk(sl, torch.float32, "slabs")
        if sl.dim() != 2 or sl.shape[1] < col0 + nk:
            raise ValueError("slab segment: expected slabs [n_slab, >= col0 + n] for n = %d, got %s" % (nk, tuple(sl.shape)))
        k1 = kj = 0
        if img is not None:
            if nk % 128:
                raise ValueError("slab segment: a fragment image belongs to a [128][K1] weight matrix")
            k1 = nk // 128
            kj = call("tsm_critic_rows_w1_image_kj", k1)
            if _chk(img, torch.float32, "frag_image").numel() != call("tsm_critic_rows_w1_image_elems", k1):
                raise ValueError("slab segment: frag_image has the wrong size for K1 = %d" % k1)
        arr[k] = _abi.tsm_slab_seg(ptr(sl) + 4 * col0, int(off), int(nk), int(sl.shape[1]), int(sl.shape[0]), k1,
                                   ptr(None if sc is None else _chk(sc, torch.float32, "scale_dev")), ptr(img), kj, 0)
    return arr


def critic_w1_image(w0_flat, in_dim: int, out=None):
    """The first-layer weights w0 [128][in_dim] (the head of a critic's flat parameter vector) in the fragment order the
    one-launch critic gradient step loads with coalesced 16-B loads (include/tsmarl.h: tsm_critic_rows_w1_image)."""
    n = call("tsm_critic_rows_w1_image_elems", in_dim)
    if n < 0:
        raise ValueError(f"critic_w1_image: in_dim = {in_dim} is not served by the rows kernels")
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=w0_flat.device)
    call("tsm_critic_rows_w1_image", ptr(_chk(w0_flat, torch.float32, "w0")), in_dim, ptr(_chk(out, torch.float32, "out")),
         stream_ptr())
    return out


def adam_step_segs(param, segs, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                   max_grad_norm=None, work=None, step_dev=None, lr_dev=None):
    """`adam_step` for a flat vector whose parts have their own slab arrays: segs = [(slabs, offset, n), ...] tiling
    [0, param.numel()) in order.  One launch (+ one reduction launch when clipping; ONE norm over all segments)."""
    n = param.numel()
    if max_grad_norm and work is None:
        work = torch.empty(call("tsm_adam_work_elems", n), dtype=torch.float32, device=param.device)
    arr = _slab_segs(segs, n)
    call("tsm_adam_step_segs", ptr(param), arr, len(segs), n, ptr(exp_avg), ptr(exp_avg_sq), int(step), ptr(step_dev),
         float(lr), ptr(lr_dev), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
         float(max_grad_norm or 0.0), ptr(work), stream_ptr())
    return param


def reduce_slabs_segs(segs, n: int, out=None, scale: float = 1.0):
    """Flat gradient [n] = scale * per-segment slab sums (see adam_step_segs); one launch."""
    dev = segs[0][0].device
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=dev)
    call("tsm_reduce_slabs_segs", _slab_segs(segs, n), len(segs), n, float(scale), ptr(out), stream_ptr())
    return out


def reduce_slabs(grad_slabs, out=None, scale: float = 1.0):
    """Flat gradient = scale * sum of per-workgroup slabs [n_slab, n] (deterministic slab order)."""
    grad_slabs = _chk(grad_slabs, torch.float32, "grad_slabs")
    n_slab, n = grad_slabs.shape
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=grad_slabs.device)
    call("tsm_reduce_slabs", ptr(grad_slabs), n_slab, n, float(scale), ptr(out), stream_ptr())
    return out


# --------------------------------------------------------------------------------------------
# fused actor+critic MLP (reinforce.py:167-192, a2c.py:121-127, ppo.py:157-212)
# --------------------------------------------------------------------------------------------
POLICY_MODES = {"none": 0, "sample": 1, "mode": 2, "given": 3}


def policy_param_count(obs_dim: int, hidden: int, n_act: int) -> int:
    n = call("tsm_policy_param_count", obs_dim, hidden, n_act)
    if n < 0:
        raise ValueError(f"fused MLP does not support hidden={hidden}")
    return n


def policy_forward(params, obs, n_act: int, hidden: int = 64, mode: str = "none", seed: int = 0, offset: int = 0,
                   act=None, want_logits=True, want_value=True, want_logp=True, offset_dev=None, out=None,
                   image=None):
    """obs [B, D] f32 -> dict(logits[B,A], value[B], act[B] i32, logp[B]) via one fused kernel."""
    obs = _chk(obs, torch.float32, "obs")
    B, D = obs.shape
    dev = obs.device
    m = POLICY_MODES[mode]
    if out is not None:  # preallocated outputs (graph capture): dict(logits, value, act, logp), entries may be None
        logits, value, logp = out.get("logits"), out.get("value"), out.get("logp")
        act = out.get("act") if m != 3 else _chk(act, torch.int32, "act")
    else:
        logits = torch.empty(B, n_act, dtype=torch.float32, device=dev) if want_logits else None
        value = torch.empty(B, dtype=torch.float32, device=dev) if want_value else None
        if m == 3:
            act = _chk(act, torch.int32, "act")
        elif m != 0:
            act = torch.empty(B, dtype=torch.int32, device=dev)
        logp = torch.empty(B, dtype=torch.float32, device=dev) if (want_logp and m != 0) else None
    call("tsm_policy_forward", ptr(_chk(params, torch.float32, "params")), ptr(image), D, hidden, n_act, ptr(obs), B, m,
         seed & (2**64 - 1), offset & (2**64 - 1), ptr(offset_dev), ptr(logits), ptr(value), ptr(act), ptr(logp),
         stream_ptr())
    return dict(logits=logits, value=value, act=act, logp=logp)


def policy_image(obs_dim: int, hidden: int, n_act: int, device):
    """(zero image f32[elems], map i32[P]) of the padded LDS-layout parameter copy."""
    import numpy as np

    n_img = call("tsm_policy_image_elems", obs_dim, hidden, n_act)
    if n_img < 0:  # layout too small for the padded staging copy: kernels re-pack `params` themselves
        return None, None
    n_par = policy_param_count(obs_dim, hidden, n_act)
    m = np.zeros(n_par, np.int32)
    call("tsm_policy_image_map", obs_dim, hidden, n_act, m.ctypes.data_as(C.c_void_p))
    return torch.zeros(n_img, dtype=torch.float32, device=device), torch.from_numpy(m).to(device)


def scatter_image(params, image, image_map):
    if image is None:
        return None
    call("tsm_scatter_image", ptr(params), params.numel(), ptr(image_map), ptr(image), stream_ptr())
    return image


def ppo_finalize_many(partial, stride_elems: int, n_blocks_dev, M_dev, cfg: tsm_ppo_cfg, scalars_out):
    """scalars_out[k] = {loss, clip, vf, ent} of gradient step k from its loss partials (one launch for all k).
    scalars_out may be a pinned host tensor (mapped memory: the kernel writes it directly, no D2H copy)."""
    if scalars_out.is_cuda:
        out_p = ptr(scalars_out)
    elif scalars_out.is_pinned() and scalars_out.is_contiguous() and scalars_out.dtype == torch.float32:
        out_p = scalars_out.data_ptr()
    else:
        raise RuntimeError("ppo_finalize_many: scalars_out must be a device tensor or pinned host memory (f32)")
    call("tsm_ppo_finalize_many", ptr(partial), stride_elems, ptr(_chk(n_blocks_dev, torch.int32, "n_blocks")),
         ptr(_chk(M_dev, torch.int64, "M")), scalars_out.shape[0], C.byref(cfg), out_p, stream_ptr())
    return scalars_out


def ppo_update_grid(M: int, max_blocks: int = 0) -> int:
    if not max_blocks:
        max_blocks = int(os.environ.get("TSM_UPDATE_MAX_BLOCKS", "0"))  # (A/B timing of the slab count; default: the rule)
    return call("tsm_ppo_update_grid", M, max_blocks)


def ppo_update_fused(params, obs, act, logp_old, adv, returns, cfg: tsm_ppo_cfg, n_act: int, hidden: int = 64,
                     adv_stats=None, v_s_old=None, perm=None, first_row=0, M=None, n_blocks=None,
                     slabs=None, partial=None, scalars=None, opt_step_dev=None, image=None, want_scalars=True):
    """One PPO gradient step up to the gradients -> (grad_slabs[n_blocks, P], scalars[4])."""
    obs = _chk(obs, torch.float32, "obs")
    D = obs.shape[-1]
    if M is None:
        M = perm.numel() if perm is not None else obs.shape[0] - first_row
    if n_blocks is None:
        n_blocks = ppo_update_grid(M)
    P = params.numel()
    dev = obs.device
    if slabs is None:
        slabs = torch.empty(n_blocks, P, dtype=torch.float32, device=dev)
    elif slabs.numel() < n_blocks * P:  # the kernel writes n_blocks slabs of P floats: never launch into a smaller buffer
        raise ValueError(f"ppo_update_fused: slabs holds {slabs.numel()} floats, {n_blocks} slabs of {P} need {n_blocks * P}")
    if partial is None:
        partial = torch.empty(n_blocks * 4, dtype=torch.float64, device=dev)
    elif partial.numel() < n_blocks * 4:
        raise ValueError(f"ppo_update_fused: partial holds {partial.numel()} values, {n_blocks} workgroups need {n_blocks * 4}")
    if perm is not None and perm.numel() < M:
        raise ValueError(f"ppo_update_fused: perm holds {perm.numel()} row ids, M = {M}")
    if scalars is None and want_scalars:
        scalars = torch.empty(4, dtype=torch.float32, device=dev)
    call("tsm_ppo_update_fused", ptr(_chk(params, torch.float32, "params")), ptr(image), D, hidden, n_act, ptr(obs),
         ptr(_chk(act, torch.int32, "act")), ptr(_chk(logp_old, torch.float32, "logp_old")),
         ptr(_chk(adv, torch.float32, "adv")), ptr(_chk(returns, torch.float32, "returns")), ptr(v_s_old),
         ptr(perm), first_row, M, ptr(adv_stats), C.byref(cfg), n_blocks, ptr(slabs), ptr(partial), ptr(scalars),
         ptr(opt_step_dev), stream_ptr())
    return slabs, scalars


def ppo_actor_rows_supported(obs_dim: int, hidden_sizes, n_act: int, act: str = "relu") -> bool:
    """Does the one-launch actor step (csrc/ppo_rows.hip) cover this actor?  obs -> 128 -> 128 -> n_act, ReLU."""
    hs = list(hidden_sizes)
    ok = act == "relu" and len(hs) == 2 and hs[0] == hs[1] and bool(call("tsm_ppo_actor_rows_supported", obs_dim, hs[0], n_act))
    if ok:
        _ppo_rows_init()
    return ok


_ppo_rows_ready = False


def _ppo_rows_init() -> None:
    """The rows kernels' LDS attributes, set when a host first asks whether they serve its nets -- at construction time,
    outside any stream capture (tsm_ppo_rows_init)."""
    global _ppo_rows_ready
    if not _ppo_rows_ready and torch.cuda.is_available():
        call("tsm_ppo_rows_init")
        _ppo_rows_ready = True


def ppo_actor_rows_grid(M: int) -> int:
    return call("tsm_ppo_actor_rows_grid", M)


def ppo_actor_rows_update(actor_params, obs, act, logp_old, adv, cfg: tsm_ppo_cfg, n_act: int, hidden: int = 128,
                          adv_stats=None, perm=None, first_row=0, M=None, n_blocks=None, slabs=None, partial=None,
                          opt_step_dev=None):
    """Actor half of one PPO gradient step in one launch -> (grad_slabs [n_blocks, P_actor], loss partials f64
    [n_blocks, 4] = {sum clip objective, 0, sum entropy, 0}).  opt_step_dev (device i64[1]): advanced by one."""
    obs = _chk(obs, torch.float32, "obs")
    D = obs.shape[-1]
    if M is None:
        M = perm.numel() if perm is not None else obs.shape[0] - first_row
    if n_blocks is None:
        n_blocks = ppo_actor_rows_grid(M)
    P = actor_params.numel()
    if P != call("tsm_ppo_actor_rows_param_count", D, hidden, n_act):
        raise ValueError(f"ppo_actor_rows_update: {P} actor parameters do not match obs {D} -> {hidden} -> {hidden} -> {n_act}")
    dev = obs.device
    if slabs is None:
        slabs = torch.empty(n_blocks, P, dtype=torch.float32, device=dev)
    elif slabs.numel() < n_blocks * P:
        raise ValueError(f"ppo_actor_rows_update: slabs holds {slabs.numel()} floats, {n_blocks} slabs of {P} need {n_blocks * P}")
    if partial is None:
        partial = torch.empty(n_blocks * 4, dtype=torch.float64, device=dev)
    elif partial.numel() < n_blocks * 4:
        raise ValueError("ppo_actor_rows_update: partial is too small")
    if perm is not None and perm.numel() < M:
        raise ValueError(f"ppo_actor_rows_update: perm holds {perm.numel()} sample ids, M = {M}")
    call("tsm_ppo_actor_rows_update", ptr(_chk(actor_params, torch.float32, "actor_params")), D, hidden, n_act, ptr(obs),
         ptr(_chk(act, torch.int32, "act")), ptr(None if logp_old is None else _chk(logp_old, torch.float32, "logp_old")),
         ptr(None if adv is None else _chk(adv, torch.float32, "adv")), ptr(perm), first_row, M, ptr(adv_stats), C.byref(cfg),
         n_blocks, ptr(slabs), ptr(partial), ptr(opt_step_dev), stream_ptr())
    return slabs, partial


def ppo_critic_rows_supported(in_dim: int, hidden_sizes, n_agent: int, act: str = "relu") -> bool:
    """Does the one-launch critic step (csrc/ppo_rows.hip) cover this critic?  in_dim -> 128 -> 128 -> 1, ReLU."""
    hs = list(hidden_sizes)
    n_slice = -(-in_dim // 32)
    ok = (act == "relu" and len(hs) == 2 and hs[0] == hs[1] and n_slice in (1, 2, 3)
          and bool(call("tsm_ppo_critic_rows_supported", in_dim, hs[0], n_agent)))
    if ok:
        _ppo_rows_init()
    return ok


def ppo_critic_rows_grid(Mr: int) -> int:
    return call("tsm_ppo_critic_rows_grid", Mr)


def ppo_critic_rows_update(critic_params, obs_rows, returns, cfg: tsm_ppo_cfg, n_agent: int, hidden: int = 128, v_s_old=None,
                           rows=None, first_row=0, Mr=None, n_blocks=None, slabs=None, partial=None):
    """Critic half of one PPO gradient step in one launch -> (grad_slabs [n_blocks, P_critic], loss partials f64
    [n_blocks, 4] = {0, sum vf, 0, 0}).  obs_rows [n, in_dim]: the joint rows (in_dim = n_agent * obs_dim)."""
    obs_rows = _chk(obs_rows, torch.float32, "obs_rows")
    K1 = obs_rows.shape[-1]
    if Mr is None:
        Mr = rows.numel() if rows is not None else obs_rows.shape[0] - first_row
    if n_blocks is None:
        n_blocks = ppo_critic_rows_grid(Mr)
    P = critic_params.numel()
    if P != call("tsm_ppo_critic_rows_param_count", K1, hidden):
        raise ValueError(f"ppo_critic_rows_update: {P} critic parameters do not match {K1} -> {hidden} -> {hidden} -> 1")
    dev = obs_rows.device
    if slabs is None:
        slabs = torch.empty(n_blocks, P, dtype=torch.float32, device=dev)
    elif slabs.numel() < n_blocks * P:
        raise ValueError("ppo_critic_rows_update: slabs is too small")
    if partial is None:
        partial = torch.empty(n_blocks * 4, dtype=torch.float64, device=dev)
    elif partial.numel() < n_blocks * 4:
        raise ValueError("ppo_critic_rows_update: partial is too small")
    if rows is not None and rows.numel() < Mr:
        raise ValueError("ppo_critic_rows_update: rows holds fewer ids than Mr")
    _critic_rows_init(K1, hidden)
    call("tsm_ppo_critic_rows_update", ptr(_chk(critic_params, torch.float32, "critic_params")), K1, hidden, n_agent,
         ptr(obs_rows), ptr(_chk(returns, torch.float32, "returns")), ptr(v_s_old), ptr(rows), first_row, Mr, C.byref(cfg),
         n_blocks, ptr(slabs), ptr(partial), stream_ptr())
    return slabs, partial


def critic_rows_forward_supported(in_dim: int, hidden_sizes, n_out: int = 1, act: str = "relu") -> bool:
    """Does the one-launch critic forward (csrc/critic_rows.hip) cover this critic?  in -> 128 -> 128 -> n_out <= 16, ReLU
    (n_out > 1: the value is the mean of the outputs)."""
    hs = list(hidden_sizes)
    return (act == "relu" and 1 <= n_out <= 16 and len(hs) == 2 and hs[0] == hs[1]
            and bool(call("tsm_critic_rows_forward_supported", in_dim, hs[0])))


_critic_rows_ready: set = set()


def _critic_rows_init(K1: int, hidden: int) -> None:
    """One-time function attributes (dynamic LDS size) of the one-launch critic kernels serving this input width; must
    happen outside any stream capture."""
    if (K1, hidden) in _critic_rows_ready:
        return
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("one-launch critic kernels: first use inside a stream capture; run them (or "
                           "ops.call('tsm_critic_rows_init', in_dim, hidden)) once before capturing")
    call("tsm_critic_rows_init", K1, hidden)
    _critic_rows_ready.add((K1, hidden))


def critic_rows_forward(critic_params, obs_rows, hidden: int = 128, rows=None, first_row: int = 0, Mr=None, run_if=None,
                        out=None, n_out: int = 1):
    """V(row) for Mr rows of obs_rows [n, in_dim] (row ids `rows`, or first_row + i) in ONE launch -> values [Mr]
    (n_out > 1: the mean of the critic's outputs).  run_if (device i32[1]): the launch is a no-op (and `out` is left
    as it is) when it holds 0."""
    obs_rows = _chk(obs_rows, torch.float32, "obs_rows")
    K1 = obs_rows.shape[-1]
    if Mr is None:
        Mr = rows.numel() if rows is not None else obs_rows.shape[0] - first_row
    if rows is not None and rows.numel() < Mr:
        raise ValueError(f"critic_rows_forward: rows holds {rows.numel()} ids, Mr = {Mr}")
    if rows is None and first_row + Mr > obs_rows.shape[0]:
        raise ValueError(f"critic_rows_forward: rows [{first_row}, {first_row + Mr}) exceed the {obs_rows.shape[0]} given")
    if critic_params.numel() != hidden * K1 + hidden + hidden * hidden + hidden + n_out * hidden + n_out:
        raise ValueError(f"critic_rows_forward: {critic_params.numel()} parameters do not match {K1} -> {hidden} -> {hidden} -> {n_out}")
    _critic_rows_init(K1, hidden)
    if out is None:
        out = torch.empty(Mr, dtype=torch.float32, device=obs_rows.device)
    elif out.numel() < Mr:
        raise ValueError("critic_rows_forward: out is too small")
    call("tsm_critic_rows_forward", ptr(_chk(critic_params, torch.float32, "critic_params")), K1, hidden, n_out, ptr(obs_rows),
         ptr(None if rows is None else _chk(rows, torch.int64, "rows")), first_row, Mr,
         ptr(None if run_if is None else _chk(run_if, torch.int32, "run_if")), ptr(out), stream_ptr())
    return out


def critic_rows_grad_supported(in_dim: int, hidden_sizes, n_out: int = 1, act: str = "relu") -> bool:
    """Does the two-launch critic gradient step (csrc/critic_train.hip + critic_dw1.hip) cover this critic?"""
    return critic_rows_forward_supported(in_dim, hidden_sizes, n_out, act) and (in_dim % 4 == 0 or in_dim <= 64)


def _critic_grad_ws(K1: int, hidden: int, n_out: int, Mr: int, td: bool, dev, ws: dict | None, split_dw2: bool = False):
    """(n_blocks, n_chunks, dh1, rest slabs, w1 slabs, partial) for a gradient step of Mr rows; cached in `ws` (the
    buffers are what captured graphs hold on to).  split_dw2: + the published H1 / dH2 and the dW2 chunk slabs; the rest slabs
    start as zeros (their W2 columns are never written in that mode)."""
    nb = call("tsm_critic_rows_grad_grid", Mr, int(td))
    nc = call("tsm_critic_rows_dw1_chunks", Mr, K1)
    key = ("critic_grad", K1, hidden, n_out, Mr, td, bool(split_dw2))
    w = None if ws is None else ws.get(key)
    if w is None:
        n_rest = hidden + hidden * hidden + hidden + n_out * hidden + n_out
        w = dict(nb=nb, nc=nc, dh1=torch.empty(Mr, hidden, dtype=torch.float32, device=dev),
                 rest=(torch.zeros if split_dw2 else torch.empty)(nb, n_rest, dtype=torch.float32, device=dev),
                 w1=torch.empty(nc, hidden * K1, dtype=torch.float32, device=dev),
                 partial=torch.zeros(nb * 4, dtype=torch.float64, device=dev))
        if split_dw2:
            w.update(h1=torch.empty(Mr, hidden, dtype=torch.float32, device=dev),
                     dh2=torch.empty(Mr, hidden, dtype=torch.float32, device=dev),
                     w2=torch.empty(nc, hidden * hidden, dtype=torch.float32, device=dev))
        if ws is not None:
            ws[key] = w
    return w


def _side_reductions(side):
    """[(slabs [n_slab, stride >= n], out [n]), ...] (at most two) -> (ctypes array or None, count): slab sets summed to one
    row each by extra workgroups of the dW1 launch (include/tsmarl.h: tsm_slab_reduce)."""
    if not side:
        return None, 0
    if len(side) > 2:
        raise ValueError("at most two side reductions ride on the dW1 launch")
    arr = (_abi.tsm_slab_reduce * len(side))()
    for k, (sl, out) in enumerate(side):
        sl, out = _chk(sl, torch.float32, "side slabs"), _chk(out, torch.float32, "side out")
        if sl.dim() != 2 or out.numel() > sl.shape[1]:
            raise ValueError("side reduction: slabs [n_slab, >= n] and out [n] expected")
        arr[k] = _abi.tsm_slab_reduce(ptr(sl), out.numel(), int(sl.shape[1]), int(sl.shape[0]), 0, ptr(out))
    return arr, len(side)


def critic_rows_grad_ppo(critic_params, obs_rows, returns, cfg: tsm_ppo_cfg, n_agent: int, hidden: int = 128, v_s_old=None,
                         rows=None, first_row=0, Mr=None, partial=None, ws: dict | None = None, w1_image=None, side_reduce=None,
                         split_dw2: bool = False):
    """Critic half of one PPO gradient step on joint rows in two launches -> (w1_slabs [n_chunks, H * in_dim],
    rest_slabs [n_blocks, P - H * in_dim], partial f64 [n_blocks * 4] = {0, sum vf, 0, 0} per workgroup): feed the two slab
    arrays to `adam_step_segs` as segments (W1 first).  `partial`: where to leave the loss partials (>= n_blocks * 4).
    w1_image: the first-layer weights in fragment order (`critic_w1_image`; must hold the values of critic_params' w0).
    side_reduce: [(slabs, out), ...] -- other kernels' complete slab sets summed to one row each inside the dW1 launch; an
    entry whose slabs is the string "rest" names this step's own rest slabs.
    split_dw2: the first launch publishes H1 / dH2 instead of forming dW2 (a rank-32 update per tile as a 64 KB slab) and the
    second forms dW2 = dH2^T H1 over its row chunks beside dW1: the W2 gradient is then `ws[...]["w2"]` [n_chunks, H * H] and the
    W2 columns of `rest` stay zero -- `critic_grad_segs` lists the optimizer's segments for either mode."""
    obs_rows = _chk(obs_rows, torch.float32, "obs_rows")
    K1 = obs_rows.shape[-1]
    if Mr is None:
        Mr = rows.numel() if rows is not None else obs_rows.shape[0] - first_row
    if critic_params.numel() != call("tsm_critic_rows_param_count", K1, hidden, 1):
        raise ValueError(f"critic_rows_grad_ppo: {critic_params.numel()} parameters do not match {K1} -> {hidden} -> {hidden} -> 1")
    if rows is not None and rows.numel() < Mr:
        raise ValueError("critic_rows_grad_ppo: rows holds fewer ids than Mr")
    _critic_rows_init(K1, hidden)
    w = _critic_grad_ws(K1, hidden, 1, Mr, False, obs_rows.device, ws, split_dw2)
    part = w["partial"] if partial is None else partial
    if part.numel() < w["nb"] * 4:
        raise ValueError("critic_rows_grad_ppo: partial is too small")
    call("tsm_critic_rows_grad_ppo", ptr(_chk(critic_params, torch.float32, "critic_params")), ptr(w1_image), K1, hidden, n_agent,
         ptr(obs_rows), ptr(_chk(returns, torch.float32, "returns")), ptr(v_s_old), ptr(rows), first_row, Mr, C.byref(cfg),
         w["nb"], ptr(w["dh1"]), ptr(w.get("h1")), ptr(w.get("dh2")), ptr(w["rest"]), ptr(part), stream_ptr())
    side = [(w["rest"] if isinstance(sl, str) else sl, out) for sl, out in (side_reduce or [])]
    arr, n_side = _side_reductions(side)
    call("tsm_critic_rows_dw1", ptr(w["dh1"]), ptr(obs_rows), K1, ptr(rows), first_row, 0, 0, Mr, w["nc"], ptr(w["w1"]),
         ptr(w.get("dh2")), ptr(w.get("h1")), ptr(w.get("w2")), arr, n_side, stream_ptr())
    return w["w1"], w["rest"], part


def critic_grad_segs(w: dict, offset: int, K1: int, hidden: int = 128, n_out: int = 1, w1_image=None, rest_row=None) -> list:
    """The optimizer's slab segments (`adam_step_segs` / `reduce_slabs_segs`) for the critic gradients a `critic_rows_grad_*`
    call left in workspace `w`, the critic's parameters starting at `offset` of the flat vector.  Plain mode: W1 chunk slabs |
    rest slabs (or `rest_row`, their side-reduced sum [1, n_rest]).  split_dw2 mode: W1 chunks | b1 (a column view of the rest
    slabs) | W2 chunks | b2, W3, b3 (another view)."""
    nW1, H = hidden * K1, hidden
    if "w2" not in w:
        rest = w["rest"] if rest_row is None else rest_row
        return [(w["w1"], offset, nW1, None, w1_image), (rest, offset + nW1, w["rest"].shape[1])]
    tail = H + n_out * H + n_out
    return [(w["w1"], offset, nW1, None, w1_image), (w["rest"], offset + nW1, H),
            (w["w2"], offset + nW1 + H, H * H), (w["rest"], offset + nW1 + H + H * H, tail, None, None, H + H * H)]


def critic_rows_grad_td(critic_params, joint_store, T: int, E: int, rew, terminated, agent: int, n_agent: int, v_last,
                        gamma: float, n_out: int, hidden: int = 128, partial=None, ws: dict | None = None, v_next_full=None,
                        use_full=None):
    """The critic half of CTDEPolicy.learn (ctde.py:149-172, 188-190) on CHAINED rows in two launches.
    joint_store f32 [T, E, in_dim] (time-major joint rows), rew f32 / terminated u8 [T, E, n_agent] (agent column `agent`),
    v_last f32 [E].  use_full (device i32[1]) != 0: targets take v_next_full [T * E] (env-major V(obs_next)) instead of the
    next row's value.  -> (w1_slabs, rest_slabs, partial f64 = {sum adv, sum sq, 0, 0} per workgroup)."""
    joint_store = _chk(joint_store, torch.float32, "joint_store")
    K1 = joint_store.shape[-1]
    B = T * E
    if joint_store.numel() < B * K1 or rew.numel() < B * n_agent or terminated.numel() < B * n_agent or v_last.numel() < E:
        raise ValueError("critic_rows_grad_td: store tensors are smaller than T x E")
    if critic_params.numel() != call("tsm_critic_rows_param_count", K1, hidden, n_out):
        raise ValueError(f"critic_rows_grad_td: {critic_params.numel()} parameters do not match {K1} -> {hidden} -> {hidden} -> {n_out}")
    _critic_rows_init(K1, hidden)
    w = _critic_grad_ws(K1, hidden, n_out, B, True, joint_store.device, ws)
    part = w["partial"] if partial is None else partial
    term = terminated.view(torch.uint8) if terminated.dtype == torch.bool else _chk(terminated, torch.uint8, "terminated")
    call("tsm_critic_rows_grad_td", ptr(_chk(critic_params, torch.float32, "critic_params")), None, K1, hidden, n_out,
         ptr(joint_store), T, E, ptr(_chk(rew, torch.float32, "rew")), ptr(term), n_agent, agent,
         ptr(_chk(v_last, torch.float32, "v_last")), ptr(None if v_next_full is None else _chk(v_next_full, torch.float32, "v_next_full")),
         ptr(None if use_full is None else _chk(use_full, torch.int32, "use_full")), float(gamma), w["nb"], ptr(w["dh1"]),
         ptr(w["rest"]), ptr(part), stream_ptr())
    call("tsm_critic_rows_dw1", ptr(w["dh1"]), ptr(joint_store), K1, None, 0, T, E, B, w["nc"], ptr(w["w1"]), None, None, None,
         None, 0, stream_ptr())
    return w["w1"], w["rest"], part


def ctde_finalize(critic_partial, nb_c: int, actor_partial, nb_a: int, B: int, scalars_out, mean_adv_out):
    """{actor_loss, critic_loss} and mean(advantage) of CTDEPolicy.learn from the kernels' partial sums (one launch).
    scalars_out: device f32[2] or pinned host f32[2]; mean_adv_out: device f32[1]."""
    if scalars_out.is_cuda:
        out_p = ptr(scalars_out)
    elif scalars_out.is_pinned() and scalars_out.is_contiguous() and scalars_out.dtype == torch.float32:
        out_p = scalars_out.data_ptr()
    else:
        raise RuntimeError("ctde_finalize: scalars_out must be a device tensor or pinned host memory (f32)")
    call("tsm_ctde_finalize", ptr(_chk(critic_partial, torch.float64, "critic_partial")), nb_c,
         ptr(_chk(actor_partial, torch.float64, "actor_partial")), nb_a, B, out_p,
         ptr(_chk(mean_adv_out, torch.float32, "mean_adv_out")), stream_ptr())
    return scalars_out


def ppo_value_loss(value, returns, cfg: tsm_ppo_cfg, M: int, v_s_old=None, perm=None, first_row=0, partial=None):
    """Value term of the PPO loss alone (loss_kind = 2) for M samples whose values are `value` (one per sample, or one
    per joint row with cfg.value_group = N) -> (dvalue [len(value)], loss partials f64 [blocks, 4] = {0, sum vf, 0, 0})."""
    value = _chk(value, torch.float32, "value").reshape(-1)
    vg = max(1, int(cfg.value_group))
    if value.numel() * vg != M:
        raise ValueError(f"ppo_value_loss: {value.numel()} values for {M} samples with value_group={vg}")
    if cfg.loss_kind != 2:
        raise ValueError("ppo_value_loss needs a cfg with loss_kind = 2")
    dev = value.device
    dvalue = torch.empty(value.numel(), dtype=torch.float32, device=dev)
    n_part = call("tsm_ppo_loss_partial_elems", M)
    if partial is None:
        partial = torch.empty(n_part, dtype=torch.float64, device=dev)
    elif partial.numel() < n_part:
        raise ValueError("ppo_value_loss: partial is too small")
    call("tsm_ppo_loss_fwd_bwd", None, ptr(value), None, None, None, ptr(_chk(returns, torch.float32, "returns")),
         ptr(v_s_old), ptr(perm), first_row, M, 1, None, C.byref(cfg), None, ptr(dvalue), ptr(partial), stream_ptr())
    return dvalue, partial


def ppo_loss_partial_elems(M: int) -> int:
    return call("tsm_ppo_loss_partial_elems", M)


# --------------------------------------------------------------------------------------------
# CTDE (ctde.py:291-300)
# --------------------------------------------------------------------------------------------
def global_state(obs_by_agent, mode: str = "concatenate") -> torch.Tensor:
    """obs_by_agent: list of [B, D] f32 tensors in env.agents order."""
    if mode not in ("concatenate", "mean"):
        raise ValueError(f"unsupported global-state mode {mode!r} (concatenate | mean)")
    obs = [_chk(o, torch.float32, "obs") for o in obs_by_agent]
    N = len(obs)
    B, D = obs[0].shape
    arr = (C.c_void_p * N)(*[ptr(o) for o in obs])
    out = torch.empty((B, N * D) if mode == "concatenate" else (B, D), dtype=torch.float32, device=obs[0].device)
    call("tsm_global_state", arr, N, B, D, 0 if mode == "concatenate" else 1, ptr(out), stream_ptr())
    return out


def random_permutations(n: int, n_perm: int, seed: int, counter: int = 0, counter_dev=None, scale: int = 1,
                        group_size: int = 1, offset_mul: int = 0, out=None, device="cuda", advance: int = 0, done_ctr=None):
    """n_perm pseudo-random permutations of range(n) in one launch -> i64 [n_perm, n] (batch.py:1219 on device).
    out[p, i] = pi_p(i) * scale + (p // group_size) * offset_mul.
    advance > 0 (with counter_dev and done_ctr, a zeroed u32[1] of this call site): the launch itself adds `advance` to
    *counter_dev once every workgroup has read it -- the draws of counter = 0 followed by tsm_u64_add, in one launch."""
    if n < 0 or n_perm < 0:
        raise ValueError("random_permutations: negative size")
    if out is None:
        out = torch.empty(n_perm, n, dtype=torch.int64, device=device)
    if out.numel() != n_perm * n:
        raise ValueError("random_permutations: out has the wrong size")
    if advance:
        if counter or counter_dev is None or done_ctr is None:
            raise ValueError("random_permutations: advance needs counter_dev and done_ctr, and no host counter")
        call("tsm_random_permutations_advance", n, n_perm, seed & (2**64 - 1), ptr(counter_dev), int(advance),
             ptr(_chk(done_ctr, torch.int32, "done_ctr")), scale, group_size, offset_mul, ptr(_chk(out, torch.int64, "out")),
             stream_ptr())
        return out
    call("tsm_random_permutations", n, n_perm, seed & (2**64 - 1), counter & (2**64 - 1), ptr(counter_dev), scale,
         group_size, offset_mul, ptr(_chk(out, torch.int64, "out")), stream_ptr())
    return out


def ctde_td_head(q, q_next, rew, terminated, gamma: float, logits, act):
    """CTDEPolicy.learn loss head (ctde.py:149-185) -> (dq, dlogits, scalars[actor_loss, critic_loss])."""
    q, q_next = _chk(q, torch.float32, "q"), _chk(q_next, torch.float32, "q_next")
    B, n_out = q.shape
    logits = _chk(logits, torch.float32, "logits")
    A = logits.shape[1]
    dev = q.device
    dq, dlogits = torch.empty_like(q), torch.empty_like(logits)
    partial = torch.empty(call("tsm_ctde_head_partial_elems", B), dtype=torch.float64, device=dev)
    scalars = torch.empty(2, dtype=torch.float32, device=dev)
    call("tsm_ctde_td_head", ptr(q), ptr(q_next), n_out, ptr(_chk(rew, torch.float32, "rew").reshape(-1)),
         ptr(_chk(terminated, torch.uint8, "terminated").reshape(-1)), float(gamma), ptr(logits),
         ptr(_chk(act, torch.int64, "act").reshape(-1)), A, B, ptr(dq), ptr(dlogits), ptr(partial), ptr(scalars),
         stream_ptr())
    return dq, dlogits, scalars


# --------------------------------------------------------------------------------------------
# fully-connected networks of arbitrary width (csrc/dense.hip)
# --------------------------------------------------------------------------------------------
_ACT = {"none": 0, None: 0, "relu": 1, "tanh": 2}


def mlp_desc(dims, act: str = "relu") -> _abi.tsm_mlp_desc:
    dims = [int(d) for d in dims]
    if not 2 <= len(dims) <= 9:
        raise ValueError("mlp_desc: between 1 and 8 layers")
    d = _abi.tsm_mlp_desc()
    d.n_layers, d.act = len(dims) - 1, _ACT[act]
    for i, v in enumerate(dims):
        d.dims[i] = v
    return d


def mlp_param_count(desc) -> int:
    return call("tsm_mlp_param_count", C.byref(desc))


def mlp_forward(desc, params, x, acts=None):
    """-> (out [B, dims[-1]] view of the last activation block, acts buffer)."""
    x = _chk(x, torch.float32, "x")
    B = x.shape[0]
    if x.shape[1] != desc.dims[0]:
        raise ValueError(f"mlp_forward: input width {x.shape[1]} != dims[0] {desc.dims[0]}")
    n = call("tsm_mlp_act_elems", C.byref(desc), B)
    if acts is None:
        acts = torch.empty(n, dtype=torch.float32, device=x.device)
    call("tsm_mlp_forward", C.byref(desc), ptr(_chk(params, torch.float32, "params")), ptr(x), B, ptr(acts),
         stream_ptr())
    O = desc.dims[desc.n_layers]
    return acts[n - B * O:].view(B, O), acts


def mlp_forward_cond(desc, params, x, run_if, acts=None):
    """`mlp_forward` whose launches are no-ops when the device flag `run_if` (i32[1]) is zero -- a pass a captured graph
    holds but only some inputs need (value_next_select).  The returned tensors hold garbage when it did not run."""
    x = _chk(x, torch.float32, "x")
    B = x.shape[0]
    if x.shape[1] != desc.dims[0]:
        raise ValueError(f"mlp_forward_cond: input width {x.shape[1]} != dims[0] {desc.dims[0]}")
    n = call("tsm_mlp_act_elems", C.byref(desc), B)
    if acts is None:
        acts = torch.empty(n, dtype=torch.float32, device=x.device)
    call("tsm_mlp_forward_cond", C.byref(desc), ptr(_chk(params, torch.float32, "params")), ptr(x), B, ptr(acts),
         ptr(_chk(run_if, torch.int32, "run_if")), stream_ptr())
    O = desc.dims[desc.n_layers]
    return acts[n - B * O:].view(B, O), acts


_GATHER_KINDS = {torch.float32: 0, torch.int32: 1, torch.int64: 2, torch.uint8: 3, torch.bool: 3}


def gather_fields(fields: list, prepared: list | None = None) -> list:
    """Several row-gathers with element conversion in ONE launch (include/tsmarl.h: tsm_gather_fields).  A field is
    `(src, dst)` -- both contiguous, equal numel: a converting copy -- or `(src, dst, T, E, src_row_stride, src_offset)`: dst row
    r = e * T + t (env-major) reads `width = dst.numel() // (T * E)` elements at src row t * E + e (`src` a contiguous time-major
    store, strides in elements).  f32 -> f32; int32 / int64 / uint8 / bool -> int32 / int64 / float32 / uint8 / bool.
    Returns the prepared descriptor arrays: pass them back as `prepared` (with fields=None) to launch the SAME gathers again
    without rebuilding them -- for call sites whose sources and destinations are static allocations (per-call host cost: one
    ctypes call instead of ~30 us of descriptor building)."""
    if prepared is not None:
        for arr, n in prepared:
            call("tsm_gather_fields", arr, n, stream_ptr())
        return prepared
    done = []
    for k0 in range(0, len(fields), _abi.MAX_GATHER_FIELDS):
        chunk = fields[k0:k0 + _abi.MAX_GATHER_FIELDS]
        arr = (_abi.tsm_gather_field * len(chunk))()
        for k, fd in enumerate(chunk):
            src, dst = fd[0], fd[1]
            if not (src.is_contiguous() and dst.is_contiguous() and src.is_cuda and dst.is_cuda):
                raise ValueError("gather_fields: contiguous device tensors only")
            if src.dtype not in _GATHER_KINDS or dst.dtype not in _GATHER_KINDS:
                raise TypeError(f"gather_fields: {src.dtype} -> {dst.dtype} is not supported")
            if len(fd) == 2:
                if src.numel() != dst.numel():
                    raise ValueError("gather_fields: a converting copy needs equal sizes")
                n_rows, width, T, E, stride, off = dst.numel(), 1, 0, 0, 1, 0
            else:
                T, E, stride, off = (int(x) for x in fd[2:6])
                n_rows = T * E
                width = dst.numel() // max(n_rows, 1)
                if width * n_rows != dst.numel() or (n_rows and ((T * E - 1) * stride + off + width) > src.numel()):
                    raise ValueError("gather_fields: the mapping leaves the source or does not tile the destination")
            arr[k] = _abi.tsm_gather_field(src.data_ptr(), dst.data_ptr(), n_rows, T, E, stride, off, width,
                                           _GATHER_KINDS[src.dtype], _GATHER_KINDS[dst.dtype], 0)
        call("tsm_gather_fields", arr, len(chunk), stream_ptr())
        done.append((arr, len(chunk)))
    return done


def any_nonzero_u8(x, out=None):
    """Device flag i32[1] = 1 if any byte of `x` (u8 / bool, contiguous) is non-zero."""
    x = _chk(x, torch.uint8, "x")
    flag = out if out is not None else torch.empty(1, dtype=torch.int32, device=x.device)
    call("tsm_any_nonzero_u8", ptr(x), x.numel(), ptr(flag), stream_ptr())
    return flag


def value_next_select(v_s, v_last, v_full, flag, T: int, U: int, out=None):
    """V(obs_next) [T, U] of chained rows: v_s shifted by one slot + the last slot's own values `v_last` [U], or the full
    pass `v_full` [T, U] when `flag` says an episode ended before the last slot (include/tsmarl.h)."""
    v_s, v_last, v_full = (_chk(t, torch.float32, n) for t, n in ((v_s, "v_s"), (v_last, "v_last"), (v_full, "v_full")))
    if v_s.numel() != T * U or v_full.numel() != T * U or v_last.numel() != U:
        raise ValueError("value_next_select: shapes do not match T x U")
    res = out if out is not None else torch.empty(T, U, dtype=torch.float32, device=v_s.device)
    call("tsm_value_next_select", ptr(v_s), ptr(v_last), ptr(v_full), ptr(_chk(flag, torch.int32, "flag")), T, U, ptr(res),
         stream_ptr())
    return res


def value_next_index(v_s, done, T: int, U: int, lanes_per_env: int = 1, out=None):
    """V(obs_next) [T, U] of a buffer without an obs_next store (ignore_obs_next, buffer_base.py:612-616): the next slot's
    V(obs), or the row's own at an episode end (`done` u8 [T, U / lanes_per_env]) and in the newest slot."""
    v_s = _chk(v_s, torch.float32, "v_s")
    done = done.contiguous().view(torch.uint8) if done.dtype == torch.bool else _chk(done, torch.uint8, "done")
    if v_s.numel() != T * U or done.numel() * lanes_per_env != T * U:
        raise ValueError("value_next_index: shapes do not match T x U")
    res = out if out is not None else torch.empty(T, U, dtype=torch.float32, device=v_s.device)
    call("tsm_value_next_index", ptr(v_s), ptr(done), T, U, lanes_per_env, ptr(res), stream_ptr())
    return res


def value_next_select_env_major(v_s, v_last, v_full, flag, E: int, T: int, U: int, out=None):
    """`value_next_select` for env-major rows [E, T, U] (the per-agent batches of the MARL trainers)."""
    v_s, v_last, v_full = (_chk(t, torch.float32, n) for t, n in ((v_s, "v_s"), (v_last, "v_last"), (v_full, "v_full")))
    if v_s.numel() != E * T * U or v_full.numel() != E * T * U or v_last.numel() != E * U:
        raise ValueError("value_next_select_env_major: shapes do not match E x T x U")
    res = out if out is not None else torch.empty(E * T, U, dtype=torch.float32, device=v_s.device)
    call("tsm_value_next_select_env_major", ptr(v_s), ptr(v_last), ptr(v_full), ptr(_chk(flag, torch.int32, "flag")), E, T, U,
         ptr(res), stream_ptr())
    return res


def mlp_n_split(B: int) -> int:
    """Default number of gradient slabs for a batch of B rows."""
    return max(1, min(64, -(-B // 256)))


def mlp_backward(desc, params, x, acts, d_out, n_split: int = 0, slabs=None, slab_stride: int = 0):
    """Gradient slabs [n_split, n_param] of sum(out * d_out) w.r.t. the flat parameter vector.
    `slabs` may point INTO the slabs of a larger joint parameter vector (slab_stride = its parameter count)."""
    x, d_out = _chk(x, torch.float32, "x"), _chk(d_out, torch.float32, "d_out")
    B = x.shape[0]
    if n_split <= 0:
        n_split = mlp_n_split(B)
    n_param = params.numel()
    if slabs is None:
        slabs = torch.empty(n_split, n_param, dtype=torch.float32, device=x.device)
    d_acts = torch.empty_like(acts)
    call("tsm_mlp_backward", C.byref(desc), ptr(params), ptr(x), B, ptr(acts), ptr(d_out), ptr(d_acts), n_split,
         slabs.data_ptr(), slab_stride, stream_ptr())
    return slabs


def device_info() -> dict:
    n_cu, wave, hbm = C.c_int(), C.c_int(), C.c_int64()
    name = C.create_string_buffer(64)
    call("tsm_device_info", C.byref(n_cu), C.byref(wave), C.byref(hbm), name)
    return dict(n_cu=n_cu.value, wave_size=wave.value, hbm_bytes=hbm.value, arch=name.value.decode())


KERNEL_OPTIONS = ("actor_tile", "split_bf16", "generic_kernels", "rollout_form")


_options_cache = None  # kernel_options() as last read: options change only through set_kernel_option (the environment defaults
                       # are resolved once, at first use), and the tuple is part of every per-call graph key


def kernel_option(name: str) -> int:
    """Value of a kernel selection option (include/tsmarl.h: tsm_kernel_option_get)."""
    v = C.c_int32()
    call("tsm_kernel_option_get", name.encode(), C.byref(v))
    return v.value


def set_kernel_option(name: str, value: int) -> int:
    """Override a kernel selection rule for this process; returns the previous value."""
    global _options_cache
    old = kernel_option(name)
    _options_cache = None
    call("tsm_kernel_option_set", name.encode(), int(value))
    return old


def kernel_options() -> tuple:
    """All option values in KERNEL_OPTIONS order: part of every launch-cache (hipGraph) key."""
    global _options_cache
    if _options_cache is None:
        _options_cache = tuple(kernel_option(n) for n in KERNEL_OPTIONS)
    return _options_cache


class graph_capture:
    """`with ops.graph_capture(graph[, pool=...]):` -- torch.cuda.graph with Python's cyclic garbage collector held off while the
    stream is capturing.  A collection that happens to run inside a capture and frees a device tensor or destroys another
    hipGraph is an operation the capturing stream does not permit: the process aborts ("Fatal Python error: Aborted ...
    Garbage-collecting", seen in the test suite once enough graph-holding objects had become garbage)."""

    def __init__(self, graph, pool=None):
        self._cm = torch.cuda.graph(graph) if pool is None else torch.cuda.graph(graph, pool=pool)
        self._was = False

    def __enter__(self):
        import gc

        self._was = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            return self._cm.__enter__()
        except BaseException:
            if self._was:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc

        try:
            return self._cm.__exit__(*exc)
        finally:
            if self._was:
                gc.enable()


class gc_hold:
    """`with ops.gc_hold():` -- one collection now, none inside: for capture sites that drive `capture_begin` / `capture_end`
    themselves (a side stream, thread-local capture mode) and therefore cannot use `graph_capture`."""

    def __enter__(self):
        import gc

        self._was = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc

        if self._was:
            gc.enable()
        return False


class kernel_override:
    """`with ops.kernel_override(actor_tile=64): ...` -- options set inside, restored on exit."""

    def __init__(self, **opts: int):
        self._opts, self._old = opts, {}

    def __enter__(self):
        for k, v in self._opts.items():
            self._old[k] = set_kernel_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self._old.items():
            set_kernel_option(k, v)
        return False


__all__ = [n for n in dir() if not n.startswith("_")]
_ = _abi
