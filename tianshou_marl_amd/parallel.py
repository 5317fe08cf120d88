"""Env-sharded data parallelism: one process per GPU, one flat all-reduce per gradient step.

The reference has no distributed backend at all (only single-process nn.DataParallel wrappers,
/root/reference/tianshou/utils/net/common.py:477-519), so this is new design:
environments are independent, so GPU g owns its own env shard, device buffer, rollout and GAE with NO
data-path collective; only the shared-policy update exchanges data -- the flat gradient of the
actor+critic vector (~11 k f32 = 45 KB: latency-bound over xGMI) is summed with one
`torch.distributed.all_reduce` (backend "nccl" == RCCL on ROCm; "gloo" on CPU for tests) per
gradient step, after which every rank applies the identical Adam step, keeping replicas bit-identical.
Advantage normalisation (ppo.py:184-186) uses the statistics of the GLOBAL minibatch -- the union of the ranks'
minibatches of one gradient step: the per-minibatch (n, sum, sum of squares) of all minibatches of an update travel in
ONE small f64 all-reduce per update (`GradSync.merge_adv_stats_`), so the sharded update is the single-GPU update on the
union minibatch; `global_adv_stats=False` keeps the rank-local statistics and saves that collective.
"""
from __future__ import annotations

import os

import torch


def shard_range(n_total: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous env range [lo, hi) owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def p2p_mode() -> str:
    """TSM_P2P_ALLREDUCE: "1" = the peer-memory all-reduce is REQUIRED (setup or handshake failure raises), "0" = off (the
    process group's all-reduce), anything else / unset = "auto": the first-use handshake decides."""
    return {"1": "on", "0": "off"}.get(os.environ.get("TSM_P2P_ALLREDUCE", "auto"), "auto")


class P2PAllReduce:
    """One-shot all-reduce of a small f32 vector over peer-mapped memory (csrc/p2p.hip): every rank stores each element into
    every peer's IPC-mapped inbox as ONE 8-byte word {value, stamp of the call} (no fences, flags or barriers), polls its own
    inbox for the peers' elements and sums the senders in rank order -- ONE launch per call, no ring, the same bits on every
    rank; `adam_step` is the replica's whole gradient step (slab sum, exchange, Adam) in one launch.
    Build it with `P2PAllReduce.negotiate`: setup, a first-use handshake and the ranks' agreement on the outcome."""

    def __init__(self, dist, group, device: torch.device, max_floats: int) -> None:
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.max_floats = int(max_floats)
        self.device = device
        self._h = None
        self._err_host = None  # pinned int32: the handle's error word as of the last `post_check`

    @classmethod
    def negotiate(cls, dist, group, device: torch.device, max_floats: int, timeout_s: float | None = None):
        """-> a working P2PAllReduce on EVERY rank, or None on every rank.  The phases below issue the same collectives on
        every rank whatever happens locally (a rank that raised half-way would leave its peers inside a collective), and the
        ranks agree on each outcome with one MIN all-reduce:
          1. create the inbox, export its IPC handle              (local; failure -> ok = 0)
          2. all-gather the handles + the ok flags                 (always)
          3. map every peer's inbox                                (only if every rank's phase 1 passed)
          4. agree; barrier (every inbox mapped before the first store)
          5. handshake: one stamped word per peer each way, bounded spin, outside any capture (csrc/p2p.hip)
          6. agree.  Any failure: every rank closes its handle and returns None -- the process group's all-reduce serves for
             the rest of the process (a decision taken once, before anything is captured; never revisited mid-run)."""
        import ctypes as C
        import sys

        from . import _abi

        self = cls(dist, group, device, max_floats)
        on_dev = dist.get_backend(group) == "nccl"
        why = []

        def agree(ok: bool) -> bool:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_dev else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(int(t.item()))

        nb, mine, ok = 64, None, True
        try:  # 1
            torch.cuda.set_device(device)
            nb = int(_abi.call("tsm_p2p_ipc_handle_bytes"))
            h = C.c_void_p()
            _abi.call("tsm_p2p_create", self.rank, self.world, self.max_floats, C.byref(h))
            self._h = h
            if timeout_s or os.environ.get("TSM_P2P_TIMEOUT_S"):
                _abi.call("tsm_p2p_set_timeout", self._h, float(timeout_s or os.environ["TSM_P2P_TIMEOUT_S"]))
            mine = (C.c_ubyte * nb)()
            _abi.call("tsm_p2p_export", self._h, mine)
            if os.environ.get("TSM_P2P_FAIL_SETUP") == str(self.rank):  # (rehearsal of the fall-back: tests)
                raise RuntimeError("TSM_P2P_FAIL_SETUP")
        except Exception as e:  # noqa: BLE001
            ok = False
            why.append(f"setup: {type(e).__name__}: {e}")
        t = torch.zeros(nb + 1, dtype=torch.uint8)  # 2: [handle bytes | ok]
        if ok:
            t[:nb] = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
            t[nb] = 1
        t = t.to(device) if on_dev else t
        gathered = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(gathered, t, group=group)
        gathered = [g.cpu() for g in gathered]
        all_ok = all(int(g[nb]) == 1 for g in gathered)
        if all_ok:  # 3
            try:
                for r, g in enumerate(gathered):
                    if r != self.rank:
                        _abi.call("tsm_p2p_import", self._h, r, (C.c_ubyte * nb).from_buffer_copy(bytes(g[:nb].numpy().tobytes())))
            except Exception as e:  # noqa: BLE001
                ok = False
                why.append(f"mapping a peer's inbox: {type(e).__name__}: {e}")
        all_ok = agree(ok and all_ok)  # 4
        dist.barrier(group=group)
        if all_ok:  # 5
            try:
                res = C.c_int32(0)
                if os.environ.get("TSM_P2P_FAIL_HANDSHAKE") != str(self.rank):  # (rehearsal: this rank never sends)
                    _abi.call("tsm_p2p_handshake", self._h, C.byref(res), _abi.stream_ptr())
                ok = bool(res.value)
                if not ok:
                    why.append("handshake: a peer's word did not arrive within the time limit (or the sum was wrong)")
            except Exception as e:  # noqa: BLE001
                ok = False
                why.append(f"handshake: {type(e).__name__}: {e}")
            all_ok = agree(ok)  # 6
        if all_ok:
            if not (timeout_s or os.environ.get("TSM_P2P_TIMEOUT_S")):
                # mid-run a rank may legitimately be late by seconds (rank 0 writes a checkpoint or evaluates, a first graph
                # capture, a GC pause): the handshake's ~2 s would turn that into a dead handle.  A lost peer still ends the
                # job -- `raise_if_failed` at every update's statistics read -- only later.
                _abi.call("tsm_p2p_set_timeout", self._h, float(os.environ.get("TSM_P2P_RUN_TIMEOUT_S", "60")))
            return self
        self.close()
        print(f"[rank {self.rank}] peer-memory all-reduce not used" + (": " + "; ".join(why) if why else " (another rank declined)")
              + " -- gradients take the process group's all-reduce", file=sys.stderr)
        return None

    def all_reduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        from . import _abi

        if flat.dtype != torch.float32 or not flat.is_cuda or not flat.is_contiguous() or flat.numel() > self.max_floats:
            raise ValueError(f"P2PAllReduce: needs a contiguous f32 device tensor of at most {self.max_floats} elements")
        _abi.call("tsm_p2p_all_reduce", self._h, flat.data_ptr(), flat.numel(), _abi.stream_ptr())
        return flat

    def adam_step(self, param, grad_slabs, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                  step_dev=None, lr_dev=None, image=None, image_map=None):
        """The replicas' whole gradient step in ONE launch (csrc/p2p.hip, p2p_adam_kernel): slab sum x 1 / world, one-shot
        all-reduce and Adam, each workgroup on its own slice of the vector.  Bit-identical to ops.reduce_slabs ->
        all_reduce_sum_ -> ops.adam_step (without a gradient-norm clip)."""
        from . import _abi
        from ._abi import ptr

        n = param.numel()
        if n > self.max_floats or param.dtype != torch.float32:
            raise ValueError(f"P2PAllReduce.adam_step: needs an f32 vector of at most {self.max_floats} elements")
        grad_slabs = grad_slabs.reshape(-1, n)
        _abi.call("tsm_p2p_adam_step", self._h, ptr(param), ptr(grad_slabs), grad_slabs.shape[0], n, ptr(exp_avg), ptr(exp_avg_sq),
                  int(step), ptr(step_dev), float(lr), ptr(lr_dev), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                  ptr(image), ptr(image_map), _abi.stream_ptr())
        return param

    def check(self) -> None:
        """Raise if a peer failed to answer inside the kernels' bounded spin (synchronises the device)."""
        from . import _abi

        if self._h is not None and _abi.call("tsm_p2p_failed", self._h):
            raise RuntimeError("P2P all-reduce: a peer did not answer within the time limit (a rank died or fell out of step)")

    def post_check(self) -> None:
        """Queue a copy of the error word into pinned host memory behind everything launched so far on the current stream (no
        synchronisation).  Every update queues it in front of the event of its loss statistics."""
        from . import _abi

        if self._h is None:
            return
        if self._err_host is None:
            self._err_host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        _abi.call("tsm_p2p_error_async", self._h, self._err_host.data_ptr(), _abi.stream_ptr())

    def raise_if_failed(self) -> None:
        """After the host has waited for the point of the stream `post_check` was queued at (the statistics' event / copy):
        raise if a peer's word did not arrive -- the gradient steps since then were skipped, never half-applied."""
        if self._err_host is not None and int(self._err_host[0]) != 0:
            raise RuntimeError("P2P all-reduce: a peer did not answer within the time limit (a rank died or fell out of step); "
                               "the gradient steps of this update were not applied")

    def close(self) -> None:
        from . import _abi

        if getattr(self, "_h", None) is not None:
            _abi.call("tsm_p2p_destroy", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class GradSync:
    """Sums a flat gradient across ranks and divides by world size (mean of per-shard mean losses)."""

    def __init__(self, dist, group=None) -> None:
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.global_adv_stats = True  # minibatch advantage statistics over all ranks (merge_adv_stats_)
        self._stat_codec = None       # (pack, unpack) callables; None = the HIP kernels (ops.ppo_adv_stats_pack / _unpack)
        self._probe_device = torch.device("cpu")  # attach_data_parallel points it at the replica's device for RCCL
        self.p2p: P2PAllReduce | None = None      # TSM_P2P_ALLREDUCE=1: the gradient takes the one-shot peer-memory path

    def all_reduce_mean_(self, flat: torch.Tensor) -> torch.Tensor:
        self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world)
        return flat

    def all_reduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        """Plain sum (the caller pre-scales its contribution by 1/world inside the slab reduction kernel, which saves
        an elementwise launch per gradient step)."""
        if self.p2p is not None and flat.dtype == torch.float32 and flat.numel() <= self.p2p.max_floats:
            return self.p2p.all_reduce_sum_(flat)
        self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
        return flat

    def post_check(self) -> None:
        """See `P2PAllReduce.post_check`; a no-op without the peer-memory path."""
        if self.p2p is not None:
            self.p2p.post_check()

    def raise_if_failed(self) -> None:
        if self.p2p is not None:
            self.p2p.raise_if_failed()

    def fused_step_ok(self, max_grad_norm, n: int) -> bool:
        """Can a gradient step of n parameters take the one-launch form (P2PAllReduce.adam_step)?  Needs the peer-memory path,
        no gradient-norm clip (the global norm is a grid-wide dependency), and TSM_P2P_FUSED_ADAM != 0."""
        return (self.p2p is not None and not max_grad_norm and n <= self.p2p.max_floats
                and os.environ.get("TSM_P2P_FUSED_ADAM", "1") != "0")

    def enable_p2p(self, device: torch.device, max_floats: int, required: bool = False) -> bool:
        """Route f32 gradient sums of up to `max_floats` elements through `P2PAllReduce` if every rank's setup and first-use
        handshake pass (`P2PAllReduce.negotiate`: the same decision on every rank); otherwise the process group's all-reduce
        stays -- or, with `required`, every rank raises."""
        self.p2p = P2PAllReduce.negotiate(self.dist, self.group, device, max_floats)
        if self.p2p is None and required:
            raise RuntimeError("TSM_P2P_ALLREDUCE=1: the peer-memory all-reduce could not be set up on every rank (see stderr)")
        return self.p2p is not None

    def require_equal(self, value: int, what: str) -> None:
        """Every rank must issue the same number of gradient all-reduces per update, or the job deadlocks: env shards of
        different sizes can split into a different number of minibatches (Batch.split merge-last rule), and with
        `n_episode` collection the number of valid rows differs per rank and per update.  One tiny all-reduce, EVERY call
        (never cached: a rank that skipped this collective because of rank-local state while another rank issued it would
        pair it with the other's gradient all-reduce).  Eager updates call it once per update; the captured update calls
        it at capture time -- its step count is a function of the graph key, and the graph path is only taken for
        host-known uniform fills (`n_step` collection), a choice every rank makes alike."""
        t = torch.tensor([int(value), -int(value)], dtype=torch.int64, device=self._probe_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        hi, lo = int(t[0].item()), -int(t[1].item())
        if hi != lo:
            raise ValueError(
                f"data-parallel ranks disagree on {what}: between {lo} and {hi} (this rank: {value}).  Give every rank the "
                "same number of environments (and the same batch_size / repeat) so that all replicas take the same "
                "number of gradient steps.")

    def check_same(self, names: list, what: str) -> None:
        """Every rank must train the same policy groups in a step (random matchmaking has to be seeded identically on
        all ranks): a stable 31-bit hash of the names is compared with one tiny all-reduce, every call (not cached -- a
        rank that skipped the collective would hang the others)."""
        import zlib

        h = zlib.crc32("|".join(str(n) for n in names).encode()) & 0x7FFFFFFF
        t = torch.tensor([h, -h], dtype=torch.int64, device=self._probe_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        if int(t[0].item()) != -int(t[1].item()):
            raise ValueError(f"data-parallel ranks disagree on {what} (this rank: {list(names)}): seed the trainers' "
                             "matchmaking identically on every rank")

    def merge_adv_stats_(self, stats: torch.Tensor, mb_start: torch.Tensor) -> torch.Tensor:
        """stats [n_mb, 2] = rank-local (mean, unbiased std) of the advantages of each minibatch (its rows on this rank:
        mb_start[k+1] - mb_start[k]) -> in place, the (mean, unbiased std) of the union of all ranks' minibatch (SURVEY.md
        section 8e: "the advantage statistics make the sharded minibatch normalisation identical to a single global
        minibatch").  Three launches: pack to f64 (n, sum x, sum x^2) (`tsm_ppo_adv_stats_pack`), ONE all-reduce for every
        minibatch of the update, unpack; every rank ends with bit-identical statistics."""
        if not self.wants_global_adv_stats():
            return stats
        pack_fn, unpack_fn = self._stat_codec or self._default_codec()
        pack = pack_fn(stats, mb_start)
        self.dist.all_reduce(pack, op=self.dist.ReduceOp.SUM, group=self.group)
        unpack_fn(pack, stats)
        return stats

    def wants_global_adv_stats(self) -> bool:
        """(a one-rank group skips the merge, except in the single-GPU rehearsal of the multi-GPU path, TSM_FORCE_DIST)"""
        import os

        return bool(self.global_adv_stats and (self.world > 1 or os.environ.get("TSM_FORCE_DIST")))

    @staticmethod
    def _default_codec():
        from . import ops

        return ops.ppo_adv_stats_pack, ops.ppo_adv_stats_unpack

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        self.dist.broadcast(t, src=src, group=self.group)
        return t


def lockstep_steps(jobs, results: list, packed_buf: torch.Tensor | None = None):
    """The lock-step of `learn_lockstep` as a generator of ITS collectives: advances every live generator to its next
    synchronisation point, packs what fell due into one buffer, yields it (the driver sums it over the ranks in place),
    hands the sums back.  `results[i]` receives generator i's return value.  `packed_buf`: a preallocated buffer to pack
    into (captured graphs want fixed addresses); tensors of one step share a dtype by construction (every group yields its
    statistics pack, then its gradients), a mixed step is packed in the widest."""
    alive = dict(enumerate(jobs))
    while alive:
        due = {}
        for i, g in list(alive.items()):
            try:
                due[i] = next(g)
            except StopIteration as stop:
                results[i] = stop.value
                del alive[i]
        if not due:
            continue
        flats = list(due.values())
        if len(flats) == 1:
            yield flats[0]
            continue
        total = sum(f.numel() for f in flats)
        dt = flats[0].dtype
        if packed_buf is not None and all(f.dtype == packed_buf.dtype for f in flats) and total <= packed_buf.numel():
            packed = packed_buf[:total]
            o = 0
            for f in flats:
                packed[o:o + f.numel()].copy_(f.reshape(-1))
                o += f.numel()
        else:
            for f in flats:
                dt = torch.promote_types(dt, f.dtype)
            packed = torch.cat([f.reshape(-1).to(dt) for f in flats])
        yield packed
        o = 0
        for f in flats:
            f.copy_(packed[o:o + f.numel()].view_as(f))
            o += f.numel()


def learn_lockstep(jobs, sync: "GradSync") -> list:
    """Advance several policies' updates (generators from `PPO.learn_steps` / `_update_steps`) gradient step by gradient
    step.  The flat gradients that fall due at the same step -- one per policy GROUP that trains this step -- are packed
    into ONE buffer and summed over the ranks with ONE all-reduce ("one reduce per policy group
    that trained this step, packed in one buffer"), then handed back so that each group applies its own Adam step.
    Every rank must submit the same groups in the same order.  Returns the generators' results in job order."""
    results = [None] * len(jobs)
    for t in lockstep_steps(jobs, results):
        sync.all_reduce_sum_(t)
    return results


def lockstep_graphs_enabled() -> bool:
    """TSM_LOCKSTEP_GRAPH=0: `learn()` of data-parallel replicas stays on eager lock-step launches."""
    import os

    return os.environ.get("TSM_LOCKSTEP_GRAPH", "1") != "0"


def learn_lockstep_graph(jobs, sync: "GradSync", names: list | None = None) -> list:
    """`learn_lockstep` from captured graphs.  jobs = [(policy, batch, batch_size, repeat)], every policy a replica on `sync`
    that offers `_learn_static / _learn_load / _learn_finish` (PPO, GenericPPO).  Each policy's batch goes into ITS static buffers;
    the lock-step of their launch sequences (`_learn_static(...)["body"]`) -- kernels of all groups interleaved, one packed
    reduction per gradient step -- is captured once per (policies, shapes):
      * collectives capturable (RCCL, probe passed: `policy.graph_collectives`): ONE hipGraph with the all-reduces inside;
      * otherwise (gloo, failed probe, TSM_GRAPH_COLLECTIVES=0): every stretch between two collectives is its own hipGraph
        on one shared pool and the collectives run eagerly in between -- as `PPO.update` does.
    The same launches in the same order as the eager lock-step, hence the same bits.  Every call first makes sure that all
    ranks submit the same groups (`names`) with the same row counts, minibatch sizes and repeats -- they select the graph,
    and ranks replaying different graphs would pair different collectives (one tiny all-reduce, never cached)."""
    rows = lambda p, b: p._learn_rows(b) if hasattr(p, "_learn_rows") else len(b.rew)  # noqa: E731  (store handles: PPO._batch_store)
    has_tr = lambda p, b: "truncated" in b or (hasattr(p, "_batch_store") and p._batch_store(b) is not None)  # noqa: E731
    sync.check_same([f"{nm}:{rows(p, b)}:{bs}:{rep}:{int(has_tr(p, b))}"
                     for nm, (p, b, bs, rep) in zip(names or range(len(jobs)), jobs)],
                    "the policy groups that train in this step (and their row counts)")
    ws = [p._learn_static(rows(p, b), bs, rep, has_tr(p, b)) for p, b, bs, rep in jobs]
    for (p, b, _, _), w in zip(jobs, ws):
        p._learn_load(w, b)
    if not all(w.get("warm", True) for w in ws):
        # a policy whose launch sequence has not run on these buffers yet (GenericPPO: one-time kernel attributes and
        # workspaces are set by the first pass): this call is the eager lock-step on the static buffers -- the same launches and
        # collectives, hence the same bits --, the next one captures
        res = [None] * len(jobs)
        for t in lockstep_steps([w["body"]() for w in ws], res, None):
            sync.all_reduce_sum_(t)
        for w in ws:
            w["warm"] = True
        return [p._learn_finish(w) for (p, _, _, _), w in zip(jobs, ws)]
    inline = all(getattr(p, "graph_collectives", False) for p, _, _, _ in jobs)
    from . import ops as _ops

    key = (tuple((id(p), id(w)) for (p, _, _, _), w in zip(jobs, ws)), inline, _ops.kernel_options())
    cache = sync.__dict__.setdefault("_lockstep_graphs", {})
    g = cache.get(key)
    if g is None:
        total = sum(p.net.flat.numel() for p, _, _, _ in jobs)
        g = dict(packed=torch.empty(total, dtype=torch.float32, device=ws[0]["obs"].device) if len(jobs) > 1 else None,
                 results=[None] * len(jobs),
                 # the key is made of object ids: keep the objects alive as long as the graphs that replay into their buffers
                 keep=([p for p, _, _, _ in jobs], ws))

        def steps():
            return lockstep_steps([w["body"]() for w in ws], g["results"], g["packed"])

        if inline:
            graph = torch.cuda.CUDAGraph()

            def run():
                for t in steps():
                    sync.all_reduce_sum_(t)

            jobs[0][0]._capture_graph(graph, run)  # side stream, thread-local capture mode; a failure is fatal (see there)
            g["graph"] = graph
        else:
            pool = torch.cuda.graph_pool_handle()
            segs, gen, more = [], steps(), True
            while more:
                g_ = torch.cuda.CUDAGraph()
                t_ = None
                from . import ops as _ops

                with _ops.graph_capture(g_, pool=pool):
                    try:
                        t_ = next(gen)
                    except StopIteration:
                        more = False
                segs.append((g_, t_))
            g["segments"] = segs
        cache[key] = g
    if "graph" in g:
        g["graph"].replay()
    else:
        for g_, t_ in g["segments"]:
            g_.replay()
            if t_ is not None:
                sync.all_reduce_sum_(t_)
    return [p._learn_finish(w) for (p, _, _, _), w in zip(jobs, ws)]


def _maybe_enable_p2p(sync: "GradSync", device: torch.device, max_floats: int) -> None:
    """TSM_P2P_ALLREDUCE (p2p_mode): on = required, off, auto = the handshake decides.  Peer memory is a one-node mechanism
    between GPUs: a CPU-only replica (tests) has nothing to map."""
    mode = p2p_mode()
    if mode != "off" and device.type == "cuda":
        sync.enable_p2p(device, max_floats, required=mode == "on")


def attach_data_parallel(algo, dist, group=None, global_adv_stats: bool = True) -> GradSync:
    """Make `algo` a data-parallel replica: parameters and optimizer state are broadcast from rank 0, and every gradient
    step all-reduces the flat gradient.  `algo` is a PPO-family algorithm, or a policy manager
    (FlexibleMultiAgentPolicyManager: every distinct policy of `.policies` is attached to ONE shared GradSync, and the
    MARL trainers then reduce the groups that train in a step together, `learn_lockstep`)."""
    sync = GradSync(dist, group)
    sync.global_adv_stats = bool(global_adv_stats)
    if hasattr(algo, "policies") and not hasattr(algo, "net"):
        seen = []
        for pol in algo.policies.values():
            if any(pol is q for q in seen) or not hasattr(pol, "net"):
                continue
            seen.append(pol)
            sync.broadcast_(pol.net.flat.data)
            sync.broadcast_(pol.exp_avg)
            sync.broadcast_(pol.exp_avg_sq)
            pol._grad_sync = sync
            if hasattr(pol.net, "sync_image"):
                pol.net.sync_image()
            # the packed reduction of the groups that train in a step is captured with their launches when the backend's
            # collectives can be (RCCL), and runs between segmented graphs otherwise (learn_lockstep_graph)
            if dist.get_backend(group) != "nccl" and os.environ.get("TSM_GRAPH_COLLECTIVES") != "force":
                pol.graph_collectives = False
        if dist.get_backend(group) == "nccl" and seen:
            sync._probe_device = seen[0].net.flat.device
        if seen:  # the packed gradient of every group that can train in one step
            _maybe_enable_p2p(sync, seen[0].net.flat.device, sum(p.net.flat.numel() for p in seen))
        algo._grad_sync = sync
        return sync
    sync.broadcast_(algo.net.flat.data)
    sync.broadcast_(algo.exp_avg)
    sync.broadcast_(algo.exp_avg_sq)
    if hasattr(algo.net, "sync_image"):
        algo.net.sync_image()  # the padded LDS image is a cache of `flat`: refresh it behind the broadcast
    algo._grad_sync = sync
    _maybe_enable_p2p(sync, algo.net.flat.device, algo.net.flat.numel())
    if dist.get_backend(group) == "nccl":
        sync._probe_device = algo.net.flat.device
    # only RCCL ("nccl") collectives can be captured into a hipGraph; with any other backend the update stays on eager
    # launches (TSM_GRAPH_COLLECTIVES=force overrides: used to rehearse the failed-capture fallback)
    if dist.get_backend(group) != "nccl" and os.environ.get("TSM_GRAPH_COLLECTIVES") != "force":
        algo.graph_collectives = False
    return sync


# ---- can this box capture RCCL collectives into a hipGraph? -----------------------------------------------------------------
def probe_collective_capture(rank: int, world: int, local_rank: int, timeout_s: float = 240.0) -> bool:
    """Try, in a CHILD process, what the update graph does with more than one rank: create the RCCL communicator, capture
    an all-reduce into a hipGraph on a side stream, replay it, check the sum.  A capture that fails leaves the stream --
    and with it the process -- unusable (hipErrorStreamCaptureInvalidated on every later collective), so the question
    cannot be asked in the process that goes on to do the work.  Call it on every rank BEFORE the caller touches the GPU
    (the children rendezvous among themselves on MASTER_PORT + 1; at most `world` processes use the GPUs at any time).
    False (child failed, crashed or timed out) -> run with TSM_GRAPH_COLLECTIVES=0 (eager collectives)."""
    import os
    import subprocess
    import sys

    env = dict(os.environ)
    env["MASTER_ADDR"] = env.get("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29533")) + 1)
    env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank))
    try:
        r = subprocess.run([sys.executable, "-m", "tianshou_marl_amd.parallel", "--probe-capture"], env=env,
                           timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    except subprocess.TimeoutExpired:
        print(f"[rank {rank}] collective-capture probe timed out after {timeout_s:.0f} s", file=sys.stderr)
        return False
    if r.returncode != 0:  # say why, for whoever reads the job's log
        tail = (r.stderr or b"").decode(errors="replace").strip().splitlines()[-6:]
        print(f"[rank {rank}] collective-capture probe failed (exit {r.returncode})" + ("".join("\n    " + t for t in tail)),
              file=sys.stderr)
    return r.returncode == 0


def _probe_child() -> int:
    import datetime
    import os

    import torch.distributed as dist

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        x = torch.ones(11142, device=dev)
        dist.all_reduce(x)  # communicator up, outside any capture
        torch.cuda.synchronize()
        x.fill_(1.0)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        from . import ops as _ops

        with _ops.gc_hold(), torch.cuda.stream(side):  # (no cyclic-GC pass inside a capture: ops.graph_capture)
            g.capture_begin(capture_error_mode="thread_local")
            dist.all_reduce(x)
            y = x * 0.5
            g.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        ok = bool(torch.allclose(x, torch.full_like(x, float(world))) and torch.allclose(y, torch.full_like(y, world / 2.0)))
        x.fill_(2.0)
        g.replay()
        torch.cuda.synchronize()
        ok = ok and bool(torch.allclose(x, torch.full_like(x, 2.0 * world)))
        dist.barrier()
        return 0 if ok else 1
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import os
    import sys

    if "--probe-capture" in sys.argv:
        if os.environ.get("TSM_PROBE_FAIL") == "1":  # rehearsal of the fall-back: pretend this box cannot capture
            sys.exit(1)
        try:
            code = _probe_child()
        except BaseException as e:  # noqa: BLE001  (any failure means "do not capture")
            print(f"probe: {type(e).__name__}: {e}", file=sys.stderr)
            code = 1
        sys.exit(code)
