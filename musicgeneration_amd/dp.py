"""Data-parallel training over RCCL/xGMI: one process per GPU, minibatch rows sharded across ranks,
ONE exchange step per optimiser step -- a gradient all-reduce (sum) of the model's flat fp32
gradient buffer, issued bucket by bucket (bucket = one layer's contiguous slice) from inside the
backward pass so it overlaps the remaining backward kernels.

The reference has no live multi-GPU path (``--multi_gpu`` is parsed and ignored,
train.py:94-97,232-235); the parity statement is "N ranks x batch B/N == 1 rank x batch B".
Bucket sizes at cfg2: 1.51 M params = 6.0 MB fp32 per layer; xGMI ring all-reduce of 6 MB is
~0.1 ms per bucket, far below a layer's backward time, so overlap matters more than algorithm.
"""
from __future__ import annotations

import contextlib
import time
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


class DataParallel:
    """Wraps a MusicTransformer: hooks its bucket-ready callbacks, broadcasts rank 0's weights."""

    MAX_TIMED = 256        # per-bucket (issue, complete) pairs kept between two bucket_ms() calls (a train loop that never asks)

    def __init__(self, model, process_group=None, force_collectives: bool = False, groups: Optional[int] = None):
        """``groups``: merge the per-layer buckets into that many contiguous groups (``bench.py --buckets 2``: fewer, larger
        collectives -- one at the half-way point of backward, one at its end; DESIGN.md section 4, co-residency mitigation 3).
        ``force_collectives``: issue every collective even in a world of one rank (broadcast, per-bucket all-reduce,
        loss-weight all-reduce).  A sum over one rank is the identity, so results must equal a run without data
        parallelism up to the summation-order noise of the kernels' own fp32 atomics (bit-identical under
        MGX_DETERMINISTIC=1) -- which lets a single-GPU box prove RCCL communicator setup, stream ordering against the backward
        kernels and the flat-buffer views before the first multi-GPU run (tests/test_gpu_dp.py)."""
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.force = bool(force_collectives) and dist.is_initialized()
        self._works: List = []
        self._sync = True
        self._exposed: List = []          # (event before wait, event after wait) pairs of the compute stream
        self.measure_overlap = False
        # overlap = False ("no-overlap" mode, bench.py --no-overlap): the buckets are only NOTED as they become ready and all
        # all-reduces are issued after backward, in wait_all().  Same results; the difference in step time against the default is
        # what the overlap buys -- on a first multi-GPU run it separates "RCCL is slow" from "the overlap was lost" (DESIGN 4).
        self.overlap = True
        self._pending: List[str] = []
        # per-bucket issue -> complete times (measure_overlap): name -> list of (issue event, done event) [nccl] or seconds [gloo]
        self._bucket_t: Dict[str, List] = {}
        self._host_t0: Dict[int, tuple] = {}
        self._mstream = None              # side stream that waits for each collective and records its completion (nccl only)
        self._groups: Optional[List[tuple]] = None    # merged buckets: (group name, lo, hi, member names in flat-buffer order)
        self._group_of: Dict[str, int] = {}
        self._arrived: Dict[int, int] = {}
        if groups is not None:
            self.merge_buckets(groups)
        model._dp = self
        if self.world > 1 or self.force:
            st = model.store()
            dist.broadcast(st.param, src=0, group=self.pg)      # one broadcast of the flat buffer
            st.sync_shadow(force=True)
        self.bytes_reduced = 0

    # gradients are summed over ranks; the 1/world factor is folded into the Adam kernel
    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    @contextlib.contextmanager
    def no_sync(self):
        """skip the all-reduce (gradient-accumulation micro-batches before the last one)"""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def merge_buckets(self, groups: int) -> None:
        """Merge the model's buckets (embedding | layer 0 .. N-1 | fc, contiguous in the flat gradient buffer in that order) into
        ``groups`` contiguous runs of about equal byte size.  A group's all-reduce is issued when its LAST member reports ready;
        backward completes the members from the top of the buffer down (fc, layer N-1, ..., layer 0, embedding), so the top group
        goes out in the middle of backward and the bottom one at its end."""
        st = self.model.store()
        names = [b[0] for b in st.buckets]
        if groups is None or groups >= len(names):
            self._groups, self._group_of = None, {}
            return
        if groups < 1:
            raise ValueError("merge_buckets: at least one group")
        total = st.buckets[-1][2] - st.buckets[0][1]
        out, cur, start = [], [], st.buckets[0][1]
        for i, (bname, lo, hi) in enumerate(st.buckets):
            cur.append(bname)
            left_names, left_groups = len(names) - i - 1, groups - len(out) - 1
            if left_groups == 0:
                continue
            if (hi - st.buckets[0][1]) * groups >= total * (len(out) + 1) or left_names == left_groups:
                out.append(("+".join(cur), start, hi, tuple(cur)))
                cur, start = [], hi
        out.append(("+".join(cur), start, st.buckets[-1][2], tuple(cur)))
        self._groups = out
        self._group_of = {n: gi for gi, g in enumerate(out) for n in g[3]}
        self._arrived = {}

    def bucket_names(self) -> List[str]:
        """the all-reduce units of a step, in flat-buffer order (merged groups when ``merge_buckets`` is in effect)"""
        return [g[0] for g in self._groups] if self._groups else [b[0] for b in self.model.store().buckets]

    def bucket_ready(self, name: str):
        """called from inside backward when every gradient of bucket ``name`` has been accumulated (from the context of the
        stream that wrote the last of them: the collective is ordered behind that stream)"""
        if (self.world == 1 and not self.force) or not self._sync:
            return
        if self._groups is not None:
            gi = self._group_of[name]
            self._arrived[gi] = self._arrived.get(gi, 0) + 1
            if self._arrived[gi] < len(self._groups[gi][3]):
                return
            self._arrived[gi] = 0
            name = self._groups[gi][0]
        if not self.overlap:
            self._pending.append(name)
            return
        self._issue(name)

    def _slices(self):
        if self._groups is not None:
            return [(g[0], g[1], g[2]) for g in self._groups]
        return self.model.store().buckets

    def _issue(self, name: str):
        st = self.model.store()
        for bname, lo, hi in self._slices():
            if bname == name:
                t = st.grad[lo:hi]
                timed = self.measure_overlap
                nccl = timed and t.is_cuda and dist.get_backend(self.pg) == "nccl"
                if nccl:
                    # the issuing stream, now: with overlap the moment the bucket's gradients are complete; in no-overlap mode
                    # _issue runs from wait_all(), i.e. after the whole backward (the bucket has been complete for a while)
                    ev0 = torch.cuda.Event(enable_timing=True)
                    ev0.record()
                elif timed:
                    t0 = time.perf_counter()
                w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                if nccl:
                    # a side stream takes the collective's completion as a dependency and stamps it: the compute stream is not held
                    if self._mstream is None:
                        self._mstream = torch.cuda.Stream()
                    with torch.cuda.stream(self._mstream):
                        w.wait()
                        ev1 = torch.cuda.Event(enable_timing=True)
                        ev1.record()
                    pairs = self._bucket_t.setdefault(name, [])
                    pairs.append((ev0, ev1))
                    del pairs[:-self.MAX_TIMED]
                elif timed:
                    self._host_t0[id(w)] = (name, t0)       # gloo: host clock, closed in wait_all (wait() blocks the host there)
                self._works.append(w)
                self.bytes_reduced += t.numel() * 4
                return
        raise KeyError(name)

    def wait_all(self):
        """make the compute stream wait for every outstanding bucket (RCCL: a stream dependency, not a host
        block).  With ``measure_overlap`` the stall of the compute stream is bracketed by two HIP events: the
        time between them is the part of the all-reduce that backward did NOT hide."""
        if not self._works and not self._pending:
            return
        ev = None
        if self.measure_overlap and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for name in self._pending:                        # no-overlap mode: everything is issued here, after backward
            self._issue(name)
        self._pending.clear()
        for w in self._works:
            w.wait()
            if id(w) in self._host_t0:
                name, t0 = self._host_t0.pop(id(w))
                secs = self._bucket_t.setdefault(name, [])
                secs.append(time.perf_counter() - t0)
                del secs[:-self.MAX_TIMED]
        self._works.clear()
        if ev is not None:
            ev[1].record()
            self._exposed.append(ev)

    def exposed_ms(self) -> Optional[float]:
        """mean exposed (un-overlapped) all-reduce time per optimiser step, from the recorded event pairs"""
        if not self._exposed:
            return None
        torch.cuda.synchronize()
        v = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed.clear()
        return sum(v) / len(v)

    def bucket_ms(self) -> Optional[Dict[str, float]]:
        """mean issue -> complete time of each bucket's all-reduce in ms (measure_overlap): with RCCL from a HIP event on the compute
        stream at issue to one on a side stream that depends on the collective; with gloo from the host clock (its wait blocks)."""
        if not self._bucket_t:
            return None
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        out = {}
        for name, v in self._bucket_t.items():
            ms = [(a[0].elapsed_time(a[1]) if isinstance(a, tuple) else 1e3 * a) for a in v]
            out[name] = sum(ms) / len(ms)
        self._bucket_t.clear()
        return out

    def loss_weight(self, n_local: torch.Tensor) -> torch.Tensor:
        """The reference divides the summed loss by the non-pad count of the batch it sees
        (criterion.py:58-60).  Each rank normalises by its LOCAL count n_r, gradients are summed over ranks and
        scaled by 1/world in the optimiser; multiplying rank r's loss by  n_r * world / sum_r n_r  makes the
        result the global-batch mean exactly (one 4-byte all-reduce; == 1 on pad-free data).  Device-side, no
        host sync."""
        if self.world == 1 and not self.force:
            return torch.ones((), dtype=torch.float32, device=n_local.device)
        tot = n_local.detach().to(torch.float32).clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.pg)
        return n_local.detach().to(torch.float32) * float(self.world) / tot.clamp_min(1.0)

    def describe(self) -> Dict:
        out = {"world": self.world, "backend": dist.get_backend(self.pg) if dist.is_initialized() else None}
        if out["backend"] == "nccl":
            try:
                out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:        # noqa: BLE001 -- informational only
                out["rccl_version"] = None
        return out

    def all_reduce_scalar_mean(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1 and not self.force:
            return t
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return t / self.world
