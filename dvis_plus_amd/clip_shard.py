"""Frame sharding of one clip over the GPUs of a node — SURVEY.md §8(e).

The segmenter treats frames as a batch, so rank r runs backbone + pixel decoder + masked-attention decoder on a
contiguous block of ceil(T/G) frames and keeps their mask_features locally (never communicated).  The tracker is a
recurrence over t and the refiner attends over all T, so both need every frame's queries: ONE all-gather of the
packed per-frame queries (pred_embds | pred_embds_without_norm | pred_logits = 4*C + K + 1 floats per query,
459.6 KB per frame at Q=100, C=256, K=124) over RCCL/xGMI (gloo in the CPU tests); tracker + refiner then run
replicated and deterministic on every rank, and each rank contracts masks for its own frames only.
The reference has no intra-video parallelism at all — this is a new capability, not a port.
"""
import os

import torch
import torch.distributed as dist


class ClipShard:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # development aid: run every collective even on a single rank (exercises RCCL calls, streams and the interplay
        # with hipGraph capture on a 1-GPU box)
        self.force = os.environ.get("DVIS_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized()

    def frames_per_rank(self, T):
        return (T + self.world - 1) // self.world

    def local_range(self, T, shift=0):
        """Frames [lo, hi) of this rank: block (rank - shift) mod world of ceil(T/world) frames.  `shift` rotates which
        rank gets which block — stream() advances it clip by clip so that the short (or empty) last blocks of a ragged
        split (T=30 over 8 ranks: 4,4,4,4,4,4,4,2) do not always land on the same rank."""
        per = self.frames_per_rank(T)
        lo = min(T, ((self.rank - shift) % self.world) * per)
        return lo, min(T, lo + per)

    def round_plan(self, T, rounds=1):
        """Pipelined schedule: the clip is cut into `rounds` consecutive spans of world*k frames; inside a span rank r
        owns the k frames [start + r*k, start + (r+1)*k).  After round c every rank holds frames [0, end_c) in order, so
        the (sequential) tracker can consume round c while the segmenter is busy with round c+1.  rounds=1 is the
        plain contiguous sharding.  Returns [(start, end, lo, hi)] with [lo, hi) this rank's frames of the span."""
        rounds = max(1, min(int(rounds), (T + self.world - 1) // self.world)) if T else 1
        k = max(1, (T + self.world * rounds - 1) // (self.world * rounds))
        plan = []
        for start in range(0, max(T, 1), self.world * k):
            end = min(T, start + self.world * k)
            lo = min(end, start + self.rank * k)
            plan.append((start, end, lo, min(end, lo + k)))
        return plan, k

    def all_gather_frames(self, parts, T, per=None, async_op=False, shift=0, guard=None):
        """parts: list of (t_local, Q, c_i) tensors for this rank's frames.  Returns the same list with all T frames,
        identical on every rank.  One collective on one packed buffer.  `per` = slot size per rank (default
        ceil(T / world)); with async_op the result is (tensors, work) and the caller waits on `work` in the stream that
        consumes the tensors.
        guard: this rank's range-guard word (int32 device tensor, functions._X3RangeGuard) or None.  It rides in the SAME
        collective — four more floats per query row, the tag in the first — and `self.last_guard_tags` then holds every rank's
        tag (a (world,) view of the gathered buffer, valid once the collective is complete): the decision to raise / re-run is
        taken from the same data on every rank (a rank that raised alone would leave the others in the next collective)."""
        if self.world == 1 and not self.force:
            self.last_guard_tags = None
            return (parts, None) if async_op else parts
        per = per or self.frames_per_rank(T)
        widths = [p.shape[-1] for p in parts]
        Q = parts[0].shape[1]
        W = sum(widths) + (4 if guard is not None else 0)
        packed = torch.zeros((per, Q, W), dtype=parts[0].dtype, device=parts[0].device)
        t_local = parts[0].shape[0]
        if t_local:
            packed[:t_local, :, :sum(widths)] = torch.cat(parts, dim=-1)
        if guard is not None:
            packed[0, 0, W - 4] = guard.reshape(-1)[0].to(packed.dtype)          # (tags are small integers: exact in fp32)
        gathered = torch.empty((self.world * per, Q, W), dtype=packed.dtype, device=packed.device)
        work = dist.all_gather_into_tensor(gathered, packed, group=self.group, async_op=async_op)
        self.last_guard_tags = gathered.view(self.world, per, Q, W)[:, 0, 0, W - 4] if guard is not None else None
        if shift % self.world:                       # block b came from rank (b + shift) mod world: back to frame order
            assert not async_op, "the rotation reads the gathered buffer"
            gathered = gathered.view(self.world, per, Q, -1).roll(-(shift % self.world), 0).reshape(self.world * per, Q, -1)
        gathered = gathered[:T]                      # blocks are frame-contiguous: padding only sits at the very end
        out = list(gathered.split(widths + ([4] if guard is not None else []), dim=-1))[:len(widths)]
        return (out, work) if async_op else out

    def all_gather_rows(self, row):
        """row: flat tensor of the same length on every rank -> (world, len), row r from rank r."""
        if self.world == 1 and not self.force:
            return row.unsqueeze(0)
        out = torch.empty(self.world * row.numel(), dtype=row.dtype, device=row.device)
        dist.all_gather_into_tensor(out, row.contiguous(), group=self.group)
        return out.view(self.world, row.numel())

    def all_reduce_sum(self, x):
        if self.world > 1 or self.force:
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
        return x

    def all_reduce_max(self, x):
        if self.world > 1 or self.force:
            dist.all_reduce(x, op=dist.ReduceOp.MAX, group=self.group)
        return x

    def check_replicas(self, tensors, what):
        """Debug guard for REPLICATED results (DVIS_CHECK_REPLICAS=1; off by default: it adds a collective and a host
        sync per clip): the replicated tracker + refiner must produce the same BITS on every rank — the post-processing
        collectives are sized from decisions taken on them, so a divergence would surface as an RCCL hang, not an error.
        All-gathers a per-tensor checksum of the raw bit patterns and raises on the first rank that disagrees."""
        if not ((self.world > 1 or self.force) and os.environ.get("DVIS_CHECK_REPLICAS") == "1"):
            return
        sums = torch.stack([t.detach().contiguous().view(torch.int32).to(torch.int64).sum() for t in tensors if t is not None])
        rows = self.all_gather_rows(sums).cpu()
        bad = (rows != rows[0:1]).any(1).nonzero().flatten().tolist()
        if bad:
            raise RuntimeError(f"replicated {what} differ between rank 0 and rank(s) {bad}: checksums {rows.tolist()}")

    def broadcast_from_rank0(self, tensors):
        """Make small decision inputs bit-identical on every rank.  (Rounds 1-2 broadcast the replicated tracker's class
        logits because its library GEMMs could pick different algorithms per rank; since round 3 the tracker / refiner run
        on the own deterministic GEMM and the pipeline no longer calls this.)"""
        if self.world > 1 or self.force:
            for t in tensors:
                if t is not None:
                    dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                   group=self.group)
        return tensors


class EmulatedShard(ClipShard):
    """Measuring aid (tools/rank_emulation.py), not a product path: ONE process does exactly the work of rank `rank` of a
    `world`-rank job — its block of every clip's frames, the same packing, buffers and kernels — with every collective
    replaced by a device-local stand-in of the same size: the other ranks' slots of an all-gather are filled with copies of
    this rank's slot, an all-reduce leaves the tensor as it is.  The per-rank time of a schedule can so be measured on a
    1-GPU box; what it leaves out is the collective itself (RCCL latency + 15 - 30 MB over xGMI per clip) and the skew
    between ranks.  The outputs are NOT those of the real job (the other frames' queries are made up)."""

    def __init__(self, world, rank=0):
        self.group, self.world, self.rank, self.force = None, int(world), int(rank), False

    def all_gather_frames(self, parts, T, per=None, async_op=False, shift=0, guard=None):
        self.last_guard_tags = None if guard is None else guard.reshape(-1)[:1].to(torch.float32)
        per = per or self.frames_per_rank(T)
        widths = [p.shape[-1] for p in parts]
        Q = parts[0].shape[1]
        packed = torch.zeros((per, Q, sum(widths)), dtype=parts[0].dtype, device=parts[0].device)
        t_local = parts[0].shape[0]
        if t_local:
            packed[:t_local] = torch.cat(parts, dim=-1)
            packed[t_local:] = packed[:1]                                   # (a short block: made-up but finite rows)
        gathered = packed.unsqueeze(0).expand(self.world, -1, -1, -1).reshape(self.world * per, Q, -1).contiguous()[:T]
        out = list(gathered.split(widths, dim=-1))
        return (out, None) if async_op else out

    def all_gather_rows(self, row):
        return row.unsqueeze(0).expand(self.world, -1).contiguous()

    def all_reduce_sum(self, x):
        return x

    def all_reduce_max(self, x):
        return x

    def check_replicas(self, tensors, what):
        return

    def broadcast_from_rank0(self, tensors):
        return tensors
