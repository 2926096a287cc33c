"""hipGraph capture of launch-bound sub-pipelines (tracker recurrence, refiner, decoder layers).

The referring tracker issues ~100 tiny kernels per frame, strictly in sequence; launched eagerly each costs
~15 us of host time, 50 ms per 30-frame clip, all of it serial under frame sharding.  Captured once per shape
into a hipGraph (torch.cuda.CUDAGraph = hipGraph on ROCm; the C-ABI kernels are captured too because they are
enqueued on torch's current stream), the same work replays with ~2 us per node and no Python in the loop.
Inputs are copied into static buffers, outputs are returned as views of static buffers (valid until the next replay
of the same graph — callers consume them within the clip or clone).
"""
import collections

import torch


class GraphRunner:
    """max_entries bounds the cache (least recently used graph dropped first): every distinct clip length costs two
    eager warm-ups plus a capture and keeps a hipGraph with its private memory pool alive, so an evaluation over
    videos of many different lengths must not accumulate them."""

    def __init__(self, fn, enabled=True, max_entries=8):
        self.fn, self.enabled, self.max_entries = fn, enabled, max_entries
        self._cache = collections.OrderedDict()

    def clear(self):
        self._cache.clear()

    def __call__(self, key, *tensors):
        """key: hashable that fixes shapes / control flow of fn(*tensors).  Tensors must be GPU tensors."""
        if not (self.enabled and tensors and tensors[0].is_cuda):
            return self.fn(*tensors)
        key = (key, tuple((tuple(t.shape), t.dtype, t.device) for t in tensors))
        entry = self._cache.get(key)
        if entry is None:
            static_in = [t.clone() for t in tensors]
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):          # warm-up outside capture: allocator pools, BLAS handles, heuristics
                for _ in range(2):
                    self.fn(*static_in)
            cur.wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # thread_local: other threads (e.g. the RCCL watchdog) may keep querying events while we capture
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                static_out = self.fn(*static_in)
            entry = (graph, static_in, static_out)
            self._cache[key] = entry
            while len(self._cache) > self.max_entries:
                self._cache.popitem(last=False)
        else:
            self._cache.move_to_end(key)
        graph, static_in, static_out = entry
        for s, t in zip(static_in, tensors):
            s.copy_(t)
        graph.replay()
        return static_out


class FusedKV:
    """K / V in-projection weights of a stack of cross-attention layers, concatenated once so that all layers' (and all
    frames') projections are ONE GEMM.  The concatenated tensors are referenced by captured hipGraphs, so they are
    allocated once per device and refreshed IN PLACE when a parameter changes (load_state_dict / optimizer step bump
    the parameters' version counters): a graph captured earlier keeps reading valid, current weights."""

    def __init__(self, rows="kv"):
        """rows: "kv" = the K and V rows [C, 3C) of every layer's in_proj, layer by layer (K0 V0 K1 V1 ...: K / V of all
        layers, one GEMM over the memory);
        "k_v" = the same rows as ALL layers' K, then all layers' V (K0 .. K5 V0 .. V5): layer l's head h is then head
        8 l + h of ONE attention call over layers x heads (the tracker's per-frame cross-attentions, tracker.py:293-318);
        "q" = the Q rows [0, C) (all layers' queries of ONE input: the tracker's per-frame reference, tracker.py:278);
        "out" = the out_proj weights stacked (layers, C, C) / biases (layers, C) — one batched GEMM (dvis_gemm_nt_bb)."""
        self._key, self._W, self._b = None, None, None
        self._rows = rows

    def get(self, layers, C):
        if self._rows == "out":
            ws = [l.multihead_attn.out_proj.weight for l in layers]
            bs = [l.multihead_attn.out_proj.bias for l in layers]
        else:
            ws = [l.multihead_attn.in_proj_weight for l in layers]
            bs = [l.multihead_attn.in_proj_bias for l in layers]
        ver = tuple(t._version for t in ws + bs) + tuple(t.data_ptr() for t in ws + bs)
        dev = ws[0].device
        sl = slice(C, None) if self._rows == "kv" else slice(0, C)
        if self._key != (ver, dev):
            if self._rows == "out":
                W = torch.stack([w.detach() for w in ws], 0)
                b = torch.stack([x.detach() for x in bs], 0)
            elif self._rows == "k_v":
                W = torch.cat([w[C:2 * C].detach() for w in ws] + [w[2 * C:].detach() for w in ws], 0)
                b = torch.cat([x[C:2 * C].detach() for x in bs] + [x[2 * C:].detach() for x in bs], 0)
            else:
                W = torch.cat([w[sl].detach() for w in ws], 0)
                b = torch.cat([x[sl].detach() for x in bs], 0)
            if self._W is not None and self._W.device == dev and self._W.shape == W.shape and self._W.dtype == W.dtype:
                self._W.copy_(W)
                self._b.copy_(b)
            else:
                self._W, self._b = W.contiguous(), b.contiguous()
            self._key = (ver, dev)
        return self._W, self._b


class ConvAsGemm:
    """nn.Conv1d weights (Co, Ci, k) re-laid as the (Co, k * Ci) matrix of the im2col GEMM (column = tap * Ci + channel),
    cached per module and — like FusedKV — refreshed IN PLACE when the parameter changes, because captured hipGraphs
    point at the cached tensor."""

    def __init__(self):
        self._cache = {}

    def get(self, conv):
        w = conv.weight
        key = (w._version, w.data_ptr(), w.device)
        ent = self._cache.get(id(conv))
        if ent is None or ent[0] != key:
            W = w.detach().permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()
            if ent is not None and ent[1].device == W.device and ent[1].shape == W.shape:
                ent[1].copy_(W)
                W = ent[1]
            self._cache[id(conv)] = ent = (key, W)
        return ent[1]
