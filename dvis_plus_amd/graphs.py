"""hipGraph capture of launch-bound sub-pipelines (tracker recurrence, refiner, decoder layers).

The referring tracker issues ~100 tiny kernels per frame, strictly in sequence; launched eagerly each costs
~15 us of host time, 50 ms per 30-frame clip, all of it serial under frame sharding.  Captured once per shape
into a hipGraph (torch.cuda.CUDAGraph = hipGraph on ROCm; the C-ABI kernels are captured too because they are
enqueued on torch's current stream), the same work replays with ~2 us per node and no Python in the loop.
Inputs are copied into static buffers, outputs are returned as views of static buffers (valid until the next replay
of the same graph — callers consume them within the clip or clone).
"""
import torch


class GraphRunner:
    def __init__(self, fn, enabled=True):
        self.fn, self.enabled, self._cache = fn, enabled, {}

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
        graph, static_in, static_out = entry
        for s, t in zip(static_in, tensors):
            s.copy_(t)
        graph.replay()
        return static_out
