"""The training step as captured hipGraphs (HIP graphs instead of a tracing compiler: the step's ~25 launches — two fused MLP
forwards, compositing, resampling, loss, the merged backward, Adam, weight packing and the handful of ATen glue kernels — are
recorded once and replayed with a single launch).  What changes from step to step stays outside the recording:

  * the batch: `GraphedStep.__call__` copies the caller's tensors into static input buffers;
  * Adam's scalars (step count, bias corrections, the decayed lr of R:784-788, the 1/world of a summed gradient):
    `FusedAdam.make_capturable()` moves them to device memory; `FusedAdam.advance()` uploads the step's 32 bytes from a ring of
    pinned blocks, stream-ordered ahead of every replay (the upload is NOT part of the recording: a recorded copy would read
    whatever the host has written by the time the GPU gets there);
  * randomness: the in-kernel jitter / resampling streams (csrc/rng.hpp) read {seed, base offset} from device memory under a
    recording (ops.RngCapture); `__call__` uploads the device generator's current pair (16 bytes, same pinned-ring discipline as
    Adam's scalars) and advances the generator by what one replay consumes.  For a step whose only random numbers are these
    in-kernel streams and which calls render_rays ONCE (the C2 / C4 step: raw_noise_std = 0) a replayed step sees exactly the
    numbers the eager step in its place would have seen (tested).  Any torch.rand / torch.randn left in the step (raw_noise_std > 0)
    is graph-safe by itself — torch advances its philox offsets per replay — and its stream stays disjoint from the in-kernel
    ones, but the offsets interleave differently from eager stepping (the recording reserves all in-kernel offsets first, torch's
    whole-graph increment after): valid, independent numbers, NOT the eager step's numbers.

Data-parallel steps (a `distributed.GradReducer` is passed): the gradient exchange sits between the backward and Adam.
  collective="split"   (default) two graphs around it: [render, loss, backward] -> eager all-reduce of the flat gradient on the
                       collective's own stream -> [Adam].  The host enqueues all three back to back (it is milliseconds ahead of
                       the GPU), so the GPU sees no bubble; works with any backend (RCCL, or gloo in the one-GPU tests).
  collective="capture" the RCCL all-reduce is recorded INSIDE one graph (fork to RCCL's stream and join back are capturable);
                       one launch per step.  RCCL only.

Everything inside is exactly the eager step (same kernels, same order).  The gain is launch latency only — at 4096 rays per GPU
the four MFMA kernels are 98.9 % of the step; at the 512 rays per GPU of the 8-way strong-scaling shard (C4) the ~100 host
launches of an eager step are no longer hidden, and the graph is what keeps the step on the kernels' time."""
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from . import ops


class GraphedStep:
    def __init__(self, step_fn: Callable[..., torch.Tensor], optimizer, example_inputs: Sequence[torch.Tensor], warmup: int = 3,
                 reducer=None, collective: str = "split"):
        """Without `reducer`: step_fn(*inputs) -> loss runs ONE full step (render, loss, zero_grad, backward, optimizer.step()).
        With `reducer` (a distributed.GradReducer built with fold_scale=True): step_fn stops after loss.backward(); the exchange
        (`reducer.finish()`) and `optimizer.step(grad_scale=reducer.grad_scale)` are appended here.  step_fn must not read
        host-side state that changes between steps.  The `warmup` runs before the recording are REAL steps on `example_inputs`
        (they move the weights and Adam's state, like `warmup` ordinary training steps on that batch); the recording itself
        executes nothing."""
        if collective not in ("split", "capture"):
            raise ValueError("collective must be 'split' or 'capture'")
        self.opt = optimizer.make_capturable()
        self.reducer, self.collective = reducer, collective
        self.static_in = [t.clone() for t in example_inputs]
        self.tail_graph = None

        def tail():
            reducer.finish()
            optimizer.step(grad_scale=reducer.grad_scale)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):             # warm-up on a side stream (allocator pools, lazy initialisations)
            for _ in range(warmup):
                step_fn(*self.static_in)
                if reducer is not None:
                    tail()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # the weight re-pack must be part of the recording: a pack cached from a render just before (possible with warmup=0)
        # would leave every replay's forward on the weights as they were at capture time
        self.opt.bump_epoch()
        timing = getattr(reducer, "timing", False)
        self._hold_before = getattr(reducer, "hold", False)
        if reducer is not None:
            reducer.timing = False                # (timed events cannot be recorded into a graph)
            reducer.hold = collective == "split"  # split: nothing leaves from inside the (recorded) backward
        self.graph = torch.cuda.CUDAGraph()
        self._mse_counters = ops.prepare_capture(self.static_in[0].device)     # this recording's own ticket block
        self._rng = ops.RngCapture(self.static_in[0].device)
        self._rng_ring = [torch.zeros(2, dtype=torch.int64).pin_memory() for _ in range(4)]
        self._rng_ev, self._rng_i = [None] * 4, 0
        ops.RngCapture.active = self._rng
        ok = False
        if reducer is not None:
            self._msgs_before, self._steps_before = reducer.messages, reducer.steps
        try:
            # capture_error_mode: with a process group alive, c10d's watchdog THREAD polls the events of earlier collectives
            # (hipEventQuery).  Under the default "global" mode such a call from another thread while this thread is capturing is
            # an error (hipErrorStreamCaptureUnsupported) that terminates the process — reproduced in 1 of 12 recordings on a
            # 1-rank RCCL group (round 6).  "thread_local" restricts the check to the capturing thread.
            import torch.distributed as _dist
            mode = "thread_local" if (reducer is not None or (_dist.is_available() and _dist.is_initialized())) else "global"
            with torch.cuda.graph(self.graph, capture_error_mode=mode):    # (recorded, not executed: the step counter does not move here)
                self.static_loss = step_fn(*self.static_in)
                if reducer is not None and collective == "capture":
                    tail()
            if reducer is not None and collective == "split":
                reducer.reset()                   # the recorded backward counted its nodes; nothing was issued
                self.tail_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.tail_graph, pool=self.graph.pool(), capture_error_mode=mode):
                    optimizer.step(grad_scale=reducer.grad_scale)
            ok = True
        finally:
            ops.RngCapture.active = None
            ops.end_capture(self.static_in[0].device)
            if reducer is not None:
                reducer.timing = timing
                if not ok:
                    # a failed recording must leave the reducer as it found it: the caller's fallback is eager stepping, which
                    # needs the in-backward issue (hold off) and none of the recorded backward's bookkeeping
                    reducer.reset()
                    reducer.hold = self._hold_before
        if reducer is not None and collective == "capture":
            reducer.reset()
            # the recorded finish() counted one step and its messages: remembered so that every replay counts the same
            self._msgs_per_replay = reducer.messages - self._msgs_before
            reducer.messages, reducer.steps = self._msgs_before, self._steps_before     # (the recording itself sent nothing)
        torch.cuda.synchronize()

    def _advance_rng(self):
        """Hand the replay the device generator's current (seed, offset) and advance the generator by what a replay consumes."""
        if self._rng.used == 0:
            return
        dev = self._rng.state.device
        gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
        off = gen.get_offset()
        gen.set_offset(off + self._rng.used)
        i = self._rng_i = (self._rng_i + 1) % len(self._rng_ring)
        if self._rng_ev[i] is not None:
            self._rng_ev[i].synchronize()
        self._rng_ring[i].numpy().view(np.uint64)[:] = (gen.initial_seed() & 0xFFFFFFFFFFFFFFFF, off)
        self._rng.state.copy_(self._rng_ring[i], non_blocking=True)
        ev = self._rng_ev[i] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._rng_ev[i] = ev

    def release(self):
        """Give the reducer back to eager stepping (split mode sets `reducer.hold`, under which nothing leaves from inside
        loss.backward()): call before stepping eagerly again with the same GradReducer."""
        if self.reducer is not None:
            self.reducer.reset()
            self.reducer.hold = self._hold_before

    def timed_call(self, *inputs: torch.Tensor):
        """DIAGNOSTIC form of __call__: the same replay with a device synchronisation after every phase and the host's wall time of
        each — {inputs_ms: the static-input copies + Adam / RNG scalar uploads, graph_ms: the recorded [render, loss, backward
        (+ all-reduce + Adam when the collective is recorded inside)], exchange_ms: reducer.finish() of the split form (the eager
        all-reduce of the flat gradient: RCCL over xGMI, or gloo's host staging in the one-GPU tests), tail_ms: the recorded Adam} —
        so that a slow data-parallel graphed step can be read phase by phase (bench.py's c4_strong leg).  The synchronisations
        remove the host / device overlap the real __call__ has: the SUM is an upper bound of a step, the SPLIT is the information."""
        import time
        t = {}

        def lap(name, t0):
            torch.cuda.synchronize()
            t[name] = (time.perf_counter() - t0) * 1e3
            return time.perf_counter()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src)
        self.opt.advance(1.0 if self.reducer is None else self.reducer.grad_scale)
        self._advance_rng()
        t0 = lap("inputs_ms", t0)
        self.graph.replay()
        t0 = lap("graph_ms", t0)
        if self.tail_graph is not None:
            self.reducer.finish()
            t0 = lap("exchange_ms", t0)
            self.tail_graph.replay()
            t0 = lap("tail_ms", t0)
        elif self.reducer is not None:
            self.reducer.steps += 1
            self.reducer.messages += self._msgs_per_replay
        self.opt.bump_epoch()
        return self.static_loss, t

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src)
        self.opt.advance(1.0 if self.reducer is None else self.reducer.grad_scale)
        self._advance_rng()
        self.graph.replay()
        if self.tail_graph is not None:
            self.reducer.finish()                 # eager: every slice of the flat gradient, on the collective's stream
            self.tail_graph.replay()
        elif self.reducer is not None:            # capture: the recorded finish() does not run again — keep its statistics honest
            self.reducer.steps += 1
            self.reducer.messages += self._msgs_per_replay
        # the recorded Python ran once: tell the packed-weight caches that the weights moved (an eval render between graphed
        # steps must re-pack, not reuse the kernel-layout copy of an earlier evaluation)
        self.opt.bump_epoch()
        return self.static_loss
