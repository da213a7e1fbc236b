"""The training step as ONE captured hipGraph (HIP graphs instead of a tracing compiler: the step's ~25 launches — two fused
MLP forwards, compositing, resampling, loss, the merged backward, Adam, weight packing and the handful of ATen glue kernels —
are recorded once and replayed with a single launch).  What changes from step to step stays outside the recording:

  * the batch: `GraphedStep.__call__` copies the caller's tensors into static input buffers;
  * Adam's scalars (step count, bias corrections, the decayed lr of R:784-788): `FusedAdam.make_capturable()` moves them to
    device memory, the recording holds the 32-byte copy from a pinned host buffer, `FusedAdam.advance()` rewrites that buffer
    before every replay;
  * randomness: torch's CUDA generator is graph-safe (philox offsets are advanced per replay).

Everything inside is exactly the eager step (same kernels, same order).  The gain is launch latency only — the four MFMA
kernels are 98.9 % of the step — so this is an option, not a requirement: `GraphedStep(...)` raises if the capture fails and
the caller keeps stepping eagerly."""
from typing import Callable, Sequence

import torch


class GraphedStep:
    def __init__(self, step_fn: Callable[..., torch.Tensor], optimizer, example_inputs: Sequence[torch.Tensor], warmup: int = 3):
        """step_fn(*inputs) -> loss runs ONE full step (render, loss, zero_grad, backward, optimizer.step()) on tensors of
        the shapes of `example_inputs`; it must not read host-side state that changes between steps.  The `warmup` runs
        before the recording are REAL steps on `example_inputs` (they move the weights and Adam's state, like `warmup`
        ordinary training steps on that batch); the recording itself executes nothing."""
        self.opt = optimizer.make_capturable()
        self.static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):             # warm-up on a side stream (allocator pools, lazy initialisations)
            for _ in range(warmup):
                step_fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):        # (recorded, not executed: the step counter does not move here)
            self.static_loss = step_fn(*self.static_in)
        torch.cuda.synchronize()

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src)
        self.opt.advance()
        self.graph.replay()
        return self.static_loss
