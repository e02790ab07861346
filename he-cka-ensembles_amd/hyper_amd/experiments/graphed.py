"""A whole training step captured in ONE HIP graph.

The reference's step (hyper/experiments/training.py: forward of the generated ensemble, method loss, backward,
optimizer) is a few hundred small launches at the MNIST / toy sizes (a 5-member LeNet step is about 275 kernels of a
few microseconds): the GPU waits for the host.  Everything on that path here is stream-ordered and allocation goes
through torch's caching allocator, so the step can be captured once (torch.cuda.CUDAGraph = hipGraph on ROCm) and
replayed with one launch.  No counterpart in the reference.

Conditions (checked or documented):
  * fixed member ids (or a fixed NUMBER of sampled members: hypernetwork) and fixed batch shape (the inputs are copied
    into static buffers before every replay);
  * the method's warm-up factor is a host float baked into the capture: capture at a step >= `warmup`, or re-capture;
  * the optimizer must be capturable (torch.optim.AdamW(..., capturable=True); SGD is as it is);
  * nothing on the path may synchronise with the host (RepulsiveKernel masks a NaN repulsion with nan_to_num)."""
import types

import torch


class GraphedTrainStep:
    def __init__(self, hyper, method, optimizer, x, y, member_ids=None, step=10 ** 9, warmup_iters=3,
                 forward_kwargs=None, trainer=None):
        if not x.is_cuda:
            raise RuntimeError('GraphedTrainStep needs GPU tensors')
        self.hyper, self.method, self.optimizer = hyper, method, optimizer
        self.x, self.y = x.clone(), y.clone()
        # a list = those members every step; an int = that many members SAMPLED every step (hypernetwork latents are
        # drawn inside the capture: torch's generator is capture-aware, every replay draws fresh ones)
        self.ids = member_ids if isinstance(member_ids, int) else (
            list(member_ids) if member_ids is not None else list(range(method.num)))
        self.step_index = step
        self.kwargs = dict(sample_params=True, ret_params=True, feature_split=True, ood_N=0)
        self.kwargs.update(forward_kwargs or {})
        self.trainer = trainer if trainer is not None else types.SimpleNamespace(training=True)
        cur = torch.cuda.current_stream(x.device)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):              # real training steps on the first batch, off the capture stream
            for _ in range(warmup_iters):
                self._body()
        cur.wait_stream(side)
        torch.cuda.synchronize(x.device)
        optimizer.zero_grad(set_to_none=True)      # gradients are allocated inside the capture, from the graph's pool
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body(captured=True)

    def _body(self, captured=False):
        if not captured:      # inside the capture .grad is None at the start: backward ASSIGNS into the graph's buffers
            self.optimizer.zero_grad(set_to_none=True)
        track = self.hyper(params=self.ids, x=self.x, **self.kwargs)
        track = self.method(self.trainer, track['params'], [], track, self.x, self.y, step=self.step_index)
        track['loss'].backward()
        self.optimizer.step()
        return track['loss']

    def __call__(self, x=None, y=None):
        """One training step on (x, y) (shapes as at capture); returns the loss tensor (static: overwritten by the next call)."""
        if x is not None:
            self.x.copy_(x)
        if y is not None:
            self.y.copy_(y)
        self.graph.replay()
        return self.loss
