"""A training-step segment captured ONCE into a hipGraph and replayed as a differentiable op.

The launch-bound shapes of the reference (EPro-PnP-6DoF trains on 32 objects per step, `lib/train.py`) spend more time
in the Python / dispatcher path than on the GPU: ~25 launches of 5-150 us each.  The step has no host round trip, so the
whole segment -- `cost_fun.set_param` -> `monte_carlo_forward` -> loss -> backward to the layer inputs -- can be recorded
into a hipGraph (`torch.cuda.CUDAGraph` on ROCm) and replayed in one submission.  `GraphedLoss` packages that recipe as
a differentiable op; with new input tensors copied in on every call it takes 0.38 -> 0.18 ms at 32 x 512 points and
0.57 -> 0.22 ms at 600 x 128 (4-DoF, RSLM) (`tools/bench_graphed.py`).  It does not pay where the GPU is the bottleneck
(4096 x 512: the input copies cost more than the launches saved):

    layer = EProPnP6DoF(...)
    def segment(x3d, x2d, w2d, pose_gt):                  # any callable built from the epropnp API
        cost_fun.set_param(x2d.detach(), w2d)
        out = layer.monte_carlo_forward(x3d, x2d, w2d, camera, cost_fun, pose_init=pose_gt, force_init_solve=False)
        return monte_carlo_pose_loss(out[4], out[5]).mean(), out[0]          # loss first, then auxiliary outputs
    graphed = GraphedLoss(segment, (x3d, x2d, w2d, pose_gt), layers=[layer])
    loss, pose_opt = graphed(x3d, x2d, w2d, pose_gt)      # same shapes every call; loss.backward() reaches x3d, x2d, w2d

The replayed graph contains the segment's own backward: the gradients of the loss w.r.t. the inputs come out of the
replay, and the autograd node returned by `graphed(...)` only scales them by the incoming gradient of the loss (so
`backward()` has to run before the next call replays the graph again; it raises otherwise).  Shapes,
dtypes and everything the callable closes over (camera, cost function objects, solver settings) are frozen at capture;
per-call data must be passed as inputs.  Random draws stay fresh on every replay because the layers' Philox call
counters are moved to device memory first (`EProPnPBase.enable_graph_safe_rng`).  On tensors that are not on a HIP
device the callable simply runs eagerly (same results, no graph).
"""
import torch


class _Replay(torch.autograd.Function):

    @staticmethod
    def forward(ctx, runner, *inputs):
        runner._load(inputs)
        runner.graph.replay()
        runner.generation += 1
        # the gradients stay in the graph's static buffers until backward (no copy); a later replay would overwrite them
        ctx.runner, ctx.generation = runner, runner.generation
        ctx.set_materialize_grads(False)
        outs = tuple(o.clone() for o in runner.static_outputs)
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, grad_loss, *_):
        runner = ctx.runner
        if grad_loss is None:
            return (None,) * (1 + len(runner.static_inputs))
        if runner.generation != ctx.generation:
            raise RuntimeError('GraphedLoss: backward() of an earlier call after the graph was replayed again; its '
                               'gradients have been overwritten (call backward before the next call)')
        return (None,) + tuple(None if s.grad is None else s.grad * grad_loss for s in runner.static_inputs)


class GraphedLoss:
    """Capture `fn(*inputs) -> loss | (loss, aux...)` together with `loss.backward()` and replay it per call.

    fn              callable on tensors of fixed shape; must return a scalar loss first; further outputs are returned
                    detached (pose_opt, samples, ...)
    example_inputs  tensors of the shapes / dtypes / device of every later call; inputs with `requires_grad` receive
                    gradients through the returned loss
    layers          EProPnP layers used inside `fn`: their Philox counters move to device memory before capture
    warmup          eager runs on a side stream before capture (allocator and lazy-initialisation warm-up)
    """

    def __init__(self, fn, example_inputs, layers=(), warmup=3):
        self.fn = fn
        example_inputs = tuple(example_inputs)
        self.enabled = all(t.is_cuda for t in example_inputs) and len(example_inputs) > 0
        self.graph = None
        self.generation = 0
        if not self.enabled:
            return
        dev = example_inputs[0].device
        for layer in layers:
            if getattr(layer, 'rng_counter', None) is None:
                layer.enable_graph_safe_rng(dev)
        self.static_inputs = [t.detach().clone().requires_grad_(t.requires_grad) for t in example_inputs]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                self._run_once()
        torch.cuda.current_stream(dev).wait_stream(side)
        for s in self.static_inputs:
            s.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_outputs = self._run_once()

    def _run_once(self):
        for s in self.static_inputs:
            s.grad = None
        out = self.fn(*self.static_inputs)
        outs = tuple(out) if isinstance(out, (tuple, list)) else (out,)
        assert outs[0].dim() == 0, 'the first output of the captured callable must be a scalar loss'
        if outs[0].requires_grad:
            outs[0].backward()
        return tuple(o.detach() for o in outs)

    def _load(self, inputs):
        assert len(inputs) == len(self.static_inputs), 'same number of inputs as at capture'
        with torch.no_grad():
            for s, t in zip(self.static_inputs, inputs):
                if t.shape != s.shape or t.dtype != s.dtype or t.device != s.device:
                    raise ValueError(f'GraphedLoss was captured for {tuple(s.shape)} {s.dtype} on {s.device}, got '
                                     f'{tuple(t.shape)} {t.dtype} on {t.device}: shapes are frozen at capture')
                s.copy_(t)

    def __call__(self, *inputs):
        if not self.enabled:
            return self.fn(*inputs)
        outs = _Replay.apply(self, *inputs)
        return outs[0] if len(outs) == 1 else outs
