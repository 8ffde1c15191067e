"""hipGraph capture of a whole loss step (forward + backward) for launch-bound paths.

The risk-sensitive losses are ~35 launches of microsecond kernels on `[B, S]`-sized tensors: at the batch
drivers' B = 100 the step is bound by launch latency, not by any kernel.  `GraphedLoss` records the forward AND the backward of
such a loss once, on static buffers, into one hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture / hipGraphLaunch;
every kernel of this package is launched on torch's current stream, so the ctypes launches are captured like ATen's) and replays it
per step: same kernels, same arithmetic, bit-identical results, one graph launch.

    step = GraphedLoss(lambda yp, yt, yb: geoRiskLambdaLoss(yp, yt, yb, listnet_transformation=2), (y_pred, y_true, y_base))
    loss, (dy_pred,) = step(y_pred, y_true, y_base)        # tensors of the captured shapes / dtypes / device

Shapes are fixed at capture (the drivers' minibatches are equal-sized except the last one: keep an eager call for that one).
Not for losses that draw host-side dropout seeds per call (they would replay one mask).

`GraphedTrainStep` does the same for a whole training step of a scorer (zero_grad + forward + loss + backward + optimizer):
the transformer scorer's step is ~450 launches, most of them microseconds long, and below ~128 slates per step it is the
dispatch of those launches that bounds it.  Its dropout seeds are kernel ARGUMENTS and freeze at capture, so the recorded
sequence begins with `ltr_enc_seed_advance`: a one-thread kernel that bumps the device-side epoch every dropout site adds to
its seed (include/ltr_encoder.h) -- replay k draws the masks of (seed + k), forward and backward of one replay the same ones,
exactly what an eager step with host seed (seed + k) draws (tests/test_graph_step_gpu.py)."""
import torch


class GraphedLoss:
    def __init__(self, fn, example_inputs, grad_inputs=(0,), warmup=3):
        if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in example_inputs):
            raise ValueError("GraphedLoss captures device tensors only")
        self.fn = fn
        self.grad_inputs = tuple(grad_inputs)
        self.static = [t.detach().clone() for t in example_inputs]
        for i in self.grad_inputs:
            self.static[i].requires_grad_(True)
        dev = self.static[0].device
        # warm-up on a side stream: first calls set per-device kernel attributes and fill the allocator's pools -- neither
        # may happen inside a capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                loss = fn(*self.static)
                torch.autograd.grad(loss.sum(), [self.static[i] for i in self.grad_inputs])
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = fn(*self.static)
            self.grads = torch.autograd.grad(self.loss.sum(), [self.static[i] for i in self.grad_inputs])

    def __call__(self, *inputs):
        """Replay on new input VALUES (same shapes).  Returns (loss, grads): tensors owned by the graph -- overwritten by the
        next call, clone them to keep them."""
        if len(inputs) != len(self.static):
            raise ValueError(f"expected {len(self.static)} inputs, got {len(inputs)}")
        with torch.no_grad():
            for dst, src in zip(self.static, inputs):
                if tuple(dst.shape) != tuple(src.shape) or dst.dtype != src.dtype:
                    raise ValueError(f"captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        self.graph.replay()
        return self.loss, self.grads


class GraphedTrainStep:
    """One hipGraph for `opt.zero_grad(); loss = loss_fn(net, *inputs); loss.backward(); opt.step()`.

        opt  = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True)     # capturable: its step counter lives on the device
        step = GraphedTrainStep(net, opt, lambda net, x, mask, y: approxNDCGLoss(net(x, mask, None), y), (x, mask, y))
        loss = step(x, mask, y)            # a tensor owned by the graph (clone to keep); the parameters are updated in place

    The warm-up steps are REAL optimizer steps on the example inputs (as in torch's own whole-network capture recipe; the
    capture itself only records): pass a real minibatch.  Reference: the minibatch loop of main_batch_execution.py:119-171."""

    def __init__(self, net, opt, loss_fn, example_inputs, warmup=3, advance_seed=True):
        if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in example_inputs):
            raise ValueError("GraphedTrainStep captures device tensors only")
        if not all(g.get("capturable", False) for g in opt.param_groups):
            raise ValueError("the optimizer must be built with capturable=True (its step counter has to live on the device)")
        from . import encoder as E
        self.net, self.opt, self.loss_fn = net, opt, loss_fn
        self.static = [t.detach().clone() for t in example_inputs]
        dev = self.static[0].device

        def one_step():
            if advance_seed:
                E.seed_advance(1)
            loss = loss_fn(net, *self.static)
            loss.backward()
            opt.step()
            return loss

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                opt.zero_grad(set_to_none=True)
                one_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)          # the captured backward then ALLOCATES .grad in the graph's pool: replays overwrite it
        with torch.cuda.graph(self.graph):
            self.loss = one_step()

    def __call__(self, *inputs):
        if len(inputs) != len(self.static):
            raise ValueError(f"expected {len(self.static)} inputs, got {len(inputs)}")
        with torch.no_grad():
            for dst, src in zip(self.static, inputs):
                if tuple(dst.shape) != tuple(src.shape) or dst.dtype != src.dtype:
                    raise ValueError(f"captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        self.graph.replay()
        return self.loss
