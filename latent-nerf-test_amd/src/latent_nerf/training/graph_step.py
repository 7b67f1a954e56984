"""Whole-step hipGraph capture: render -> backward -> (gradient sync) -> optimiser as one replayable
graph, so the host cost of a step is one graph launch instead of ~40 Python/ctypes/autograd calls.

Every C-ABI entry point only enqueues work on the current stream and never synchronises or allocates
(include/lnerf_hip.h), data-dependent sizes stay on the device, and the optimiser keeps its step counter
on the device (`FusedAdam(capturable=True)`), so the step is capturable as is.  With more than one rank
the exchange is captured too where the backend's collectives can be (RCCL: `sync_in_graph=True`, one graph
launch per step and rank); otherwise (gloo stages through the host) it stays OUTSIDE the graphs: graph A
(render + backward), eager collectives on the static gradient tensors, graph B / eager optimiser."""
import torch

# hipStreamCaptureModeThreadLocal: only the capturing thread is restricted.  Under the default (global) mode RCCL's
# process-group watchdog thread, which polls the events of outstanding collectives, turns a capture into
# hipErrorStreamCaptureUnsupported whenever its poll lands inside one.
CAPTURE_MODE = "thread_local"


class GraphedTrainStep:
    def __init__(self, fwd_bwd, opt_step, params, sync=None, world=1, warmup=3, stream=None, opt_in_graph=True,
                 steps_per_graph=1, sync_in_graph=False):
        """fwd_bwd() -> dict of output tensors (leaves `.grad` set on `params`);
        opt_step() consumes the gradients; sync() all-reduces `.grad` in place (world > 1).

        Run the whole training loop (eager steps included) on ONE non-default stream and pass it as
        `stream` (or make it current): autograd pins each parameter's gradient accumulation to the stream
        it first ran on, and accumulation on the legacy default stream cannot be captured.

        opt_in_graph=False (world > 1, exchange outside the graphs, pipelined): the optimiser step stays eager -- it
        waits for one level group's eagerly launched all-reduce at a time (FusedAdam.step(row_groups=...)).

        sync_in_graph=True (world > 1, RCCL): sync() and opt_step() are captured behind fwd_bwd() in the SAME graph --
        the collectives' launches on RCCL's stream, the optimiser's waits on them and the hand-offs become graph edges.

        steps_per_graph (world == 1): that many whole steps per captured graph -- 2 for a `fwd_bwd` that alternates
        between two buffer sets (rays of step k+1 marched on a side stream while step k is shaded), whose pointers a
        one-step graph could not alternate.  `__call__` then runs that many steps; `self.steps_per_call` says so."""
        if steps_per_graph != 1 and world != 1:
            raise ValueError("steps_per_graph > 1 needs world == 1")
        self.steps_per_call = int(steps_per_graph)
        self.opt_step = opt_step
        self.world = world
        self.sync = sync
        self.params = list(params)
        if stream is None:
            stream = torch.cuda.current_stream()
            if stream == torch.cuda.default_stream():
                stream = torch.cuda.Stream()
        self.stream = stream
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for _ in range(warmup):
                fwd_bwd()
                if world > 1 and sync is not None:
                    sync()
                opt_step()
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        for p in self.params:  # gradients must be created inside the capture (graph pool), not accumulated into
            p.grad = None
        self.graph_a = torch.cuda.CUDAGraph()
        self.graph_b = None
        if world > 1 and sync_in_graph:
            with torch.cuda.graph(self.graph_a, stream=stream, capture_error_mode=CAPTURE_MODE):
                self.out = fwd_bwd()
                self.static_grads = [p.grad for p in self.params]
                if sync is not None:
                    sync()
                opt_step()
            self.graph_b = None
        elif world == 1:
            with torch.cuda.graph(self.graph_a, stream=stream, capture_error_mode=CAPTURE_MODE):
                for _ in range(self.steps_per_call):
                    self.out = fwd_bwd()
                    self.static_grads = [p.grad for p in self.params]  # graph-pool tensors, rewritten by every replay
                    opt_step()
                    if self.steps_per_call > 1:
                        for p in self.params:   # the next step's backward creates its gradients anew
                            p.grad = None
        else:
            with torch.cuda.graph(self.graph_a, stream=stream, capture_error_mode=CAPTURE_MODE):
                self.out = fwd_bwd()
            self.static_grads = [p.grad for p in self.params]
            if opt_in_graph:
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), stream=stream, capture_error_mode=CAPTURE_MODE):
                    opt_step()
            else:
                self.graph_b = False
        for p in self.params:
            p.grad = None

    def __call__(self):
        self.graph_a.replay()
        if self.graph_b is not None:
            for p, g in zip(self.params, self.static_grads):
                p.grad = g
            if self.sync is not None:
                self.sync()
            if self.graph_b is False:
                self.opt_step()
            else:
                self.graph_b.replay()
            for p in self.params:
                p.grad = None
        return self.out


class GraphedRenderStep:
    """The trainer's step as TWO captured graphs around the guidance call (the reference's loop,
    src/latent_paint/training/trainer.py:113-144 with the explicit backward of
    src/latent_paint_mesh/training/trainer.py:657-658):

        graph F   forward()            render of the view whose pose / intrinsics sit in static device buffers -> `pred`
        eager     grad = guidance.train_step(text_z, pred)  (any Python: a UNet, a stub), copied into a static buffer
        graph B   backward(out, pred)  injects the static gradient, back-propagates, (world == 1) steps the optimiser

    Both graphs share one memory pool: the autograd graph that F's capture built (its saved tensors are the static
    activations) is what B's capture walks, exactly once; replays re-run the same launches on the same addresses.
    Nothing is executed while capturing, so a capture does not advance the training state; host-side counters that the
    captured Python incremented are the caller's to restore.

    forward() -> (out dict, pred tensor); backward(out, pred) -> None."""

    def __init__(self, forward, backward, params, stream):
        self.stream = stream
        self.params = list(params)
        for p in self.params:   # gradients must be created inside the capture (graph pool), not accumulated into
            p.grad = None
        torch.cuda.synchronize()
        self.graph_f = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_f, stream=stream, capture_error_mode=CAPTURE_MODE):
            self.out, self.pred = forward()
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, pool=self.graph_f.pool(), stream=stream, capture_error_mode=CAPTURE_MODE):
            backward(self.out, self.pred)
        self.static_grads = [p.grad for p in self.params]   # graph-pool tensors (None where the step consumed them itself)
        for p in self.params:
            p.grad = None

    def forward(self):
        self.graph_f.replay()
        return self.out, self.pred

    def backward(self):
        self.graph_b.replay()
        return self.static_grads


class GraphedWholeStep:
    """One captured graph for a step whose guidance is itself capturable (device ops on static shapes, device-side RNG:
    the seeded synthetic guidance): render -> guidance -> backward -> optimiser, ONE graph launch per step.  `fn()` runs
    the whole step and returns (out dict, pred)."""

    def __init__(self, fn, params, stream):
        self.stream = stream
        self.params = list(params)
        for p in self.params:
            p.grad = None
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=stream, capture_error_mode=CAPTURE_MODE):
            self.out, self.pred = fn()
        self.static_grads = [p.grad for p in self.params]
        for p in self.params:
            p.grad = None

    def replay(self):
        self.graph.replay()
        return self.static_grads
