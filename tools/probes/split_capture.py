"""Round 6 probe, NOT part of the library: a training step captured as several single-stream hipGraphs with the weight gradients'
batches on a side stream (ops._queue_wgrad, MCQUIC_AMD_WGRAD_SIDE=1).  Measured on one MI355X (tools/bench_train.py --split against
--graph, 8 x 3 x 256 x 256, profiles/r06_wgrad_side_stream.txt): one graph 21.2 ms; the six graphs of this class all on one stream
21.65 ms (what cutting costs); with the batches beside the input gradients' chain 22.6 ms, 22.8 with the main stream at high
priority -- the chip-filling row walks hold every wave slot, and each launch of the latency-bound chain over the 16x16 ... 4x4 maps
waits for one to free.  The overlap loses; the step stays ONE graph on one stream."""
import os

import torch


class SplitCapture:
    """A training step captured as SEVERAL single-stream hipGraphs, cut where the weight gradients leave the main stream.

    Inside `autograd.backward` nothing reads a weight gradient before the pass ends, so their launches are queued and issued in
    batches (ops._queue_wgrad): the decoder's row walks when the pass reaches the few-pixel maps of the quantizer, the few-pixel
    ones when it reaches the encoder.  A batch that runs BESIDE the chain of input gradients fills the chip where that chain
    only waits for launch latencies -- but as a second branch inside ONE hipGraph it costs more than it gives (ROCm 7.2: the
    step replays in 37 ms instead of 21; every node of a multi-branch graph pays a cross-queue handshake).  So each fork ends the
    main stream's graph, the batch is captured as a graph of its own on the side stream, and a new main graph begins:

        main   [M0: forward, loss, decoder backward] [M1: quantizer backward      ] [M2: encoder backward       ] [F: the rest, reduce passes, ...]
        side                                         [W0: decoder weight gradients] [W1: few-pixel weight grads ]
                                                     ^ side waits for M0                                         ^ main waits for the side stream

    `replay()` enqueues the graphs in that order with one event per arrow.  All graphs share one memory pool; what a side graph
    reads is held until the join (ops._defer["keep"]), so the main graphs cannot be handed those blocks.

        cap = SplitCapture(device)
        with torch.cuda.stream(cap.main): warm_up()          # (gradient accumulators remember the stream they were created on)
        with cap:                                             # torch's current stream is cap.main inside
            loss = step()                                     # forward + autograd.backward(loss) [+ update]
        cap.replay()

    Cuts are made from the autograd engine's thread (the capture follows the stream, not the thread): relaxed capture mode."""

    def __init__(self, device):
        self.dev = torch.device(device)
        self.main = torch.cuda.Stream(self.dev, priority=int(os.environ.get("MCQUIC_AMD_SPLIT_MAIN_PRIO", "0")))
        self.side = torch.cuda.Stream(self.dev, priority=0)
        self.pool = torch.cuda.graph_pool_handle()
        self.program: list = []                               # ("main" | "side", graph) | ("fork", None) | ("join", None)
        self._g = None

    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
        self._g.capture_begin(pool=self.pool, capture_error_mode="relaxed")

    def _end(self, where: str):
        g, self._g = self._g, None
        g.capture_end()
        self.program.append((where, g))

    def __enter__(self):
        from mcquic_amd import ops
        if self.program:
            raise RuntimeError("SplitCapture: already captured")
        torch.cuda.synchronize(self.dev)
        self._outer = torch.cuda.current_stream(self.dev)
        self.main.wait_stream(self._outer)
        self._ctx = torch.cuda.stream(self.main)
        self._ctx.__enter__()
        ops._defer["split"] = self
        self._begin()
        return self

    def __exit__(self, exc_type, exc, tb):
        from mcquic_amd import ops
        ops._defer["split"] = None
        try:
            if self._g is not None:
                self._end("main")
        finally:
            self._ctx.__exit__(exc_type, exc, tb)
        if exc_type is None:
            self._outer.wait_stream(self.main)
        return False

    # ---- called by ops._issue_wgrads / ops._join_wgrad_side while capturing (the autograd thread's current stream is self.main) ----
    def fork(self, queue) -> None:
        self._end("main")
        self.program.append(("fork", None))
        with torch.cuda.stream(self.side):
            self._begin()
            try:
                for launch in queue:
                    launch()
            finally:
                self._end("side")
        self._begin()

    def join(self) -> None:
        self._end("main")
        self.program.append(("join", None))
        self._begin()

    def replay(self) -> None:
        outer = torch.cuda.current_stream(self.dev)
        self.main.wait_stream(outer)
        serial = os.environ.get("MCQUIC_AMD_SPLIT_SERIAL") == "1"
        for what, g in self.program:
            if what == "main" or (serial and what == "side"):
                with torch.cuda.stream(self.main):
                    g.replay()
            elif what == "side":
                with torch.cuda.stream(self.side):
                    g.replay()
            elif what == "fork":
                self.side.wait_stream(self.main)
            else:
                self.main.wait_stream(self.side)
        outer.wait_stream(self.main)
