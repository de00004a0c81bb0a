"""Whole-step HIP graph: capture one training step (forward, backward, optimizer) once and replay it.

The config-5 step is ~730 kernel launches of 5-500 us (16 UNet layers, projection, IF-Net) and is HOST bound when every
launch goes through Python: 30 ms per step for ~20 ms of GPU work.  All entry points of libsvr_hip.so only enqueue on
the stream they are given (no allocation, no synchronisation: include/svr_hip.h), the side-stream sorts are forked and
joined with events, so the step can be captured by torch.cuda.graph as it is.  Requirements on the caller: static shapes,
an optimizer created with capturable=True, and no host reads of device values inside training_step (the mesh-labelling
branch subsample_points != 0 reads a `holes.any()` flag and cannot be captured)."""
import torch


class GraphedStep:
    def __init__(self, trainer, optimizer, example_batch, warmup=3):
        if warmup < 1:
            # the optimizer creates its state (zero-filled moments, step counter) lazily in its first step(): inside the
            # capture those fills would become part of the graph and every replay would reset the state
            raise ValueError("GraphedStep needs at least one eager warm-up step before the capture")
        self.trainer, self.optimizer = trainer, optimizer
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                                   # eager warm-up off the default stream
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._step()

    def _step(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.trainer.training_step(self.static, 0)["loss"]
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def run(self, batch):
        """Copy the batch into the captured step's input buffers and replay; returns the (device) loss of that step."""
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        return self.loss
