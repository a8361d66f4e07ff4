"""Host-to-host streaming SR: frames arrive in host memory and leave in host memory (the boundary the reference's
harness has, eval_lut_sr.py:514-665).

Two transports behind one interface, both on pinned host buffers owned by the object:

* ``transport="dma"`` (default): the fused kernel reads the batch's pinned input itself and writes DEVICE memory; the result goes
  down on a copy stream with the runtime's copy engines while the next launch runs.  At most ONE download is in flight, and the
  next one is issued from the host the moment the previous one has landed: with several copies queued at once the runtime
  executes some of them as blit kernels, which take turns with the SR workgroups on every CU (0.76-0.83 ms per frame instead of
  0.46-0.50, with stalls of tens of milliseconds; profiles/r04_experiments.txt, experiment G).  The link is the limit: it moves
  56.8 GB/s in either direction but no more than that in both together on these hosts, and a 1080p -> 4K frame is 6.2 MB up +
  24.9 MB down = 0.548 ms; measured 0.574 ms per frame (14.2 Gpix/s), the same in every run.
* ``transport="zero_copy"``: the fused kernel reads its input tiles straight from the pinned host memory and writes its output
  rows straight to it (pointers of hipHostMalloc'ed buffers are valid on the device): no device buffers and no copy
  engines, the PCIe traffic rides inside the kernel; 0.70 ms per frame (the kernel's stores cross the link at 36 GB/s).

    st = StreamingSR(engine, (1080, 1920), 2, frames_per_batch=8)
    buf = st.input(slot)                 # uint8 [B,H,W,3] numpy VIEW of the slot's pinned input: decode straight into it
    for out in st.run(batches):          # or: an iterable of uint8 [B,H,W,3] arrays (copied into the pinned input)
        ...                              # out: uint8 [B,oH,oW,3] numpy VIEW of a pinned buffer, valid until the slot is reused
"""
from __future__ import annotations

import numpy as np

from . import _lib, ops


class StreamingSR:
    def __init__(self, engine, in_hw, scale, frames_per_batch=1, depth=None, transport="dma"):
        torch = _lib.require_gpu()
        if transport not in ("dma", "zero_copy"):
            raise ValueError("transport must be 'dma' or 'zero_copy'")
        self.transport = transport
        self.engine, self.B = engine, int(frames_per_batch)
        self.depth = int(depth) if depth else (3 if transport == "dma" else 2)
        self.geo = engine.sr_geometry(in_hw, scale)
        H, W = int(in_hw[0]), int(in_hw[1])
        oH, oW = self.geo.out_hw
        dev = engine.luts.device
        self.stream = torch.cuda.Stream(device=dev)
        self.down = torch.cuda.Stream(device=dev) if transport == "dma" else None
        nb = int(_lib.lib().lerf_sr_fused_workspace_bytes(H, W, 3, self.B))
        self.slots = []
        for _ in range(self.depth):
            s = dict(h_in=torch.empty((self.B, H, W, 3), dtype=torch.uint8).pin_memory(),
                     h_out=torch.empty((self.B, oH, oW, 3), dtype=torch.uint8).pin_memory(),
                     ws=torch.empty(max(1, nb), dtype=torch.uint8, device=dev),
                     done=torch.cuda.Event(), busy=False)
            if transport == "dma":
                s.update(d_out=torch.empty((self.B, oH, oW, 3), dtype=torch.uint8, device=dev), sr_done=torch.cuda.Event(), issued=False)
            self.slots.append(s)
        self._next = 0
        self._flying = None          # dma: the slot whose download is in flight
        self._waiting = []           # dma: launched slots whose download has not been issued yet, oldest first

    def _pump(self, block=False):
        """dma: retire the download in flight (waiting for it when `block`) and issue the next one."""
        torch = _lib.require_gpu()
        if self._flying is not None:
            f = self.slots[self._flying]
            if block:
                f["done"].synchronize()
            if f["done"].query():
                self._flying = None
        if self._flying is None and self._waiting:
            j = self._waiting.pop(0)
            s = self.slots[j]
            with torch.cuda.stream(self.down):
                self.down.wait_event(s["sr_done"])
                s["h_out"].copy_(s["d_out"], non_blocking=True)
                s["done"].record(self.down)
            s["issued"] = True
            self._flying = j

    def input(self, i=None):
        """numpy view of the pinned input buffer of slot i (default: the slot the next submit() uses)."""
        return self.slots[self._next if i is None else i]["h_in"].numpy()

    def submit(self, frames=None):
        """Launch one batch.  frames: uint8 [B,H,W,3] array / CPU tensor copied into the slot's pinned input, or None
        when the producer has already written input().  Returns the slot index to pass to result()."""
        torch = _lib.require_gpu()
        i = self._next
        s = self.slots[i]
        if s["busy"]:
            raise RuntimeError("slot %d still holds an uncollected result: call result() first (depth=%d)" % (i, self.depth))
        if frames is not None:
            src = frames if isinstance(frames, np.ndarray) else (frames.numpy() if not frames.is_cuda else None)
            if src is None or tuple(src.shape) != tuple(s["h_in"].shape) or src.dtype != np.uint8:
                raise ValueError("expected uint8 host frames of shape %s" % (tuple(s["h_in"].shape),))
            # numpy's single-threaded copy: torch's tensor.copy_ between host tensors runs on its CPU thread pool, which stalls for
            # ~90 ms now and then in a container with fewer CPUs than the machine shows (1 % of the batches of a 1500-batch soak)
            np.copyto(s["h_in"].numpy(), src)
        e = self.engine
        if self.transport == "zero_copy":
            with torch.cuda.stream(self.stream):
                ops.sr_fused_u8(s["h_in"], e.luts, self.geo, e.kind, e.max_sigma, out=s["h_out"], workspace=s["ws"])
                s["done"].record(self.stream)
        else:
            with torch.cuda.stream(self.stream):
                ops.sr_fused_u8(s["h_in"], e.luts, self.geo, e.kind, e.max_sigma, out=s["d_out"], workspace=s["ws"])
                s["sr_done"].record(self.stream)
            s["issued"] = False
            self._waiting.append(i)
            self._pump()
        s["busy"] = True
        self._next = (i + 1) % self.depth
        return i

    def result(self, i):
        s = self.slots[i]
        if not s["busy"]:
            raise RuntimeError("slot %d has nothing pending" % i)
        if self.transport == "dma":
            while not s["issued"]:
                self._pump(block=True)
            s["done"].synchronize()
            self._pump()                                   # the next download starts before the caller looks at this one
        else:
            s["done"].synchronize()
        s["busy"] = False
        return s["h_out"].numpy()

    def run(self, batches):
        """Pipelined map over an iterable of batches; yields outputs in order."""
        pending = []
        for b in batches:
            if len(pending) == self.depth:
                yield self.result(pending.pop(0))
            pending.append(self.submit(b))
        while pending:
            yield self.result(pending.pop(0))
