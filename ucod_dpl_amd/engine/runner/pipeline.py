"""Feature pipeline: the frozen backbone forward of batch k+1 overlapped with the decoder step of batch k.

The first-stage training step has two parts with no dependence across batches: the frozen backbone pass (large MFMA kernels,
~88 % of the step) and the DBA / APM / discriminator / optimiser step (two dozen small, latency-bound kernels that leave most
CUs idle).  With the backbone on side HIP streams (``ViTEngine.forward_async``) the small kernels of step k run while the
backbone kernels of step k+1 occupy the rest of the chip.  Results are identical to the serial order: the backbone does not
read anything the decoder step writes (data/utils/feature_extractor.py: frozen, no_grad), and decoder steps stay in order on
the main stream.  This is the device-side counterpart of the reference's cached-features loop, where the backbone cost was
moved out of the training loop altogether (base_dataset.py:124-145).
"""
import torch


class FeaturePipeline:
    def __init__(self, engine, depth=2):
        self.engine = engine
        self.depth = depth
        self._ring = [None] * depth
        self._n = 0
        self._pending = []

    def submit(self, images):
        """Enqueue the backbone pass for ``images`` (side streams).  Call after the previous decoder step was enqueued."""
        B, _, H, W = images.shape
        slot = self._n % self.depth
        shape = (B, self.engine.D, H // self.engine.P, W // self.engine.P)
        if self._ring[slot] is None or tuple(self._ring[slot].shape) != shape:
            self._ring[slot] = torch.empty(shape, dtype=torch.float32, device=images.device)
        key, events = self.engine.forward_async(images, out=self._ring[slot])
        self._pending.append((key, events))
        self._n += 1

    def next_features(self):
        """Key map of the oldest submitted batch; the current stream waits for its side streams."""
        key, events = self._pending.pop(0)
        cur = torch.cuda.current_stream(key.device)
        for e in events:
            cur.wait_event(e)
        return key
