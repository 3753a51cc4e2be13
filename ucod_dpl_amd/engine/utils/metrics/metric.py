"""COD measures -- host-side mirror of engine/utils/metrics/metric.py::statistics on the HIP kernels (csrc/cod_metrics.hip).

Same interface: ``statistics().step(gt_tensor, pred_tensor)`` per validation batch, ``get_result()`` -> the reference's dict
{"ACC", "mIOU", "E_MAX", "E_MEAN", "F_MAX", "F_MEAN", "SMeasure", "MAE", "WFM"} (metric.py:19-74), ``reset()``.  The reference
moves every batch to the host and runs seven numpy measures per image; here a batch of same-sized images is one library call
that leaves a [B, 1032] float64 record on the device, records are only concatenated by ``step`` (no synchronisation), and
``get_result`` reduces them (means over images; maxima / means over the 256 thresholds of the mean curves).
"""
import torch

from .... import ops

_MAE, _ACC, _IOU, _SM, _WFM, _ADP_EM, _ADP_FM = range(7)
_EM, _FM, _P, _R = (slice(8 + 256 * i, 8 + 256 * (i + 1)) for i in range(4))


def compute_cod_metric(em, sm, fm, mae, wfm):
    """metric.py:12-18."""
    return em["curve"].max(), em["curve"].mean(), fm["curve"].max(), fm["curve"].mean(), sm, mae, wfm


class statistics:
    def __init__(self):
        self.reset()

    def reset(self):
        self._records = []

    @staticmethod
    def _planes(t):
        """[B,1,H,W] / [B,H,W] (the two shapes step() accepts, metric.py:41-50) -> f32 [B,H,W] on the device."""
        if t.dim() == 4:
            if t.shape[1] != 1:
                raise ValueError(f"expected one channel, got {tuple(t.shape)}")
            t = t[:, 0]
        if t.dim() != 3:
            raise ValueError(f"expected [B,1,H,W] or [B,H,W], got {tuple(t.shape)}")
        if not t.is_cuda:
            raise RuntimeError("statistics.step: tensors must be on the GPU (the measures run there; there is no CPU path)")
        return t.to(torch.float32).contiguous()

    def step(self, gt_tensor, pred_tensor):
        self._records.append(ops.cod_metrics(self._planes(pred_tensor), self._planes(gt_tensor)))

    def per_image(self):
        """f64 [images, 1032] on the device (layout: include/ucod_dpl.h)."""
        return torch.cat(self._records, dim=0)

    def gather_records(self, device=None, dataset_len=None):
        """Multi-rank validation (what ``accelerator.gather_for_metrics`` is for in the reference, loop_UCOD_DPL.py:310): every rank
        has stepped over ITS shard of the validation set; collect all ranks' per-image records so that ``get_result`` is the
        measure over the whole set and identical on every rank.  Two collectives per validation run (counts, then the records
        padded to the longest shard), none per image; ranks may hold different numbers of images, including none.  ``device``: where a
        rank WITHOUT records builds its empty contribution (default: the current GPU under nccl, the CPU under gloo) -- it must be the
        device the other ranks' records live on.  ``dataset_len``: for shards cut by an interleaving DistributedSampler WITHOUT drop_last
        (rank r holds images r, r + world, ...; the sampler pads the tail with repeats of the first images): the i-th record of rank r is
        image i * world + r, and records at or past ``dataset_len`` are those repeats -- they are dropped, as gather_for_metrics drops them."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        if self._records:
            mine = self.per_image()
        else:
            dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu")
            mine = torch.zeros(0, 1032, dtype=torch.float64, device=dev)
        counts = [torch.zeros(1, dtype=torch.int64, device=mine.device) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([mine.shape[0]], dtype=torch.int64, device=mine.device))
        counts = [int(c.item()) for c in counts]
        longest = max(counts)
        if longest == 0:
            return
        padded = torch.zeros(longest, mine.shape[1], dtype=mine.dtype, device=mine.device)
        padded[:mine.shape[0]] = mine
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded)
        if dataset_len is not None:
            parts = [p[:max(0, min(n, -(-(int(dataset_len) - r) // world)))] for r, (p, n) in enumerate(zip(parts, counts))]
            self._records = [torch.cat(parts, dim=0)]
            return
        self._records = [torch.cat([p[:n] for p, n in zip(parts, counts)], dim=0)]      # rank order: deterministic on every rank

    def get_result(self):
        r = self.per_image()
        mean = r.mean(dim=0)                                   # over images
        em, fm = mean[_EM], mean[_FM]
        out = {"ACC": mean[_ACC], "mIOU": mean[_IOU], "E_MAX": em.max(), "E_MEAN": em.mean(), "F_MAX": fm.max(), "F_MEAN": fm.mean(),
               "SMeasure": mean[_SM], "MAE": mean[_MAE], "WFM": mean[_WFM]}
        return {k: float(v) for k, v in out.items()}
