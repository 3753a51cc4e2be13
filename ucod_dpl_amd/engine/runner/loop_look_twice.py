"""Look-Twice validation -- host-side mirror of engine/runner/loop_UCOD_DPL.py::ValLoop_Look_Twice (:276-417).

Same method names and semantics (``run``, ``process_preds``, ``look_twice``, ``resize_bbox``, ``expand_bbox``), with the
integer box logic reproduced bit-for-bit, quirks included: ``process_preds`` passes (h, w) into
``expand_bbox``'s (img_width, img_height) slots (:379), ``br = (h*y)/(H*W)`` (:404), ``sqrt(1 - br/fr + 1)`` raises
ValueError when br/fr > 2 (:405), ``int()`` truncates toward zero (:392-395,417), ``new_x`` may go negative (:412-416).

What moved to the GPU: the 68->518 bilinear upsample + threshold (one pass), the crop + Pillow-BILINEAR resize +
ToTensor + Normalize of EVERY box of an image in one batched launch pair (the reference loops PIL crop/resize per box on
the host), one batched backbone forward over all crops and one decoder pass at the native 37x37 grid (:343-345).  What
stays on the host, as in the reference: connected components (C++ in libucod_dpl.so instead of OpenCV), the float box
arithmetic, and the Pillow-BICUBIC resize + paste of the small refined masks (C++ restatement of Pillow's resampler).
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from ... import native as N, ops, parallel
from .loop_UCOD_DPL import BaseLoop
from ..utils.metrics import statistics

DEFAULT_BOX = [129, 129, 259, 259]


# ---------------------------------------------------------------------------------------------- host helpers
def connected_components(mask_u8):
    """cv2.connectedComponents(mask, connectivity=8) -> (num_labels, labels int32)."""
    m = np.ascontiguousarray(mask_u8, dtype=np.uint8)
    labels = np.empty(m.shape, np.int32)
    n = N.load().ucod_ccl8_host(m.ctypes.data, m.shape[0], m.shape[1], labels.ctypes.data)
    if n < 0:
        raise RuntimeError("ucod_ccl8_host failed")
    return n, labels


def bounding_rect(mask):
    """cv2.boundingRect -> (x, y, w, h)."""
    ys, xs = np.nonzero(mask)
    return int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)


def pil_resize_u8(src, out_w, out_h, bicubic=True):
    """Pillow Image.resize((out_w,out_h)) on an 'L' image (default BICUBIC), bit-identical."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty((out_h, out_w), np.uint8)
    N.check(N.load().ucod_pil_resize_u8_host(src.ctypes.data, src.shape[0], src.shape[1], dst.ctypes.data, out_h, out_w, 1 if bicubic else 0),
            "ucod_pil_resize_u8_host")
    return dst


class MAEStatistics:
    """The MAE part of engine/utils/metrics/metric.py::statistics (_prepare_data :125-133, MAEmeasure :187-207) on host arrays --
    kept for callers that hold numpy data; the validation loops use the full device-side ``statistics``."""

    def __init__(self):
        self.maes = []

    def step(self, gt_tensor, pred_tensor):
        gt = gt_tensor.detach().to("cpu").numpy().astype(float)
        pred = pred_tensor.detach().to("cpu").numpy().astype(float)
        for g, p in zip(gt, pred):
            g, p = np.squeeze(g), np.squeeze(p)
            if g.max() != g.min():
                g = (g - g.min()) / (g.max() - g.min())
            g = g > 0.5
            if p.max() != p.min():
                p = (p - p.min()) / (p.max() - p.min())
            else:
                p = p.astype(int)
            self.maes.append(np.mean(np.abs(p - g)))

    def get_result(self):
        return {"MAE": float(np.mean(np.array(self.maes, np.float64)))}


# ---------------------------------------------------------------------------------------------- the loop
class ValLoop_Look_Twice(BaseLoop):
    def __init__(self, config, runner, feature_extractor=None):
        super().__init__(config, runner)
        self._mode = "val"
        self.img_size = tuple(self.cfg.dataset_cfg.valset_cfg.image_size)
        if feature_extractor is None:
            from ...data.utils.feature_extractor import backbone
            feature_extractor = backbone(self.cfg.dataset_cfg.feature_extractor_cfg, device=runner.device)
        self.feature_extractor = feature_extractor
        self.device = runner.device
        self._ws = None
        self.gpu_tail = True                                                     # device CCL + device resize/paste (row N2)

    # ------------------------------------------------------------------ integer box logic (bit-exact)
    @staticmethod
    def resize_bbox(bbox, original_width, original_height, new_width, new_height):
        x, y, w, h = bbox
        width_scale = new_width / original_width
        height_scale = new_height / original_height
        return [int(x * width_scale), int(y * height_scale), int(w * width_scale), int(h * height_scale)]

    @staticmethod
    def expand_bbox(mask, bbox, img_width, img_height, expand_type="const", scale=1.3):
        x, y, w, h = bbox
        inside = mask[y:y + h, x:x + w].sum() if expand_type == "dynamic" else 0
        return ValLoop_Look_Twice._expand(inside, mask.shape[-2] * mask.shape[-1], bbox, img_width, img_height, expand_type, scale)

    @staticmethod
    def _expand(fg_in_box, mask_pixels, bbox, img_width, img_height, expand_type="const", scale=1.3):
        """expand_bbox with the only two things it reads from the mask passed as numbers (:386-417)."""
        x, y, w, h = bbox
        if expand_type == "dynamic":
            fr = np.float64(fg_in_box) / (h * w)
            br = (h * y) / mask_pixels
            scale = math.sqrt(1 - br / fr + 1)
        new_w = w * scale
        new_h = h * scale
        new_x = x - (new_w - w) / 2
        new_y = y - (new_h - h) / 2
        new_x = max(0, new_x)
        if new_x + new_w > img_width:
            new_x = img_width - new_w
        new_y = max(0, new_y)
        if new_y + new_h > img_height:
            new_y = img_height - new_h
        return [int(new_x), int(new_y), int(new_w), int(new_h)]

    def boxes_from_mask(self, mask_u8):
        """Integer tail of process_preds (:366-384) on a 0/255 uint8 [h,w] mask."""
        h, w = self.img_size
        num_labels, labels = connected_components(mask_u8)
        p = [(labels == i).sum() / (h * w) for i in range(1, num_labels)]
        if len(p) == 0:
            return [list(DEFAULT_BOX)]
        if max(p) < self.cfg.val_cfg.look_twice_th:
            bboxes = []
            for i in range(1, num_labels):
                if p[i - 1] > 0.01:
                    binary_mask = (labels == i).astype(np.uint8)
                    bboxes.append(self.expand_bbox(binary_mask, bounding_rect(binary_mask), h, w, expand_type=self.cfg.val_cfg.expand_type))
            return sorted(bboxes, key=lambda b: -1 * b[2] * b[3])
        return None

    def components_gpu(self, mask_u8_dev):
        """Device CCL -> host list of (area, x, y, w, h) per component in cv2's label order (raster order of the first 2 x 2 block)."""
        Hh, Ww = mask_u8_dev.shape
        lib = N.load()
        need = lib.ucod_ccl8_workspace_bytes(Hh, Ww)
        if getattr(self, "_ccl_ws", None) is None or self._ccl_ws.numel() < need:
            self._ccl_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        cap = 4096
        while True:
            table = torch.empty(cap, 7, dtype=torch.int32, device=self.device)
            count = torch.zeros(1, dtype=torch.int32, device=self.device)
            N.check(lib.ucod_ccl8_components(N.ptr(mask_u8_dev), Hh, Ww, N.ptr(table), cap, N.ptr(count), N.ptr(self._ccl_ws), self._ccl_ws.numel(),
                                             N.stream()), "ucod_ccl8_components")
            n = int(count.item())
            if n <= cap:
                break
            cap = n
        rows = table[:n].cpu().numpy()
        rows = rows[np.argsort(rows[:, 6], kind="stable")]
        return [(int(r[1]), int(r[2]), int(r[4]), int(r[3] - r[2] + 1), int(r[5] - r[4] + 1)) for r in rows]

    def components_gpu_batch(self, masks_u8_dev):
        """components_gpu for a batch [B,h,w] of device masks: every image's labelling is enqueued back to back (one workspace: the launches are
        stream-ordered), the B component counts come back in ONE transfer, a table that overflowed is redone alone."""
        Bn, Hh, Ww = masks_u8_dev.shape
        lib = N.load()
        need = lib.ucod_ccl8_workspace_bytes(Hh, Ww)
        if getattr(self, "_ccl_ws", None) is None or self._ccl_ws.numel() < need:
            self._ccl_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        cap = 4096
        tables = torch.empty(Bn, cap, 7, dtype=torch.int32, device=self.device)
        counts = torch.zeros(Bn, dtype=torch.int32, device=self.device)
        masks_u8_dev = masks_u8_dev.contiguous()
        for i in range(Bn):
            N.check(lib.ucod_ccl8_components(N.ptr(masks_u8_dev[i]), Hh, Ww, N.ptr(tables[i]), cap, N.ptr(counts[i:i + 1]), N.ptr(self._ccl_ws),
                                             self._ccl_ws.numel(), N.stream()), "ucod_ccl8_components")
        n_host = counts.cpu().tolist()
        biggest = max(min(n, cap) for n in n_host) if n_host else 0
        rows_host = tables[:, :max(biggest, 1)].cpu().numpy()
        out = []
        for i, n in enumerate(n_host):
            if n > cap:
                out.append(self.components_gpu(masks_u8_dev[i]))
                continue
            rows = rows_host[i, :n]
            rows = rows[np.argsort(rows[:, 6], kind="stable")]
            out.append([(int(r[1]), int(r[2]), int(r[4]), int(r[3] - r[2] + 1), int(r[5] - r[4] + 1)) for r in rows])
        return out

    def _boxes_from_components(self, comps, npix):
        """Integer tail of process_preds (:366-384) from one image's component list (area, x, y, w, h in cv2's label order)."""
        h, w = self.img_size
        p = [np.int64(c[0]) / (h * w) for c in comps]
        if len(p) == 0:
            return [list(DEFAULT_BOX)]
        if max(p) < self.cfg.val_cfg.look_twice_th:
            bboxes = []
            for (area, x, y, bw, bh), pi in zip(comps, p):
                if pi > 0.01:
                    # the reference hands expand_bbox the one-component mask: its sum inside the component's own box IS the area
                    bboxes.append(self._expand(np.uint64(area), npix, (x, y, bw, bh), h, w, expand_type=self.cfg.val_cfg.expand_type))
            return sorted(bboxes, key=lambda b: -1 * b[2] * b[3])
        return None

    def boxes_from_mask_gpu(self, mask_u8_dev):
        """Integer tail of process_preds (:366-384) from a DEVICE 0/255 uint8 [h,w] mask; same result as ``boxes_from_mask``."""
        return self._boxes_from_components(self.components_gpu(mask_u8_dev.contiguous()), int(mask_u8_dev.shape[-2]) * int(mask_u8_dev.shape[-1]))

    def process_preds(self, preds, label_tensor=None):
        """:354-384.  preds [1,1,fs,fs] logits -> (preds_up float [1,h,w] on the GPU, boxes | None)."""
        h, w = self.img_size
        up = ops.binarize(ops.bilinear_resize(preds.to(self.device, torch.float32), h, w), logits=True)     # one resize + threshold
        preds_up = up.reshape(-1, h, w)[:1]
        if getattr(self, "gpu_tail", True):
            return preds_up, self.boxes_from_mask_gpu((preds_up[0] * 255).to(torch.uint8))
        mask = (preds_up[0].cpu().numpy() * 255).astype(np.uint8)
        return preds_up, self.boxes_from_mask(mask)

    def process_preds_batch(self, preds):
        """:354-384 for a whole validation batch: preds [B,1,fs,fs] logits -> (preds_up float [B,h,w] on the GPU, one box list | None per image).
        One resize + threshold launch pair for the batch, every image's components enqueued back to back, one transfer of the tables; the box
        arithmetic per image is the single-image one (``_boxes_from_components``), so every image gets exactly the boxes ``process_preds`` gives it."""
        h, w = self.img_size
        Bn = preds.shape[0]
        up = ops.binarize(ops.bilinear_resize(preds.to(self.device, torch.float32), h, w), logits=True).reshape(Bn, h, w)
        comps = self.components_gpu_batch((up * 255).to(torch.uint8))
        return up, [self._boxes_from_components(c, h * w) for c in comps]

    def paste_gpu(self, masks_u8_dev, bboxes, canvas_u8_dev):
        """Pillow-BICUBIC resize of mask i to box i + paste, in order, on the device (:346-352)."""
        nb, sh, sw = masks_u8_dev.shape
        boxes = np.ascontiguousarray(np.asarray(bboxes, np.int32).reshape(nb, 4))
        if (boxes[:, 2:] <= 0).any():
            raise ValueError("height and width must be > 0")                  # what PIL's resize raises
        lib = N.load()
        need = lib.ucod_paste_workspace_bytes(nb, int(boxes[:, 2].max()), int(boxes[:, 3].max()), sh, sw)
        if getattr(self, "_paste_ws", None) is None or self._paste_ws.numel() < need:
            self._paste_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        CH, CW = canvas_u8_dev.shape
        N.check(lib.ucod_paste_resized_u8(N.ptr(masks_u8_dev), nb, sh, sw, boxes.ctypes.data, N.ptr(canvas_u8_dev), CH, CW, N.ptr(self._paste_ws),
                                          self._paste_ws.numel(), N.stream()), "ucod_paste_resized_u8")
        return canvas_u8_dev

    # ------------------------------------------------------------------ crop / re-encode / paste (:326-352)
    def crop_batch(self, img_u8, boxes_xywh):
        """img_u8: uint8 [H,W,3] (numpy or CUDA tensor); boxes in source pixels -> normalised crops [nbox,3,ih,iw] on the GPU."""
        ih, iw = self.img_size
        img = torch.as_tensor(img_u8).to(self.device).contiguous()
        H, W = img.shape[:2]
        nb = len(boxes_xywh)
        boxes = np.ascontiguousarray(np.asarray(boxes_xywh, np.int32).reshape(nb, 4))
        lib = N.load()
        need = lib.ucod_crop_workspace_bytes(nb, int(boxes[:, 3].max()), int(boxes[:, 2].max()), ih, iw)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty(nb, 3, ih, iw, dtype=torch.float32, device=self.device)
        N.check(lib.ucod_crop_resize_norm(N.ptr(img), H, W, boxes.ctypes.data, nb, N.ptr(out), ih, iw, N.ptr(self._ws), self._ws.numel(), N.stream()),
                "ucod_crop_resize_norm")
        return out

    def look_twice(self, path, bboxes, old_mask):
        """path: image file or uint8 [H,W,3] array.  old_mask float [1,h,w] in {0,1}.  Returns ToTensor(new_mask) [1,h,w]
        (on the device with ``gpu_tail``, on the CPU with the host tail)."""
        ih, iw = self.img_size
        if len(bboxes) == 0:
            # process_preds returns an EMPTY list when no component reaches 1 % of the image (:372-382): the reference's loop over it pastes nothing
            # and hands back ToTensor(ToPILImage(old_mask)) -- the old mask through an exact x 255 / 255 round trip
            canvas = (old_mask.squeeze(0) * 255).to(torch.uint8)
            if getattr(self, "gpu_tail", True):
                return torch.div(canvas.to(self.device).to(torch.float32), torch.full((), 255.0, device=self.device)).unsqueeze(0)
            return torch.from_numpy(canvas.cpu().numpy().astype(np.float32) / 255.0).unsqueeze(0)
        if isinstance(path, (str, os.PathLike)):
            from PIL import Image
            Image.MAX_IMAGE_PIXELS = None
            img = np.asarray(Image.open(path).convert("RGB"))
        else:
            img = np.asarray(path.cpu() if torch.is_tensor(path) else path)
        H, W = img.shape[:2]
        src_boxes = [self.resize_bbox(b, iw, ih, W, H) for b in bboxes]          # img.size = (W, H) (:335)
        crops = self.crop_batch(img, src_boxes)
        _, key = self.feature_extractor(crops)                                   # [nbox,C,37,37], all boxes in one pass
        with torch.no_grad():
            preds = self.runner.model(key)[0]                                    # decoder at the native grid, no 68x68 resize
        pred01_dev = ops.binarize(preds.contiguous(), logits=True).reshape(len(bboxes), preds.shape[-2], preds.shape[-1])
        if getattr(self, "gpu_tail", True):
            canvas_dev = (old_mask.squeeze(0).to(self.device) * 255).to(torch.uint8).contiguous()
            self.paste_gpu((pred01_dev * 255).to(torch.uint8).contiguous(), bboxes, canvas_dev)
            # ToTensor's x/255 as a true IEEE division: torch turns `tensor / python_scalar` on the GPU into a multiplication by
            # the reciprocal, and 255 * (1/255.f) != 1.0f
            return torch.div(canvas_dev.to(torch.float32), torch.full((), 255.0, device=self.device)).unsqueeze(0)
        canvas = (old_mask.squeeze(0).cpu().numpy() * 255).astype(np.uint8)
        pred01 = pred01_dev.cpu().numpy()
        for b, m in zip(bboxes, pred01):
            bx, by, bw, bh = b
            if bw <= 0 or bh <= 0:
                raise ValueError("height and width must be > 0")                  # what PIL's resize raises
            rs = pil_resize_u8((m * 255).astype(np.uint8), bw, bh, bicubic=True)
            x0, y0, x1, y1 = max(bx, 0), max(by, 0), min(bx + bw, canvas.shape[1]), min(by + bh, canvas.shape[0])
            if x1 > x0 and y1 > y0:
                canvas[y0:y1, x0:x1] = rs[y0 - by:y1 - by, x0 - bx:x1 - bx]
        return torch.from_numpy(canvas.astype(np.float32) / 255.0).unsqueeze(0)

    @staticmethod
    def _load_image(path):
        if isinstance(path, (str, os.PathLike)):
            from PIL import Image
            Image.MAX_IMAGE_PIXELS = None
            return np.asarray(Image.open(path).convert("RGB"))
        return path if torch.is_tensor(path) else np.asarray(path)

    max_crops_per_pass = 64                                                       # second-pass backbone batches (workspace ~0.3 GB per ViT-L crop)

    def look_twice_batch(self, paths, boxes_list, old_masks):
        """``look_twice`` (:326-352) for a validation batch -- SURVEY 8a row L3 "batch all crops of all images", BASELINE configs[3].
        paths: image files or uint8 [H,W,3] arrays (any sizes); boxes_list: per image a box list or None (image keeps its first-stage mask);
        old_masks float [B,h,w] in {0,1} on the device.  ONE crop launch pair over every crop of every image, the backbone over all crops (in passes of
        ``max_crops_per_pass``), one decoder pass at the native grid, one paste call.  Returns float [B,h,w] on the device; every image's result is what
        ``look_twice`` returns for it (same crop indices, same resampling tables, same paste order)."""
        ih, iw = self.img_size
        todo = [i for i, b in enumerate(boxes_list) if b]                          # None: no second look; []: nothing to paste (see look_twice)
        canvases = (old_masks.to(self.device) * 255).to(torch.uint8).contiguous()
        if not todo:
            return torch.div(canvases.to(torch.float32), torch.full((), 255.0, device=self.device))
        imgs, hw, box_img, src_boxes, dst_boxes, box_canvas = [], [], [], [], [], []
        for j, i in enumerate(todo):
            img = torch.as_tensor(self._load_image(paths[i])).to(self.device).contiguous()
            H, W = img.shape[:2]
            imgs.append(img)
            hw.append((H, W))
            for b in boxes_list[i]:
                src_boxes.append(self.resize_bbox(b, iw, ih, W, H))                # img.size = (W, H) (:335)
                dst_boxes.append(list(b))
                box_img.append(j)
                box_canvas.append(i)
        nb = len(src_boxes)
        lib = N.load()
        sb = np.ascontiguousarray(np.asarray(src_boxes, np.int32).reshape(nb, 4))
        need = lib.ucod_crop_workspace_bytes(nb, int(sb[:, 3].max()), int(sb[:, 2].max()), ih, iw)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        crops = torch.empty(nb, 3, ih, iw, dtype=torch.float32, device=self.device)
        ptrs = (C.c_void_p * len(imgs))(*[t.data_ptr() for t in imgs])
        hw_a = np.ascontiguousarray(np.asarray(hw, np.int32))
        bi_a = np.ascontiguousarray(np.asarray(box_img, np.int32))
        N.check(lib.ucod_crop_resize_norm_multi(ptrs, hw_a.ctypes.data, len(imgs), bi_a.ctypes.data, sb.ctypes.data, nb, N.ptr(crops), ih, iw, N.ptr(self._ws),
                                                self._ws.numel(), N.stream()), "ucod_crop_resize_norm_multi")
        keys = []
        for c0 in range(0, nb, self.max_crops_per_pass):
            keys.append(self.feature_extractor(crops[c0:c0 + self.max_crops_per_pass])[1])
        key = keys[0] if len(keys) == 1 else torch.cat(keys, 0)
        with torch.no_grad():
            preds = self.runner.model(key)[0]                                      # decoder at the native grid, no 68x68 resize (:343-345)
        pred01 = ops.binarize(preds.contiguous(), logits=True).reshape(nb, preds.shape[-2], preds.shape[-1])
        masks = (pred01 * 255).to(torch.uint8).contiguous()
        db = np.ascontiguousarray(np.asarray(dst_boxes, np.int32).reshape(nb, 4))
        if (db[:, 2:] <= 0).any():
            raise ValueError("height and width must be > 0")                      # what PIL's resize raises
        bc = np.ascontiguousarray(np.asarray(box_canvas, np.int32))
        pneed = lib.ucod_paste_workspace_bytes(nb, int(db[:, 2].max()), int(db[:, 3].max()), masks.shape[1], masks.shape[2])
        if getattr(self, "_paste_ws", None) is None or self._paste_ws.numel() < pneed:
            self._paste_ws = torch.empty(pneed, dtype=torch.uint8, device=self.device)
        N.check(lib.ucod_paste_resized_u8_multi(N.ptr(masks), nb, masks.shape[1], masks.shape[2], db.ctypes.data, bc.ctypes.data, N.ptr(canvases), canvases.shape[0],
                                                canvases.shape[1], canvases.shape[2], N.ptr(self._paste_ws), self._paste_ws.numel(), N.stream()),
                "ucod_paste_resized_u8_multi")
        return torch.div(canvases.to(torch.float32), torch.full((), 255.0, device=self.device))      # ToTensor's x / 255 as a true division (see look_twice)

    def validate_batch(self, features, paths):
        """First-stage decode + Look-Twice refinement of a batch: features [B,C,gh,gw] (cache items), paths: the images.  -> float [B,h,w] in {0,1}."""
        fs = self.cfg.model_cfg.feature_size
        features = ops.bilinear_resize(features.to(self.device, torch.float32), fs, fs)
        with torch.no_grad():
            preds = self.runner.model(features)[0]
        preds_up, boxes = self.process_preds_batch(preds)
        if self.cfg.val_cfg.look_twice and any(b is not None for b in boxes):
            preds_up = self.look_twice_batch(paths, boxes, preds_up)
        return preds_up, boxes

    # ------------------------------------------------------------------ :297-324
    def run(self):
        stats = statistics()                                                       # all nine COD measures, on the device (:299)
        self.runner.model.eval()
        fs = self.cfg.model_cfg.feature_size
        # Multi-rank: every rank walks ITS shard of the validation set (the reference's accelerator.prepare shards the loader) with no
        # per-image collective; the per-image records meet once, before get_result (the role of gather_for_metrics, :310).
        # The reference walks the set one image at a time (its code assumes batch size 1, :313-317); here `look_twice_batch` images (val_cfg, default 16 =
        # BASELINE configs[3]) are decoded, boxed, cropped, re-encoded and pasted together -- per image the same result (tests/test_gpu_look_twice.py) --
        # and only the last step, the resize to each image's own label size + the measures, is per image as in the reference (:315-317).
        group = int(self.cfg.val_cfg.get("look_twice_batch", 16))
        pending = []

        def flush():
            if not pending:
                return
            feats = torch.cat([f.reshape(1, *f.shape[-3:]) for _, f, _ in pending], 0)
            preds_up, _ = self.validate_batch(feats, [p for _, _, p in pending])
            for (label_tensor, _, _), pu in zip(pending, preds_up):
                out = ops.bilinear_resize(pu.reshape(1, 1, *pu.shape[-2:]).contiguous(), label_tensor.shape[-2], label_tensor.shape[-1])
                stats.step(label_tensor.to(self.device), (out.reshape(1, *out.shape[-2:]) > 0.5))
            pending.clear()

        for batch in parallel.shard(self.runner.val_dataloader):
            _, label_tensor, features, img_path = batch.values()
            for j in range(features.shape[0]):                                     # (the shipped configs use batch size 1)
                pending.append((label_tensor[j:j + 1], features[j], img_path[j]))
            if len(pending) >= group:
                flush()
        flush()
        stats.gather_records(device=self.device, dataset_len=parallel.padded_sampler_len(self.runner.val_dataloader))
        result = stats.get_result()
        self.runner.logger.log_table({k: [round(v, 4)] for k, v in result.items()})
        return result
