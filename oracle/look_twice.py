"""Oracle: Look-Twice crop-and-rezoom refinement.  TEST INFRASTRUCTURE ONLY.

Restates engine/runner/loop_UCOD_DPL.py:354-384 (process_preds), :387-397 (resize_bbox),
:399-417 (expand_bbox) and :326-352 (look_twice).  The integer box logic reproduces the
reference bit-for-bit *including its quirks* (SURVEY.md section 7 / Appendix A):
  * process_preds passes (h, w) into expand_bbox's (img_width, img_height) slots (:379);
  * ``br = (h*y)/(H*W)`` (:404) and ``sqrt(1 - br/fr + 1)`` raising ValueError when br/fr > 2 (:405);
  * ``int()`` truncation toward zero (:392-395,417) and a possibly negative new_x/new_y (:412-416).
cv2 is not installed here (nor vendored by the reference): ``connected_components`` restates
8-connected labelling with OpenCV's NUMBERING (label k = k-th component in raster order of its first
2 x 2 block: cv2 labels 2 x 2 blocks in raster order -- Grana's BBDT / Bolelli's Spaghetti,
modules/imgproc/src/connectedcomponents.cpp -- unions keep the smaller provisional label and flattenL
renumbers roots in increasing order; label 0 = background) and ``bounding_rect`` restates
cv2.boundingRect -> (x, y, w, h).
"""
import math
import numpy as np
import torch

from .resize import torch_bilinear, pil_resize_u8

DEFAULT_BOX = [129, 129, 259, 259]                      # loop_UCOD_DPL.py:370
IMAGENET_MEAN = (0.485, 0.456, 0.406)                   # :285
IMAGENET_STD = (0.229, 0.224, 0.225)


def connected_components(img):
    """8-connectivity labelling of a uint8/bool [H,W] image -> (num_labels, labels int32)."""
    fg = np.asarray(img) > 0
    H, W = fg.shape
    labels = np.zeros((H, W), np.int32)
    parent = [0]

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for y in range(H):
        for x in range(W):
            if not fg[y, x]:
                continue
            neigh = []
            if x > 0 and labels[y, x - 1]:
                neigh.append(labels[y, x - 1])
            if y > 0:
                for dx in (-1, 0, 1):
                    xx = x + dx
                    if 0 <= xx < W and labels[y - 1, xx]:
                        neigh.append(labels[y - 1, xx])
            if not neigh:
                parent.append(len(parent))
                labels[y, x] = len(parent) - 1
            else:
                roots = [find(n) for n in neigh]
                m = min(roots)
                labels[y, x] = m
                for r in roots:
                    parent[r] = m
    remap = {}
    for by in range(0, H, 2):                              # OpenCV's numbering: first 2 x 2 block in raster order
        for bx in range(0, W, 2):
            for y in range(by, min(by + 2, H)):
                for x in range(bx, min(bx + 2, W)):
                    if labels[y, x]:
                        r = find(labels[y, x])
                        if r not in remap:
                            remap[r] = len(remap) + 1
    out = np.zeros_like(labels)
    for y in range(H):
        for x in range(W):
            if labels[y, x]:
                out[y, x] = remap[find(labels[y, x])]
    return len(remap) + 1, out


def bounding_rect(mask):
    ys, xs = np.nonzero(mask)
    return int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)


def resize_bbox(bbox, original_width, original_height, new_width, new_height):
    """loop_UCOD_DPL.py:387-397."""
    x, y, w, h = bbox
    ws = new_width / original_width
    hs = new_height / original_height
    return [int(x * ws), int(y * hs), int(w * ws), int(h * hs)]


def expand_bbox(mask, bbox, img_width, img_height, expand_type="const", scale=1.3):
    """loop_UCOD_DPL.py:399-417 (python-float == IEEE double arithmetic)."""
    x, y, w, h = bbox
    if expand_type == "dynamic":
        fr = float(mask[y:y + h, x:x + w].sum()) / (h * w)
        br = (h * y) / (mask.shape[-2] * mask.shape[-1])
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


def boxes_from_mask(mask_u8, h, w, look_twice_th, expand_type):
    """The integer tail of process_preds (:366-384) on a 0/255 uint8 [h,w] mask."""
    num_labels, labels = connected_components(mask_u8)
    p = [(labels == i).sum() / (h * w) for i in range(1, num_labels)]
    if len(p) == 0:
        return [list(DEFAULT_BOX)]
    if max(p) < look_twice_th:
        bboxes = []
        for i in range(1, num_labels):
            if p[i - 1] > 0.01:
                bm = (labels == i).astype(np.uint8)
                bboxes.append(expand_bbox(bm, bounding_rect(bm), h, w, expand_type=expand_type))   # (h, w) sic
        return sorted(bboxes, key=lambda b: -1 * b[2] * b[3])
    return None


def process_preds(preds, img_size, look_twice_th, expand_type):
    """loop_UCOD_DPL.py:354-384.  preds [1,1,fs,fs] logits -> (mask float [1,h,w], boxes|None)."""
    h, w = img_size
    up = torch_bilinear(preds, h, w)[..., :h, :w]
    up = (torch.sigmoid(up) > 0.5).squeeze(0).float()
    m = (up.numpy() * 255).astype(np.uint8)
    if m.ndim == 3:
        m = m.squeeze(0)
    return up, boxes_from_mask(m, h, w, look_twice_th, expand_type)


def crop_resize_normalize(img_u8, box_xywh, out_hw):
    """PIL ``img.crop((l,t,r,b))`` + torchvision ``Resize(out_hw)`` (Pillow BILINEAR, antialiased)
    + ``ToTensor`` + ``Normalize`` (:282-286,341-342).  img_u8 [H,W,3] uint8.  Crops that
    leave the image are zero-filled as PIL's crop does."""
    x, y, w, h = box_xywh
    H, W = img_u8.shape[:2]
    crop = np.zeros((max(h, 0), max(w, 0), 3), np.uint8)
    sx0, sy0 = max(x, 0), max(y, 0)
    sx1, sy1 = min(x + w, W), min(y + h, H)
    if sx1 > sx0 and sy1 > sy0:
        crop[sy0 - y:sy1 - y, sx0 - x:sx1 - x] = img_u8[sy0:sy1, sx0:sx1]
    oh, ow = out_hw
    r = pil_resize_u8(crop, ow, oh, "bilinear")
    t = torch.from_numpy(r.astype(np.float32) / 255.0).permute(2, 0, 1)
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return (t - mean) / std


def paste_mask(canvas_u8, pred01, bbox):
    """:348-351.  pred01 float [h,w] in {0,1} -> 'L' image (x255) -> Image.resize((bw,bh)) with the
    Pillow default BICUBIC -> paste at (bx,by) with PIL clipping semantics.  canvas_u8 [H,W] mutated."""
    bx, by, bw, bh = bbox
    src = (pred01.numpy() * 255).astype(np.uint8)
    rs = pil_resize_u8(src, bw, bh, "bicubic")
    H, W = canvas_u8.shape
    x0, y0 = max(bx, 0), max(by, 0)
    x1, y1 = min(bx + bw, W), min(by + bh, H)
    if x1 > x0 and y1 > y0:
        canvas_u8[y0:y1, x0:x1] = rs[y0 - by:y1 - by, x0 - bx:x1 - bx]
    return canvas_u8


def look_twice(img_u8, bboxes, old_mask, img_size, encode_fn):
    """:326-352.  img_u8 [H,W,3]; old_mask float [1,h,w] in {0,1}; ``encode_fn(crop[1,3,h,w]) ->
    fg logits [1,1,gh,gw]`` (backbone key feature -> decoder at the native 37x37 grid, :343-345).
    Returns float [1,h,w] = ToTensor(new_mask)."""
    ih, iw = img_size
    canvas = (old_mask.squeeze(0).numpy() * 255).astype(np.uint8)
    H, W = img_u8.shape[:2]
    for bbox in bboxes:
        nb = resize_bbox(bbox, iw, ih, W, H)             # img.size = (W, H)
        crop = crop_resize_normalize(img_u8, nb, (ih, iw)).unsqueeze(0)
        logits = encode_fn(crop)
        pred = (torch.sigmoid(logits) > 0.5).reshape(logits.shape[-2], logits.shape[-1]).float()
        paste_mask(canvas, pred, bbox)
    return torch.from_numpy(canvas.astype(np.float32) / 255.0).unsqueeze(0)
