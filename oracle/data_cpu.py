"""CPU restatement of the reference's image transform pipeline.  TEST INFRASTRUCTURE ONLY (imported by tests/ only).

``transforms.Compose([ToTensor(), CenterCrop(148), Resize(patch_size)])`` (dataset.py:72-80) applied to a tensor.
torchvision is not installed here, so its two tensor-path functions are restated from their published definitions
(torchvision 0.13, the version requirements.txt pins):

* ``F.center_crop``: an image smaller than the crop is zero-padded with ``(c-h)//2`` rows above and ``(c-h+1)//2`` below
  (same for columns); then ``top = int(round((h - c) / 2.0))``, ``left = int(round((w - c) / 2.0))``.
* ``F.resize`` of a tensor with an int size and square input = ``torch.nn.functional.interpolate(mode="bilinear",
  align_corners=False)`` (antialias defaults to off for tensors) -- torch itself IS available, so that call is the real one.
"""
import torch
import torch.nn.functional as F


def center_crop(img_chw: torch.Tensor, crop: int) -> torch.Tensor:
    _, h, w = img_chw.shape
    if crop > w or crop > h:
        pl = (crop - w) // 2 if crop > w else 0
        pt = (crop - h) // 2 if crop > h else 0
        pr = (crop - w + 1) // 2 if crop > w else 0
        pb = (crop - h + 1) // 2 if crop > h else 0
        img_chw = F.pad(img_chw, (pl, pr, pt, pb), value=0.0)
        _, h, w = img_chw.shape
        if crop == w and crop == h:
            return img_chw
    top = int(round((h - crop) / 2.0))
    left = int(round((w - crop) / 2.0))
    return img_chw[:, top:top + crop, left:left + crop]


def reference_transform(images_u8: torch.Tensor, rows: torch.Tensor, crop: int = 148, size: int = 64) -> torch.Tensor:
    """images_u8 [N,H,W,3] uint8 (CPU), rows int64 [B] -> [B,3,size,size] float32."""
    out = []
    for r in rows.tolist():
        t = images_u8[r].permute(2, 0, 1).to(torch.float32) / 255.0            # ToTensor
        t = center_crop(t, crop)
        t = F.interpolate(t.unsqueeze(0), size=(size, size), mode="bilinear", align_corners=False)[0]
        out.append(t)
    return torch.stack(out)
