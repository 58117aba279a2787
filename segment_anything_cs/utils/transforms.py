"""ResizeLongestSide (reference: segment_anything_cs/utils/transforms.py:14-102), host side.

The image resize itself is third-party arithmetic in the reference (PIL bilinear through
torchvision, parity unpinned); it is an identity for frames already 1024 on the long side, which is
what CrowdSAM.crop_image produces in all but the int(r*w) == 1023 case (SURVEY.md trap 9).
"""
from copy import deepcopy

import numpy as np
from PIL import Image


class ResizeLongestSide:
    def __init__(self, target_length):
        self.target_length = target_length

    @staticmethod
    def get_preprocess_shape(oldh, oldw, long_side_length):
        scale = long_side_length * 1.0 / max(oldh, oldw)
        return int(oldh * scale + 0.5), int(oldw * scale + 0.5)

    def apply_image(self, image):
        """HxWxC uint8 -> resized uint8 (PIL bilinear, as torchvision's resize(to_pil_image(.)))."""
        th, tw = self.get_preprocess_shape(image.shape[0], image.shape[1], self.target_length)
        if (th, tw) == image.shape[:2]:
            return np.array(image)
        return np.array(Image.fromarray(image).resize((tw, th), Image.BILINEAR))

    def apply_coords(self, coords, original_size):
        """(...,2) xy coordinates -> input frame, float64 arithmetic (transforms.py:33-45)."""
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(old_h, old_w, self.target_length)
        out = deepcopy(coords).astype(float)
        out[..., 0] = out[..., 0] * (new_w / old_w)
        out[..., 1] = out[..., 1] * (new_h / old_h)
        return out

    def apply_boxes(self, boxes, original_size):
        return self.apply_coords(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)

    def apply_coords_torch(self, coords, original_size):
        """Tensor form of apply_coords (transforms.py:68-81): fp32 on the tensor's device."""
        import torch
        old_h, old_w = original_size
        new_h, new_w = self.get_preprocess_shape(old_h, old_w, self.target_length)
        out = coords.clone().to(torch.float)
        out[..., 0] = out[..., 0] * (new_w / old_w)
        out[..., 1] = out[..., 1] * (new_h / old_h)
        return out

    def apply_boxes_torch(self, boxes, original_size):
        """Bx4 XYXY tensor -> input frame (transforms.py:83-91)."""
        return self.apply_coords_torch(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)
