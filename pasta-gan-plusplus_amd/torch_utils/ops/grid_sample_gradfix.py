"""API stub of the reference's grid_sample_gradfix (torch_utils/ops/grid_sample_gradfix.py:24-28):
only the discriminator-side AugmentPipe uses it (augment.py:298), which is outside the synthesis
hot path; on current PyTorch the reference itself falls through to F.grid_sample (:34-40)."""

import torch

enabled = False


def grid_sample(input, grid):
    return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)
