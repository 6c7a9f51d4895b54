"""CPU: geometry of the multi-scale test pipeline (tool/test.py:150-170,191-200)."""
import torch


def test_forward_count_matches_survey():
    """512x512 image, six ADE scales, crop 473 -> 23 crops x 2 flips = 46 forwards."""
    from semseg_amd.infer import MultiScaleTester

    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
    t = MultiScaleTester(_M(), 150, 512, 473, 473, (0.5, 0.75, 1.0, 1.25, 1.5, 1.75))
    assert t.num_forwards(512, 512) == 46
