"""Device-side segmentation metrics: drop-in for `util.util.intersectionAndUnionGPU`
(reference util/util.py:55-67).  Unlike the reference it does not modify `output` in place and does not
depend on torch.histc (not implemented for int64 on CPU)."""
import torch

from . import ops
from ._lib import lib


def intersectionAndUnionGPU(output, target, K, ignore_index=255):
    assert output.dim() in [1, 2, 3]
    assert output.shape == target.shape
    assert output.is_cuda and target.is_cuda, "semseg_amd metrics run on the MI355X only"
    o = output.reshape(-1).to(torch.int64).contiguous()
    t = target.reshape(-1).to(torch.int64).contiguous()
    dev = o.device
    hist = torch.empty(3 * K, dtype=torch.int64, device=dev)
    out = torch.empty(3, K, dtype=torch.float32, device=dev)
    rc = lib.semseg_intersection_and_union(o.data_ptr(), t.data_ptr(), o.numel(), K, ignore_index,
                                           hist.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                           out[2].data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise ops.HipError("intersection_and_union failed with code %d" % rc)
    return out[0], out[1], out[2]
