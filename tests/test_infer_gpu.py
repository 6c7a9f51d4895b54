"""Device-side multi-scale test pipeline (semseg_amd/infer.py) against the CPU restatement of
tool/test.py (oracle/test_pipeline.py) with the oracle network on the same weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_resize_and_geometry(report):
    from oracle import test_pipeline as tp
    from semseg_amd import ops
    g = np.random.default_rng(0)
    img = (g.random((37, 53, 3)) * 255).astype(np.float32)
    for (nh, nw) in [(74, 106), (19, 27), (37, 53), (50, 41)]:
        ref = tp.cv2_resize_linear(img, nw, nh)
        dst = torch.empty(nh, nw, 3, device="cuda")
        ops.resize_linear_hwc(torch.from_numpy(img).cuda(), 37, 53, dst, nh, nw, 3)
        e = np.abs(dst.cpu().numpy() - ref).max() / 255
        assert e < 1e-5, (nh, nw, e)
    report("resize_linear_hwc == half-pixel bilinear (cv2 INTER_LINEAR formula) on 4 sizes")


@pytest.mark.parametrize("hw,scales", [((97, 130), (0.5, 1.0, 1.75)), ((150, 90), (0.75, 1.25))])
def test_multi_scale_pipeline_vs_oracle(hw, scales, report):
    from model.pspnet import PSPNet
    from oracle import segnet, test_pipeline as tp
    from semseg_amd.infer import MultiScaleTester
    classes, crop, base = 7, 73, 96
    m = PSPNet(layers=50, classes=classes, zoom_factor=8, pretrained=False)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = segnet.recipe_state_dict(shapes, seed=5)
    # the recipe's eval-mode logits reach ~1e4, where softmax turns fp32 round-off (3e-6 relative) into
    # O(0.1) probability changes; scale the classifier so the probabilities are well conditioned
    sd["cls.4.weight"] *= 1e-3
    sd["cls.4.bias"] *= 1e-3
    m.load_state_dict(sd)
    g = np.random.default_rng(1)
    img = (g.random((hw[0], hw[1], 3)) * 255).astype(np.float32)
    mean = [0.485 * 255, 0.456 * 255, 0.406 * 255]
    std = [0.229 * 255, 0.224 * 255, 0.225 * 255]

    def cpu_model(x):
        return segnet.forward({k: v.clone() for k, v in sd.items()}, x, 50, "psp", training=False)
    ref_arg, ref_prob = tp.multi_scale_predict(cpu_model, img, classes, base, crop, crop, scales, mean, std)
    t = MultiScaleTester(m.cuda(), classes, base, crop, crop, scales, mean, std)
    pred, prob = t.predict(img, return_prob=True)
    prob = prob.permute(1, 2, 0).cpu().numpy()
    e = np.abs(prob - ref_prob).max()
    agree = float((pred.cpu().numpy() == ref_arg).mean())
    report("multi-scale test pipeline %s scales %s: prob max-abs err %.2e argmax agreement %.5f (%d forwards)"
           % (hw, scales, e, agree, t.num_forwards(*hw)))
    assert e < 2e-4 and agree > 0.998


def test_sharded_multi_scale_two_ranks(report):
    """BASELINE.json configs[4] / SURVEY.md section 8e "Test path": the crops of all scales sharded over 2 ranks + ONE
    reduce of the [C,h,w] probability sum reproduce the single-process result (fp32 summation order differs)."""
    import os
    import socket
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tmp = tempfile.mkdtemp(prefix="semseg_infer_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(root, "tests", "infer_worker.py")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    subprocess.check_call([sys.executable, worker, tmp], env=env, timeout=600)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), worker, tmp], env=env, timeout=900)
    one = np.load(os.path.join(tmp, "infer_rank0_of1.npz"))
    r0 = np.load(os.path.join(tmp, "infer_rank0_of2.npz"))
    r1 = np.load(os.path.join(tmp, "infer_rank1_of2.npz"))
    assert r1["pred"].size == 0 and int(r0["ncrops"]) + int(r1["ncrops"]) == int(one["ncrops"])
    assert abs(int(r0["ncrops"]) - int(r1["ncrops"])) <= 1
    e = float(np.abs(r0["prob"] - one["prob"]).max())
    agree = float((r0["pred"] == one["pred"]).mean())
    report("crop-sharded multi-scale test, 2 ranks (%d + %d crops) vs 1: prob max-abs diff %.2e, argmax agreement %.5f"
           % (int(r0["ncrops"]), int(r1["ncrops"]), e, agree))
    assert e < 2e-6 and agree > 0.9999
