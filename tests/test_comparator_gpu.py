"""OPT-IN measurement, not a parity test ("optional second comparator"): the reference's arithmetic
(oracle/segnet.py, bit-identical to the imported reference) run through torch-ROCm's own operators (MIOpen convolutions,
ATen BatchNorm / interpolate / cross-entropy) on the same MI355X, forward + backward + SGD — "what the unmodified
reference would get on this GPU".  It is skipped unless SEMSEG_RUN_COMPARATOR=1 because MIOpen's first-run kernel search
over ~100 convolution shapes x 3 directions can take many minutes.  Writes its line to gpurun_out/parity_report.txt.

  SEMSEG_RUN_COMPARATOR=1 [COMPARATOR_BATCH=16 COMPARATOR_LAYERS=101 COMPARATOR_SIZE=473] \\
      python -m pytest tests/test_comparator_gpu.py -m gpu -q -s
"""
import os
import time

import pytest
import torch


def measure(device, layers, classes, size, batch, iters, warm):
    from oracle import segnet
    from model.pspnet import PSPNet
    torch.manual_seed(0)
    m = PSPNet(layers=layers, classes=classes, zoom_factor=8, pretrained=False)
    sd = {k: v.detach().clone().to(device) for k, v in m.state_dict().items()}
    params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
    opt = torch.optim.SGD(params, lr=0.01, momentum=0.9, weight_decay=1e-4)
    x = torch.randn(batch, 3, size, size, device=device)
    y = torch.randint(0, classes, (batch, size, size), device=device)

    def step():
        _, ml, al = segnet.forward(sd, x, layers, "psp", training=True, y=y)
        loss = ml + 0.4 * al
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return ml

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize()

    for _ in range(warm):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        ml = step()
    sync()
    dt = (time.perf_counter() - t0) / iters
    return batch / dt, dt * 1e3, float(ml)


def test_measure_runs_on_cpu_at_toy_size():
    """keeps the measurement code itself exercised (CPU, seconds)"""
    ips, ms, loss = measure(torch.device("cpu"), 50, 5, 33, 2, 1, 1)
    assert ips > 0 and ms > 0 and loss == loss


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get("SEMSEG_RUN_COMPARATOR") != "1", reason="opt-in: MIOpen kernel search takes minutes")
def test_torch_rocm_comparator(report):
    B = int(os.environ.get("COMPARATOR_BATCH", "16"))
    layers = int(os.environ.get("COMPARATOR_LAYERS", "101"))
    size = int(os.environ.get("COMPARATOR_SIZE", "473"))
    # COMPARATOR_FIND=1: MIOpen find mode (fastest solver per shape, minutes of search); 0: immediate mode
    torch.backends.cudnn.benchmark = os.environ.get("COMPARATOR_FIND", "1") == "1"
    ips, ms, loss = measure(torch.device("cuda"), layers, 150, size, B, 5, 3)
    report("torch-ROCm / MIOpen comparator (find=%s; reference arithmetic, fp32): PSPNet-%d %dx%d batch %d train step %.1f ms = "
           "%.2f images/s (loss %.4f)" % (os.environ.get("COMPARATOR_FIND", "1"), layers, size, size, B, ms, ips, loss))
    assert ips > 0
