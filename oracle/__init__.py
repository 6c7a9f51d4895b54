"""TEST INFRASTRUCTURE.  CPU oracles for the PSPNet/PSANet hot path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(semseg_amd/, model/, lib/) must never do so."""
