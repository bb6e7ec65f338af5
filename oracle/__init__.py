"""CPU oracle for the sampling hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain fp32 PyTorch-CPU / numpy restatement of the reference's
sampler arithmetic (U-Net forward, DDPM tables + ancestral step, fixed-step
Euler, conditioning builders, uint8 post-processing).  Every function cites the
reference file:line it follows.

Rules (enforced by tests/test_layout_rules.py):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import anything from here;
  * the product package never imports it and never falls back to it.

Pinning: the U-Net / GroupNorm / attention / DDPM-table / likelihood
restatements are pinned against outputs of the reference's own modules
(imported read-only in the build container by tools/make_goldens.py; vectors
committed under tests/golden/).  The sampler *loops* (sampling.py needs the
un-vendored `plum` package; cifar10/ and mnist/ need torchdyn/torchcfm) are
restated from the source text: their per-step arithmetic is pinned through the
reference's DDPM methods, the loop order itself is "parity unpinned".
"""
