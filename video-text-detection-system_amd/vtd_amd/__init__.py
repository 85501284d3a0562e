"""vtd_amd -- MI355X-native per-frame text detection / recognition hot path.

Host side (Python on PyTorch-ROCm) of the drop-in replacement for the reference's
``app/ml`` package (reference: app/ml/__init__.py:1-22).  All compute runs in the
hand-written HIP kernels of ``csrc/`` behind the C ABI declared in ``include/vtd.h``;
this package only owns parameters, device buffers, streams and result plumbing.
There is no CPU fallback: constructing a model without the HIP library or without a
GPU raises (see ``_native.require``).
"""

from .vocab import build_vocab, VOCAB_CHARS  # noqa: F401

__version__ = "0.1.0"
