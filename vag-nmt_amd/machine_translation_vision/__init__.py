"""MI355X-native drop-in for the hot path of ``machine_translation_vision`` (Eurus-Holmes/VAG-NMT).

Same import paths, class names, constructor/forward signatures and parameter names as the reference's
``models`` / ``layers`` / ``losses`` / ``utils.utils`` for the classes on the training/decoding path
(SURVEY.md section 8b); all arithmetic runs in hand-written HIP kernels (libvagnmt.so, include/vag_nmt.h).
There is no CPU fallback: CPU tensors, or a missing library, raise."""

from . import _checkout

# everything this package does not define (the METEOR wrapper, ...) comes from the user's checkout of the reference when one
# is on sys.path behind this package; see _checkout.py
_checkout.extend_path(__path__)
