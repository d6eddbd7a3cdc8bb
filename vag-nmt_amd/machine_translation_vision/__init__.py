"""MI355X-native drop-in for the hot path of ``machine_translation_vision`` (Eurus-Holmes/VAG-NMT).

Same import paths, class names, constructor/forward signatures and parameter names as the reference's
``models`` / ``layers`` / ``losses`` / ``utils.utils`` for the classes on the training/decoding path
(SURVEY.md section 8b); all arithmetic runs in hand-written HIP kernels (libvagnmt.so, include/vag_nmt.h).
There is no CPU fallback: CPU tensors, or a missing library, raise."""
