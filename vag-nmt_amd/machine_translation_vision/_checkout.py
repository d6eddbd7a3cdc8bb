"""Names of the reference package that are NOT on the hot path (the METEOR wrapper,
sixteen research model variants, their layers) are served from the user's own checkout of the reference: the directory of the
script ``python -m vagnmt_hip.run SCRIPT`` launched (recorded in ``VAG_REFERENCE_CHECKOUT``; a user may name another).
Nothing of the reference is copied: this module only extends package search paths.

* ``machine_translation_vision.meteor`` (imported by all four entry scripts, e.g. nmt_multimodal_beam_DE.py:14) resolves to
  the checkout's sub-package because the top package's ``__path__`` ends with the checkout's package directory.
* ``from machine_translation_vision.models import NMT_Seq2Seq_Beam, LIUMCVC_Seq2Seq_Beam`` (nmt_monomodal_beam_DE.py:15;
  neither class is instantiated by a default run) resolves through a module ``__getattr__`` that imports the checkout's
  file as a sub-module of OUR sub-package.  Its relative imports (``from ..layers import LIUMCVC_Encoder``) therefore see
  the HIP layers first and the checkout's layers for every name this package does not define.
* Without a checkout the same names resolve to placeholder classes that raise when instantiated, so that the import block
  of the scripts still executes and the hot-path classes stay usable.

Modules defined by this package always win: its directories come first in every ``__path__``."""
import importlib
import os
import sys

_PKG = "machine_translation_vision"
_HERE = os.path.dirname(os.path.abspath(__file__))

# class name -> file (without .py) in the reference's sub-package; the interface of the reference's __init__ files
# (models/__init__.py:1-18, layers/__init__.py:1-13), minus what this package implements itself
REFERENCE_NAMES = {
    "models": {n: n for n in (
        ["LIUMCVC_Seq2Seq", "LIUMCVC_Seq2Seq_Beam", "NMT_Seq2Seq", "NMT_Seq2Seq_Beam", "NMT_AttentionImagine_Seq2Seq",
         "NMT_AttentionImagine_Seq2Seq_Beam", "NMT_AttentionImagine_Seq2Seq_Beam_V12"]
        + ["NMT_AttentionImagine_Seq2Seq_Beam_V%d" % v for v in range(2, 11)])},
    "layers": dict({n: n for n in (
        "LIUMCVC_Decoder", "NMT_Decoder_V2", "NMT_Decoder_V3", "VSE_Imagine", "VSE_Imagine_Mean", "VSE_Imagine_Im",
        "VSE_Imagine_Text", "VSE_Imagine_Enc_Dec", "VSE_Imagine_Enc_Dec_V2")}, FF="ff"),
}

_checkout = False       # False: not searched yet; None: searched, none there; str: the checkout's package directory


def find_checkout():
    """Package directory of the reference checkout: ``$VAG_REFERENCE_CHECKOUT/machine_translation_vision`` -- the directory of the
    script ``python -m vagnmt_hip.run`` launched, or what the user named -- if it is not this package and has the reference's
    ``meteor`` sub-package or model variants; else None.  sys.path is not searched (round 6: whatever happened to come first on it
    was adopted)."""
    global _checkout
    if _checkout is not False:
        return _checkout
    _checkout = None
    root = os.environ.get("VAG_REFERENCE_CHECKOUT")
    if root:
        d = os.path.join(os.path.abspath(root), _PKG)
        if os.path.isfile(os.path.join(d, "__init__.py")) and os.path.realpath(d) != os.path.realpath(_HERE) and \
                (os.path.isdir(os.path.join(d, "meteor")) or os.path.isfile(os.path.join(d, "models", "NMT_Seq2Seq_Beam.py"))):
            _checkout = d
    return _checkout


def extend_path(path, *sub):
    """Append the checkout's directory for sub-package ``sub`` to a package ``__path__`` (ours stays first)."""
    root = find_checkout()
    if root is None:
        return path
    d = os.path.join(root, *sub)
    if os.path.isdir(d) and d not in path:
        path.append(d)
    return path


def _placeholder(sub, name):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "%s.%s.%s is not on the MI355X hot path (SURVEY.md section 8); put a checkout of the reference on sys.path "
            "AFTER vag-nmt_amd to use the reference's own implementation of it" % (_PKG, sub, name))
    return type(name, (object,), {"__init__": __init__, "__module__": "%s.%s" % (_PKG, sub), "_vag_placeholder": True,
                                  "__doc__": "placeholder for the reference class of this name (no checkout on sys.path)"})


def resolve(sub, name):
    """Module ``__getattr__`` of sub-package ``sub``: the reference class ``name`` from the checkout, else a placeholder."""
    table = REFERENCE_NAMES.get(sub, {})
    if name not in table:
        raise AttributeError("module %r has no attribute %r" % ("%s.%s" % (_PKG, sub), name))
    pkg = sys.modules["%s.%s" % (_PKG, sub)]
    obj = None
    if find_checkout() is not None:
        full = "%s.%s" % (pkg.__name__, table[name])
        try:
            obj = getattr(importlib.import_module(full), name)
        except ModuleNotFoundError as e:          # a checkout that lacks this file (a trimmed copy): placeholder, like no checkout
            if e.name != full:
                raise
    if obj is None:
        obj = _placeholder(sub, name)
    setattr(pkg, name, obj)
    return obj
