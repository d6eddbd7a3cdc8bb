"""Launcher for the reference's own entry scripts on the MI355X hot path:

    python -m vagnmt_hip.run /path/to/VAG-NMT/nmt_multimodal_beam_DE.py --data_path ... (the script's own arguments)

``python script.py`` puts the script's directory FIRST on ``sys.path``, ahead of ``PYTHONPATH``, and that directory holds
the reference's ``machine_translation_vision`` package: a plain ``PYTHONPATH=vag-nmt_amd python nmt_multimodal_beam_DE.py``
therefore still imports the reference's classes.  This launcher orders the path -- this package's parent first, the
script's directory (the checkout: ``preprocessing``, ``train``, ``bleu``, and everything of ``machine_translation_vision``
that is off the hot path, see machine_translation_vision/_checkout.py) right behind it -- records the script's directory as
THE checkout (``VAG_REFERENCE_CHECKOUT``, unless the user set it) and runs the script unchanged as ``__main__``."""
import os
import runpy
import sys


def order_path(script):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # .../vag-nmt_amd
    sdir = os.path.dirname(os.path.abspath(script))
    rest = [p for p in sys.path if os.path.realpath(p or os.getcwd()) not in (os.path.realpath(here), os.path.realpath(sdir))]
    sys.path[:] = [here, sdir] + rest
    # where the shadow modules look for the checkout (train.py, machine_translation_vision/_checkout.py): the launched script's
    # own directory and nothing else -- not "the first hit on sys.path", which could be an unrelated train.py in the working
    # directory.  A user who keeps the script elsewhere names the checkout in VAG_REFERENCE_CHECKOUT.
    os.environ.setdefault("VAG_REFERENCE_CHECKOUT", sdir)
    stale = [m for m in sys.modules if m == "machine_translation_vision" or m.startswith("machine_translation_vision.")]
    for m in stale:                                                             # imported from the wrong place before us
        f = getattr(sys.modules[m], "__file__", None) or ""
        if not os.path.realpath(f).startswith(os.path.realpath(here)):
            del sys.modules[m]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m vagnmt_hip.run SCRIPT [script arguments]")
    order_path(argv[0])
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
