"""AUTHORING-CONTAINER ONLY: import the read-only Python reference from /root/reference.

Test infrastructure (never imported by the product path, never runs on the GPU box: the
reference does not travel).  Used to (a) validate the oracle restatement in oracle/ and
(b) generate the committed golden vectors under tests/golden/ (see oracle/make_goldens.py).

The reference's package __init__ imports torchvision (absent here), so the packages are
pre-registered as empty namespace modules and the sub-modules imported directly
(SURVEY.md Appendix B).  Nothing is copied: modules execute from /root/reference in place.
"""
import importlib
import os
import sys
import types

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "segment_anything_cs"))


def load_modeling():
    """Returns (modeling, build_sam, amg) reference modules (zero-shim tier O1)."""
    sys.dont_write_bytecode = True
    for name, path in [("segment_anything_cs", f"{REF}/segment_anything_cs"),
                       ("segment_anything_cs.utils", f"{REF}/segment_anything_cs/utils")]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    modeling = importlib.import_module("segment_anything_cs.modeling")
    build = importlib.import_module("segment_anything_cs.build_sam")
    amg = importlib.import_module("segment_anything_cs.utils.amg")
    return modeling, build, amg


def load_crowdhuman_eval():
    """tools/crowdhuman_eval.py is pure numpy: importable with zero shims."""
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_crowdhuman_eval", f"{REF}/tools/crowdhuman_eval.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
