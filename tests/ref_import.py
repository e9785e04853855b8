"""Import pieces of the Python reference from where they lie (/root/reference) -- TEST INFRASTRUCTURE, development
container only.  The GPU box has no /root/reference: everything here returns None there and the callers fall back to
the committed fixtures under tests/golden/ (generated with these same loaders by the make_golden_* scripts).

Nothing of the reference is copied: the modules are executed in place with the three imports the container lacks
(termcolor, visdom, ldm -> the `.attention` sibling) replaced by empty stand-ins that the loaded code never calls on the
path under test.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("MGS_REFERENCE_ROOT", "/root/reference")
MG = os.path.join(REF_ROOT, "agents", "manigaussian_bc")


def have_reference() -> bool:
    return os.path.isfile(os.path.join(MG, "resnetfc.py"))


def _stub(name, **attrs):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    return sys.modules[name]


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_resnetfc():
    """The reference's agents/manigaussian_bc/resnetfc.py (ResnetFC, ResnetBlockFC), or None without /root/reference.
    Its real utils.py is loaded too (combine_interleaved is on the path); termcolor / visdom / `.attention` (which needs
    the un-vendored `ldm`) are stand-ins."""
    if not have_reference():
        return None
    if "agents.manigaussian_bc.resnetfc" in sys.modules:
        return sys.modules["agents.manigaussian_bc.resnetfc"]
    _stub("termcolor", colored=lambda s, *a, **k: s, cprint=lambda *a, **k: None)
    _stub("visdom")
    pkg = _stub("agents")
    pkg.__path__ = []
    sub = _stub("agents.manigaussian_bc")
    sub.__path__ = []
    _stub("agents.manigaussian_bc.attention", Visual3DLangTransformer=object)
    _load("agents.manigaussian_bc.utils", os.path.join(MG, "utils.py"))
    return _load("agents.manigaussian_bc.resnetfc", os.path.join(MG, "resnetfc.py"))


def load_reference_render(path=None):
    """The reference's gaussian_renderer/__init__.py (render()), executed UNMODIFIED against this repository's drop-in
    `diff_gaussian_rasterization` package (it is the module the file imports by name).  `path`: an explicit copy
    (oracle/_ref/ref_gaussian_renderer.py travels to the GPU box); default: the file under /root/reference."""
    if path is None:
        path = os.path.join(MG, "gaussian_renderer", "__init__.py")
    if not os.path.isfile(path):
        return None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import diff_gaussian_rasterization  # noqa: F401  this repository's package, resolved by name like in ManiGaussian
    assert os.path.dirname(os.path.abspath(diff_gaussian_rasterization.__file__)).startswith(root)
    return _load("_mgs_reference_gaussian_renderer", path)
