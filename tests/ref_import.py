"""Import pieces of the Python reference and EXECUTE them in place -- TEST INFRASTRUCTURE.

Where the files come from, in this order:
  1. /root/reference (the development container; $MGS_REFERENCE_ROOT overrides), or
  2. oracle/_ref/mg/ -- byte copies `make -C oracle` takes where /root/reference exists (git-ignored like the reference-kernel
     libraries next to them, they travel to the GPU box with the snapshot; nothing of the reference is committed).
Without either, every loader returns None and the callers fall back to the committed fixtures under tests/golden/
(generated with these same loaders by the make_golden_* scripts).

The modules are executed unmodified; the imports the container lacks (termcolor, visdom, dotmap, torchvision, and the
`ldm`-dependent `.attention` sibling) are empty stand-ins that the code under test never calls.
"""
import importlib.util
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CANDIDATES = [os.environ.get("MGS_REFERENCE_ROOT", "/root/reference"), os.path.join(ROOT, "oracle", "_ref", "mg")]
PKG = "agents.manigaussian_bc"


def _mg_dir():
    for r in _CANDIDATES:
        d = os.path.join(r, "agents", "manigaussian_bc")
        if os.path.isfile(os.path.join(d, "resnetfc.py")):
            return d
    return None


MG = _mg_dir() or os.path.join(_CANDIDATES[0], "agents", "manigaussian_bc")


def have_reference() -> bool:
    return _mg_dir() is not None


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


def _prepare():
    _stub("termcolor", colored=lambda s, *a, **k: s, cprint=lambda *a, **k: None)
    _stub("visdom")
    _stub("dotmap", DotMap=dict)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    pkg = _stub("agents")
    pkg.__path__ = []
    sub = _stub(PKG)
    sub.__path__ = []
    _stub(PKG + ".attention", Visual3DLangTransformer=object)


def _module(short, rel=None):
    """agents.manigaussian_bc.<short>, loaded from its file (once)."""
    full = f"{PKG}.{short}"
    if full in sys.modules and getattr(sys.modules[full], "__file__", None):
        return sys.modules[full]
    d = _mg_dir()
    if d is None:
        return None
    _prepare()
    return _load(full, os.path.join(d, rel or (short + ".py")))


def load_resnetfc():
    """The reference's agents/manigaussian_bc/resnetfc.py (ResnetFC, ResnetBlockFC); its real utils.py is loaded too
    (combine_interleaved is on the path)."""
    if _module("utils") is None:
        return None
    return _module("resnetfc")


def load_models_embed():
    """agents/manigaussian_bc/models_embed.py: GeneralizableGSEmbedNet (voxel gather + positional code, Gaussian regressor and
    its epilogue, deformation-field input assembly, MLP and apply -- SURVEY.md 8a rows a14-a16, 8f rows 2-3)."""
    if load_resnetfc() is None:
        return None
    return _module("models_embed")


def load_graphics_utils():
    """agents/manigaussian_bc/graphics_utils.py: getWorld2View2, getProjectionMatrix, focal2fov."""
    return _module("graphics_utils")


def load_neural_rendering():
    """agents/manigaussian_bc/neural_rendering.py (NeuralRenderer; get_novel_calib is the method under test).  It imports
    gaussian_renderer, i.e. `diff_gaussian_rasterization` -- this repository's drop-in package."""
    if load_models_embed() is None or load_graphics_utils() is None:
        return None
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    _module("loss")
    _module("gaussian_renderer", os.path.join("gaussian_renderer", "__init__.py"))
    return _module("neural_rendering")


def load_reference_render(path=None):
    """The reference's gaussian_renderer/__init__.py (render()), executed UNMODIFIED against this repository's drop-in
    `diff_gaussian_rasterization` package (it is the module the file imports by name).  `path`: an explicit copy
    (oracle/_ref/ref_gaussian_renderer.py travels to the GPU box); default: the file under the reference tree."""
    if path is None:
        d = _mg_dir()
        path = os.path.join(d, "gaussian_renderer", "__init__.py") if d else ""
    if not os.path.isfile(path):
        return None
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import diff_gaussian_rasterization  # noqa: F401  this repository's package, resolved by name like in ManiGaussian
    assert os.path.dirname(os.path.abspath(diff_gaussian_rasterization.__file__)).startswith(ROOT)
    return _load("_mgs_reference_gaussian_renderer", path)


class Cfg(dict):
    """The slice of an OmegaConf node the reference modules use: attribute and item access, nested."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    @staticmethod
    def of(d):
        return Cfg({k: Cfg.of(v) if isinstance(v, dict) else v for k, v in d.items()})


def method_cfg(d_hidden=512, use_dynamic_field=True, use_action=True, foundation_model_name=None, image=128):
    """conf/method/ManiGaussian_BC.yaml:89-162 (neural_renderer section), the keys GeneralizableGSEmbedNet / NeuralRenderer
    read; d_hidden is a parameter so that fixtures stay small."""
    return Cfg.of(dict(
        use_dynamic_field=use_dynamic_field, foundation_model_name=foundation_model_name, d_latent=128, d_lang=128,
        image_width=image, image_height=image, coordinate_bounds=[-0.3, -0.5, 0.6, 0.7, 0.5, 1.6], use_code=True,
        use_code_viewdirs=False, use_xyz=True,
        mlp=dict(n_blocks=5, d_hidden=d_hidden, combine_layer=3, combine_type="average", beta=0.0, use_spade=False,
                 opacity_scale=1.0, opacity_bias=-2.0, scale_bias=0.02, scale_scale=0.003, xyz_scale=0.1, xyz_bias=0.0,
                 max_sh_degree=1),
        next_mlp=dict(d_in=3, d_lang=128, d_out=3, n_blocks=5, d_hidden=d_hidden, combine_layer=3, combine_type="average",
                      beta=0.0, use_spade=False, warm_up=3000, use_action=use_action),
        code=dict(num_freqs=6, freq_factor=1.5, include_input=True),
        dataset=dict(bg_color=[0, 0, 0], znear=0.1, zfar=4.0, trans=[0.0, 0.0, 0.0], scale=1.0),
        d_embed=3, loss_embed_fn="cosine", lambda_embed=0.01, lambda_rgb=1.0))
