"""Import shim for the *reference* (TEST INFRASTRUCTURE ONLY — never imported by the product path).

This file contains no reference code.  It pre-seeds ``sys.modules`` with stand-ins for the
third-party packages the reference imports but this image lacks (xformers, open_clip, fairscale,
rotary_embedding_torch, easydict, tyro, kiui, pynvml), registers bare package objects for
``tools`` / ``tools.modules`` / ``tools.modules.unet`` / ``tools.modules.diffusions`` so the
reference's package ``__init__`` files (which pull in cv2-based datasets) never run, and puts
``/root/reference`` on ``sys.path``.

It is used in exactly one place, in the authoring container only (``/root/reference`` does not exist on
the GPU box): ``oracle/make_golden.py``, which generates ``tests/golden/*.safetensors`` from the imported
reference (``tests/test_oracle_golden.py`` then holds the oracle to those fixtures everywhere).

The xformers stand-in maps ``memory_efficient_attention`` to exact softmax attention
(``F.scaled_dot_product_attention``): FMHA is exact attention up to rounding, SURVEY §8c.
"""
import os
import sys
import types
import importlib.machinery

REF_ROOT = os.environ.get("VMV_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "tools", "modules", "unet"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    m.__spec__.submodule_search_locations = [path]
    sys.modules[name] = m
    return m


class _AttrDict(dict):
    """easydict.EasyDict stand-in: attribute access + recursive wrapping."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(u) for u in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v


_installed = False


def install():
    """Idempotently install the stubs and make ``/root/reference`` importable."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference not found at {REF_ROOT}")
    import torch
    import torch.nn.functional as F

    sys.dont_write_bytecode = True

    def memory_efficient_attention(q, k, v, attn_bias=None, op=None, p=0.0, scale=None):
        causal = attn_bias is not None
        if q.dim() == 3:  # [B, M, K]
            return F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=scale)
        # [B, M, H, K] layout
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                           is_causal=causal, scale=scale)
        return o.transpose(1, 2)

    class LowerTriangularMask:  # noqa: D401 - marker only
        pass

    ops = _mod("xformers.ops", memory_efficient_attention=memory_efficient_attention,
               LowerTriangularMask=LowerTriangularMask, unbind=torch.unbind)
    xf = _mod("xformers", ops=ops)
    xf.__path__ = []
    _mod("open_clip")

    class RotaryEmbedding(torch.nn.Module):
        def __init__(self, *a, **kw):
            super().__init__()

        def rotate_queries_or_keys(self, x, *a, **kw):
            return x

    _mod("rotary_embedding_torch", RotaryEmbedding=RotaryEmbedding)
    fs = _mod("fairscale")
    fs.__path__ = []
    fsnn = _mod("fairscale.nn")
    fsnn.__path__ = []
    _mod("fairscale.nn.checkpoint", checkpoint_wrapper=lambda m, *a, **kw: m)
    _mod("easydict", EasyDict=_AttrDict)
    ty = _mod("tyro")
    ty.__path__ = []
    ty.extras = _mod("tyro.extras", subcommand_type_from_defaults=lambda *a, **kw: None)
    ki = _mod("kiui")
    ki.__path__ = []

    def safe_normalize(x, eps=1e-20):       # kiui.op.safe_normalize (third-party, absent): x / max(|x|, sqrt(eps))
        return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))

    ki.op = _mod("kiui.op", safe_normalize=safe_normalize)
    ki.lpips = _mod("kiui.lpips", LPIPS=type("LPIPS", (torch.nn.Module,), {}))
    _mod("roma")
    _mod("pynvml")

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _pkg("tools", os.path.join(REF_ROOT, "tools"))
    _pkg("tools.modules", os.path.join(REF_ROOT, "tools", "modules"))
    _pkg("tools.modules.unet", os.path.join(REF_ROOT, "tools", "modules", "unet"))
    _pkg("tools.modules.diffusions", os.path.join(REF_ROOT, "tools", "modules", "diffusions"))
    _installed = True


def load_lgm_reference():
    """core.unet / core.attention / core.options / core.utils / core.models of the reference (the LGM branch).
    ``core.gs`` needs the absent third-party rasteriser, so a placeholder with an inert ``GaussianRenderer`` is
    registered in its place: only ``LGM.forward_gaussians`` (U-Net + activations) is exercised."""
    install()
    import importlib

    _mod("core.gs", GaussianRenderer=lambda opt: None)
    ns = types.SimpleNamespace()
    ns.unet = importlib.import_module("core.unet")
    ns.options = importlib.import_module("core.options")
    ns.utils = importlib.import_module("core.utils")
    ns.models = importlib.import_module("core.models")
    return ns


def load_reference():
    """Return a namespace with the reference classes on the hot path."""
    install()
    import importlib

    ns = types.SimpleNamespace()
    ns.util = importlib.import_module("tools.modules.unet.util")
    ns.unet_t2v = importlib.import_module("tools.modules.unet.unet_t2v")
    ns.ddim = importlib.import_module("tools.modules.diffusions.diffusion_ddim")
    ns.schedules = importlib.import_module("tools.modules.diffusions.schedules")
    ns.autoencoder = importlib.import_module("tools.modules.autoencoder")
    return ns
