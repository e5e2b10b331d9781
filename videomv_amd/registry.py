"""Name -> class/function registries with the reference's plugin API (utils/registry.py:24-155,
utils/registry_class.py:9-18): ``REG.register_class()``, ``REG.register_function()``, ``REG.build(dict(type=...))``.
Behaviour kept: ``type`` is popped from a deep copy of the config, extra kwargs override config keys, unknown
names raise ``KeyError``, and constructor failures are re-raised as a bare ``Exception("Failed to init class ...")``.
"""
import copy
import inspect
import warnings


class Registry:
    def __init__(self, name, build_func=None, allow_types=("class", "function")):
        self.name = name
        self.allow_types = allow_types
        self.class_map = {}
        self.func_map = {}
        self.build_func = build_func or build_from_config

    def get(self, req_type):
        return self.class_map.get(req_type) or self.func_map.get(req_type)

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register(self, obj, name, want_class):
        kind = "class" if want_class else "function"
        ok = inspect.isclass(obj) if want_class else inspect.isfunction(obj)
        if not ok:
            raise TypeError(f"{self.name}: expected a {kind}, got {type(obj)}")
        if kind not in self.allow_types:
            raise TypeError(f"Register {self.name} only allows type {self.allow_types}, got {kind}")
        table = self.class_map if want_class else self.func_map
        key = name or obj.__name__
        if key in table:
            warnings.warn(f"{kind} {key} already registered in {self.name}; replacing")
        table[key] = obj
        return obj

    def register_class(self, name=None):
        return lambda cls: self._register(cls, name, True)

    def register_function(self, name=None):
        return lambda fn: self._register(fn, name, False)

    def __repr__(self):
        keys = sorted(list(self.class_map) + list(self.func_map))
        return f"Registry [{self.name}]: " + ", ".join(keys)


def build_from_config(cfg, registry, **kwargs):
    if not isinstance(cfg, dict):
        raise TypeError(f"config must be type dict, got {type(cfg)}")
    if "type" not in cfg:
        raise KeyError(f"config must contain key type, got {cfg}")
    if not isinstance(registry, Registry):
        raise TypeError(f"registry must be type Registry, got {type(registry)}")
    cfg = copy.deepcopy(cfg)
    entry = cfg.pop("type")
    if isinstance(entry, str):
        found = registry.get(entry)
        if found is None:
            raise KeyError(f"{entry} not found in {registry.name} registry")
        entry = found
    if kwargs:
        cfg.update(kwargs)
    if inspect.isclass(entry):
        try:
            return entry(**cfg)
        except Exception as e:
            raise Exception(f"Failed to init class {entry}, with {e}")
    if inspect.isfunction(entry):
        try:
            return entry(**cfg)
        except Exception as e:
            raise Exception(f"Failed to invoke function {entry}, with {e}")
    raise TypeError(f"type must be str or class, got {type(entry)}")


AUTO_ENCODER = Registry("AUTO_ENCODER")
DATASETS = Registry("DATASETS")
DIFFUSION = Registry("DIFFUSION")
DISTRIBUTION = Registry("DISTRIBUTION")
EMBEDDER = Registry("EMBEDDER")
ENGINE = Registry("ENGINE")
INFER_ENGINE = Registry("INFER_ENGINE")
MODEL = Registry("MODEL")
PRETRAIN = Registry("PRETRAIN")
VISUAL = Registry("VISUAL")
