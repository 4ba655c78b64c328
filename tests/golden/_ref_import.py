"""Import the reference's in-tree Python modules (from /root/reference, THIS container only) to generate golden
vectors.  Missing third-party packages are replaced by empty stub modules (they are only imported, never executed,
on the code paths the fixtures exercise), except:
  * `diffusers`        -> a fake package with the mixin base classes / decorators wan_utils.py needs;
  * `torch_scatter`    -> scatter_add / scatter_max with real semantics (voxel fusion goldens).
Nothing here travels to the GPU box as executable reference code: only the produced tensors are committed."""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import sys
import types

import torch

REF = "/root/reference"
STUB_ROOTS = (
    "jaxtyping", "loguru", "gsplat", "torchvision", "omegaconf", "dacite", "cv2", "skvideo", "e3nn", "colorspacious",
    "lightning", "pytorch_lightning", "xformers", "plyfile", "imageio", "roma", "timm", "lpips", "open_clip", "wandb",
    "peft", "moviepy", "matplotlib", "PIL", "huggingface_hub",
)


class _Anything:
    """Attribute sink: any attribute / call / subscript yields another sink (type annotations, decorators)."""

    def __init__(self, name="stub"):
        self._n = name

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(self._n + "." + k)

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # used as a decorator
        return _Anything(self._n + "()")

    def __getitem__(self, k):
        return _Anything(self._n + "[]")

    def __mro_entries__(self, bases):
        return (object,)

    def __or__(self, o):
        return self

    __ror__ = __or__


class _StubModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(self.__name__ + "." + k)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if root in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _fake_diffusers():
    import torch.nn as nn
    d = types.ModuleType("diffusers")
    d.__path__ = []

    class ModelMixin(nn.Module):
        pass

    class ConfigMixin:
        pass

    class FromOriginalModelMixin:
        pass

    class AutoencoderKLWan(nn.Module):
        pass

    def register_to_config(f):
        return f

    def apply_forward_hook(f):
        return f

    def get_activation(name):
        return {"silu": nn.SiLU(), "gelu": nn.GELU(), "relu": nn.ReLU()}[name]

    d.AutoencoderKLWan = AutoencoderKLWan
    d.ModelMixin = ModelMixin
    d.ConfigMixin = ConfigMixin
    mods = {
        "diffusers": d,
        "diffusers.configuration_utils": dict(ConfigMixin=ConfigMixin, register_to_config=register_to_config),
        "diffusers.loaders": dict(FromOriginalModelMixin=FromOriginalModelMixin),
        "diffusers.loaders.single_file_model": dict(FromOriginalModelMixin=FromOriginalModelMixin),
        "diffusers.models": dict(),
        "diffusers.models.activations": dict(get_activation=get_activation),
        "diffusers.models.autoencoders": dict(),
        "diffusers.models.autoencoders.vae": dict(DecoderOutput=_Anything("DecoderOutput"),
                                                  DiagonalGaussianDistribution=_Anything("DGD")),
        "diffusers.models.modeling_outputs": dict(AutoencoderKLOutput=_Anything("AKLO")),
        "diffusers.models.modeling_utils": dict(ModelMixin=ModelMixin),
        "diffusers.utils": dict(logging=_Anything("logging")),
        "diffusers.utils.accelerate_utils": dict(apply_forward_hook=apply_forward_hook),
        "diffusers.pipelines": dict(),
        "diffusers.pipelines.wan": dict(),
        "diffusers.pipelines.wan.pipeline_wan": dict(prompt_clean=lambda s: s),
        "diffusers.schedulers": dict(),
        "diffusers.schedulers.scheduling_unipc_multistep": dict(UniPCMultistepScheduler=_Anything("UniPC")),
    }
    for name, attrs in mods.items():
        if isinstance(attrs, types.ModuleType):
            sys.modules[name] = attrs
            continue
        m = _StubModule(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    d.WanPipeline = _Anything("WanPipeline")


def _torch_scatter():
    m = types.ModuleType("torch_scatter")

    def scatter_add(src, index, dim=0, out=None, dim_size=None):
        n = int(dim_size if dim_size is not None else index.max().item() + 1)
        shape = list(src.shape)
        shape[dim] = n
        o = torch.zeros(shape, dtype=src.dtype, device=src.device) if out is None else out
        idx = index
        if idx.dim() != src.dim():
            view = [1] * src.dim()
            view[dim] = -1
            idx = idx.view(view).expand_as(src)
        return o.scatter_add_(dim, idx, src)

    def scatter_max(src, index, dim=0, out=None, dim_size=None):
        n = int(dim_size if dim_size is not None else index.max().item() + 1)
        shape = list(src.shape)
        shape[dim] = n
        o = torch.full(shape, float("-inf"), dtype=src.dtype, device=src.device)
        idx = index
        if idx.dim() != src.dim():
            view = [1] * src.dim()
            view[dim] = -1
            idx = idx.view(view).expand_as(src)
        o = o.scatter_reduce(dim, idx, src, reduce="amax", include_self=True)
        return o, None

    m.scatter_add, m.scatter_max = scatter_add, scatter_max
    sys.modules["torch_scatter"] = m


_installed = False


def install():
    global _installed
    if _installed:
        return
    _fake_diffusers()
    _torch_scatter()
    sys.meta_path.insert(0, _Finder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _installed = True
