"""dynamic_llava_amd -- MI355X-native (gfx950) implementation of Dynamic-LLaVA's sparsified prefill+decode
hot path behind the reference's Python API.  See DESIGN.md / INTEGRATION.md."""
from .config import IGNORE_INDEX, IMAGE_TOKEN_INDEX, DynamicLlavaConfig  # noqa: F401

__all__ = ["DynamicLlavaConfig", "IMAGE_TOKEN_INDEX", "IGNORE_INDEX"]


def __getattr__(name):  # lazy: importing the package must not require the .so
    if name in ("DynamicLlavaLlamaForCausalLM", "VisionPredictor", "TextPredictor"):
        from . import model

        return getattr(model, name)
    if name == "LlavaLlamaForCausalLM":  # the name upstream-LLaVA harnesses import (llava/model/__init__.py:1-13 exports both)
        from . import model

        return model.DynamicLlavaLlamaForCausalLM
    if name == "LlavaConfig":
        return DynamicLlavaConfig
    if name in ("load_pretrained_model", "build_random_model", "build_from_state_dict"):
        from . import builder

        return getattr(builder, name)
    if name == "KVSlabCache":
        from .cache import KVSlabCache

        return KVSlabCache
    raise AttributeError(name)
