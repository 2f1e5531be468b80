"""TEST INFRASTRUCTURE ONLY -- loads the *reference* (Osilly/dynamic_llava) from /root/reference.

Only usable in the build container (the reference never travels to the GPU box). Used by
`oracle/make_golden.py` to generate the committed golden vectors under tests/golden/ and by
`tests/test_oracle_vs_reference.py` (skipped when /root/reference is absent).

Shims: transformers 5.x removed `is_torch_fx_available` (used at
llava/model/language_model/dynamic_modeling_llama.py:53,75-79) and folds rope_theta / rope_scaling
into rope_parameters (read at dynamic_modeling_llama.py:368,396); llava/__init__.py:1 hard-imports the
whole package, so empty package modules are pre-registered with __path__ pointing into the reference.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("DYNLLAVA_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "llava", "model", "language_model"))


def import_reference():
    """Returns the reference module `llava.model.language_model.dynamic_llava_llama`."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True  # never write .pyc into the read-only reference tree
    import transformers.utils as tu
    import transformers.utils.import_utils as iu

    if not hasattr(iu, "is_torch_fx_available"):
        iu.is_torch_fx_available = lambda: False
    if not hasattr(tu, "is_torch_fx_available"):
        tu.is_torch_fx_available = lambda: False
    for name in (
        "llava",
        "llava.model",
        "llava.model.language_model",
        "llava.model.multimodal_encoder",
        "llava.model.multimodal_projector",
    ):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF_ROOT, *name.split("."))]
            sys.modules[name] = m
    from llava.model.language_model import dynamic_llava_llama as dll  # noqa: E402

    return dll
