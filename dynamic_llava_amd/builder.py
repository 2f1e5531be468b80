"""Model construction: the reference's `load_pretrained_model` surface
(llava/model/dynamic_llava_builder.py:35-249) plus random-init builders for benches and tests.

    tokenizer, model, image_processor, context_len = load_pretrained_model(model_path, model_base, model_name)

Out of scope (fail loudly): 8-bit/4-bit bitsandbytes loading (BLD:51-62), LoRA merging (BLD:70-140),
`device_map="auto"` layer placement -- one process drives one GPU, weights are replicated per GPU.
"""
from __future__ import annotations

import glob
import os

import torch

from . import hip_ops as ops
from .config import DynamicLlavaConfig
from .model import DynamicLlavaLlamaForCausalLM


def _construct(cfg, dtype, device, with_vision_tower=True):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            model = DynamicLlavaLlamaForCausalLM(cfg, with_vision_tower=with_vision_tower)
    finally:
        torch.set_default_dtype(prev)
    return model


@torch.no_grad()
def build_random_model(cfg: DynamicLlavaConfig, dtype=torch.bfloat16, device="cuda", seed=0, init_std=None, predictor_gain=1.0):
    """Random-init model of the given architecture directly on the GPU (HF `_init_weights`:
    dynamic_modeling_llama.py:1492-1501 -> N(0, initializer_range=0.02) for Linear / Embedding, zero bias)."""
    ops.require_gpu()
    model = _construct(cfg, dtype, device)
    g = torch.Generator(device=device).manual_seed(seed)
    std = 0.02 if init_std is None else init_std
    for name, p in model.named_parameters():  # every parameter comes from the seeded generator: all DP ranks hold identical weights
        if p.dim() >= 2 or "embedding" in name:
            p.normal_(0.0, std, generator=g)
        elif name.endswith("bias"):
            p.zero_()
        elif "vision_tower" in name and ("norm" in name or "layrnorm" in name) and name.endswith("weight"):
            p.fill_(1.0)
    if predictor_gain != 1.0:
        m = model.model
        if hasattr(m, "image_score_predictor"):
            m.image_score_predictor.output_mlp[4].weight.mul_(predictor_gain)
        if hasattr(m, "output_text_score_predictor"):
            m.output_text_score_predictor.output_mlp[7].weight.mul_(predictor_gain)
    return model.finalize()


@torch.no_grad()
def build_from_state_dict(cfg: DynamicLlavaConfig, state_dict, clip_state_dict=None, dtype=torch.bfloat16, device="cuda"):
    """Model with the reference's state-dict keys (model.layers.*, model.image_score_predictor.*, ...).
    `clip_state_dict`: a transformers CLIPVisionModel state dict (keys relative to that module)."""
    ops.require_gpu()
    model = _construct(cfg, dtype, device, with_vision_tower=True)
    sd = {k: v.to(device=device, dtype=dtype) for k, v in state_dict.items() if "vision_tower" not in k}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "vision_tower" not in m]
    if missing or unexpected:
        raise RuntimeError(f"state dict mismatch: missing={missing[:8]} unexpected={unexpected[:8]}")
    if clip_state_dict is not None:
        model.model.vision_tower.vision_tower.load_state_dict({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in clip_state_dict.items()})
    return model.finalize()


def _load_checkpoint_tensors(model_path):
    files = sorted(glob.glob(os.path.join(model_path, "*.safetensors")))
    if files:
        from safetensors.torch import load_file

        for f in files:
            yield from load_file(f).items()
        return
    files = sorted(glob.glob(os.path.join(model_path, "pytorch_model*.bin")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {model_path}")
    for f in files:
        yield from torch.load(f, map_location="cpu", weights_only=True).items()


@torch.no_grad()
def load_pretrained_model(model_path, model_base=None, model_name=None, load_8bit=False, load_4bit=False, device_map="auto", device="cuda", use_flash_attn=False, torch_dtype=None, **kwargs):
    """dynamic_llava_builder.py:35-249.  Returns (tokenizer, model, image_processor, context_len)."""
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes quantised loading (BLD:51-62) is out of scope for the MI355X hot path")
    if model_base is not None:
        raise NotImplementedError("LoRA / projector-only checkpoints with a model_base (BLD:70-180) are out of scope")
    if device != "cuda":
        raise ops.HipOpsError("dynamic_llava_amd has no CPU path: device must be 'cuda' (an MI355X)")
    ops.require_gpu()
    dtype = torch_dtype or torch.float16  # the reference's eval loaders use fp16 (BLD:62)
    cfg = DynamicLlavaConfig.from_pretrained(model_path)
    model = _construct(cfg, dtype, device)
    own = dict(model.state_dict())
    seen = set()
    for k, v in _load_checkpoint_tensors(model_path):
        if k in own:
            own[k].copy_(v.to(dtype) if v.is_floating_point() else v)
            seen.add(k)
    missing = [k for k in own if k not in seen and "vision_tower" not in k]
    if missing:
        raise RuntimeError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:5]}")
    vt_path = cfg.mm_vision_tower
    image_processor = None
    cache_dir = kwargs.get("cache_dir")
    if vt_path:
        # BLD:237-242 + clip_encoder.py:22-38: the tower (and its image processor) is loaded BY NAME -- a local directory or a hub id such as
        # "openai/clip-vit-large-patch14-336" (what the released checkpoints' config.json says).  There is no network here, so a hub id must resolve from
        # the local HF cache (`local_files_only=True`; `cache_dir=` / HF_HOME select it); a checkpoint that carries the tower's tensors itself is accepted too.
        from transformers import CLIPImageProcessor, CLIPVisionModel

        local = os.path.isdir(str(vt_path))
        ckpt_has_tower = any("vision_tower" in k for k in seen)
        try:
            #  a checkpoint that carries the tower's own (possibly fine-tuned: unfreeze_mm_vision_tower) tensors keeps them -- in the reference
            # the checkpoint's tensors win because from_pretrained runs BEFORE load_model's copy (BLD:237-242) -- and only the image processor is loaded by name
            clip = None if ckpt_has_tower else CLIPVisionModel.from_pretrained(vt_path, local_files_only=not local, cache_dir=cache_dir)
            image_processor = CLIPImageProcessor.from_pretrained(vt_path, local_files_only=not local, cache_dir=cache_dir)
        except (OSError, EnvironmentError) as e:  # not found locally / not in the cache: re-raised below unless the checkpoint itself holds the tower
            if not ckpt_has_tower:
                raise FileNotFoundError(
                    f"vision tower {vt_path!r} (config.mm_vision_tower) is neither a local directory nor present in the local Hugging Face cache "
                    f"(cache_dir={cache_dir!r}, HF_HOME={os.environ.get('HF_HOME')!r}); there is no network to fetch it from, and the checkpoint holds no vision_tower tensors") from e
            clip = None
        if clip is not None:
            model.model.vision_tower.vision_tower.load_state_dict(clip.state_dict())
            model.model.vision_tower.to(device=device, dtype=dtype)
    elif not any("vision_tower" in k for k in seen):
        raise FileNotFoundError("vision tower weights not found: config.mm_vision_tower is empty and the checkpoint holds no vision_tower tensors")
    # BLD:45-49 (AutoTokenizer.from_pretrained(model_path, use_fast=False)): a checkpoint without a usable tokenizer is an error in the reference and here.
    # require_tokenizer=False (tests / weight-only checkpoints): the failure becomes a warning and `tokenizer` is None.
    from transformers import AutoTokenizer

    try:
        tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=False)
    except Exception as e:  # noqa: BLE001
        if kwargs.get("require_tokenizer", True):
            raise
        import warnings

        warnings.warn(f"load_pretrained_model: no tokenizer could be loaded from {model_path!r} ({type(e).__name__}: {e}); returning tokenizer=None", RuntimeWarning)
        tokenizer = None
    model.finalize()
    context_len = cfg.extra.get("max_sequence_length", 2048)
    return tokenizer, model, image_processor, context_len


def save_pretrained(model: DynamicLlavaLlamaForCausalLM, path):
    """Writes config.json + model.safetensors with the reference's key names (round-trips through load_pretrained_model)."""
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    model.config.save_pretrained(path)
    save_file({k: v.detach().to("cpu").contiguous().clone() for k, v in model.state_dict().items()}, os.path.join(path, "model.safetensors"))
