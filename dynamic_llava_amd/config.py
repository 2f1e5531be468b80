"""Configuration mirror of the reference's DynamicLlavaConfig (LlamaConfig + `sparse_config`).

Reference: llava/model/language_model/dynamic_llava_llama.py:39-40 (model_type "dynamic_llava_llama"),
llava/train/train_sparse.py:145-165 (SparseArguments -> config.sparse_config, persisted in config.json at
train_sparse.py:1007-1008).  Key names are kept so that a real checkpoint's config.json loads unchanged and
harness code such as `model.model.config.sparse_config["use_vision_predictor"] = True`
(llava/dynamic_eval/bench_test/dynamic_llava_image_time_and_mem.py:63-65) keeps working.
"""
from __future__ import annotations

import copy
import json
import os

IMAGE_TOKEN_INDEX = -200  # llava/constants.py:8
IGNORE_INDEX = -100  # llava/constants.py:7

DEFAULT_SPARSE_CONFIG = dict(
    use_vision_predictor=True,
    vision_keep_rate=0.2,
    use_text_predictor=True,
    use_output_text_predictor=True,
    output_text_keep_rate=0.5,
    output_text_len_for_training=50,
    use_instruct_predictor=False,
    instruct_keep_rate=0.7,
    instruct_len_for_training=25,
    sparse_layer=2,
    d_model=512,
    nhead=8,
    dim_feedforward=2048,
    num_layers=2,
    mask_loss_weight=100.0,
)

DEFAULT_CLIP = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336, patch_size=14)


class DynamicLlavaConfig:
    model_type = "dynamic_llava_llama"

    def __init__(
        self,
        hidden_size=4096,
        intermediate_size=11008,
        num_hidden_layers=32,
        num_attention_heads=32,
        num_key_value_heads=None,
        vocab_size=32000,
        max_position_embeddings=4096,
        rms_norm_eps=1e-5,
        rope_theta=10000.0,
        rope_scaling=None,
        hidden_act="silu",
        attention_bias=False,
        pad_token_id=0,
        bos_token_id=1,
        eos_token_id=2,
        use_cache=True,
        mm_vision_tower=None,
        mm_hidden_size=1024,
        mm_projector_type="mlp2x_gelu",
        mm_vision_select_layer=-2,
        mm_vision_select_feature="patch",
        mm_use_im_start_end=False,
        tokenizer_model_max_length=None,
        sparse_config=None,
        clip=None,
        **extra,
    ):
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads or num_attention_heads
        self.vocab_size = vocab_size
        self.max_position_embeddings = max_position_embeddings
        self.rms_norm_eps = rms_norm_eps
        self.rope_theta = rope_theta
        self.rope_scaling = rope_scaling
        self.hidden_act = hidden_act
        self.attention_bias = attention_bias
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.use_cache = use_cache
        self.mm_vision_tower = mm_vision_tower
        self.mm_hidden_size = mm_hidden_size
        self.mm_projector_type = mm_projector_type
        self.mm_vision_select_layer = mm_vision_select_layer
        self.mm_vision_select_feature = mm_vision_select_feature
        self.mm_use_im_start_end = mm_use_im_start_end
        self.tokenizer_model_max_length = tokenizer_model_max_length
        sc = copy.deepcopy(DEFAULT_SPARSE_CONFIG)
        sc.update(sparse_config or {})
        self.sparse_config = sc
        self.clip = dict(DEFAULT_CLIP, **(clip or {}))
        self.extra = extra
        if self.rope_scaling is not None:
            raise NotImplementedError("rope_scaling is None for LLaVA-1.5 (dynamic_modeling_llama.py:395-420); scaled RoPE is not built")
        if self.hidden_act != "silu" or self.attention_bias:
            raise NotImplementedError("only hidden_act='silu', attention_bias=False (LLaVA-1.5 / Vicuna) are supported")

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @property
    def n_image_tokens(self):
        return (self.clip["image_size"] // self.clip["patch_size"]) ** 2

    def to_dict(self):
        d = {k: v for k, v in self.__dict__.items() if k != "extra"}
        d["model_type"] = self.model_type
        return d

    @classmethod
    def from_dict(cls, d):
        d = dict(d)
        d.pop("model_type", None)
        d.pop("architectures", None)
        return cls(**d)

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_dict(json.load(f))

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2)

    @classmethod
    def from_namespace(cls, ns):
        """Builds from any attribute bag with the same field names (tests use oracle.fixtures configs)."""
        keys = (
            "hidden_size intermediate_size num_hidden_layers num_attention_heads num_key_value_heads vocab_size "
            "max_position_embeddings rms_norm_eps rope_theta mm_hidden_size mm_projector_type mm_vision_select_layer "
            "mm_vision_select_feature sparse_config clip"
        ).split()
        return cls(**{k: copy.deepcopy(getattr(ns, k)) for k in keys if hasattr(ns, k)})
