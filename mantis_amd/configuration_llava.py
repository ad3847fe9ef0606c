"""Config contract of the hot path, mirroring the reference's `LlavaConfig`
(/root/reference/mantis/models/mllava/configuration_llava.py:32-134): same field names, defaults and nesting, without the
transformers dependency.  `vision_config` / `text_config` accept dicts (HF config.json style) or `SubConfig` objects."""
import json


class SubConfig:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to_dict(self):
        return dict(self.__dict__)

    def get(self, k, default=None):
        return self.__dict__.get(k, default)

    def __repr__(self):
        return f"SubConfig({self.__dict__})"


_VISION_DEFAULTS = dict(model_type="clip_vision_model", hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                        num_attention_heads=16, image_size=336, patch_size=14, num_channels=3, hidden_act="quick_gelu",
                        layer_norm_eps=1e-5)
_SIGLIP_DEFAULTS = dict(hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)
_TEXT_DEFAULTS = dict(model_type="llama", hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                      num_attention_heads=32, num_key_value_heads=None, vocab_size=32000, rope_theta=10000.0,
                      rms_norm_eps=1e-6, hidden_act="silu", head_dim=None, initializer_range=0.02, attention_bias=False,
                      mlp_bias=False, tie_word_embeddings=False)


class LlavaConfig:
    model_type = "llava"

    def __init__(self, vision_config=None, text_config=None, ignore_index=-100, image_token_index=32000,
                 projector_hidden_act="gelu", vision_feature_select_strategy="default", vision_feature_layer=-2,
                 vocab_size=32000, pad_token_id=None, fix_unequal_counts=False, **kwargs):
        # Not a reference field.  False: image rows are placed exactly as the reference places them, including its mis-placement for a
        # right-padded batch whose samples hold different numbers of images (modeling_llava.py:343-345; the reference never sees such
        # a batch: processing_llava.py:277-285 asserts bs = 1).  True: every image lands on its own <image> token's span whatever the
        # padding side, i.e. each sample gets what the reference computes for it alone.
        self.fix_unequal_counts = bool(fix_unequal_counts)
        self.ignore_index = ignore_index
        self.image_token_index = image_token_index
        self.projector_hidden_act = projector_hidden_act
        self.vision_feature_select_strategy = vision_feature_select_strategy
        self.vision_feature_layer = vision_feature_layer
        self.vocab_size = vocab_size
        self.pad_token_id = pad_token_id
        v = dict(_VISION_DEFAULTS)
        if isinstance(vision_config, SubConfig):
            vision_config = vision_config.to_dict()
        if vision_config:
            if vision_config.get("model_type") == "siglip_vision_model":
                v.update(_SIGLIP_DEFAULTS)
            v.update(vision_config)
        self.vision_config = SubConfig(**v)
        t = dict(_TEXT_DEFAULTS)
        if isinstance(text_config, SubConfig):
            text_config = text_config.to_dict()
        if text_config:
            t.update(text_config)
            rp = text_config.get("rope_parameters") or {}
            if "rope_theta" in rp:
                t["rope_theta"] = rp["rope_theta"]
            # only the default rotary embedding is built (HF:models/llama/modeling_llama.py:95-110); llama3 / linear / dynamic /
            # yarn scaling would silently change the numerics, so refuse instead of ignoring
            scaling = text_config.get("rope_scaling")
            kinds = [rp.get("rope_type"), rp.get("type")] + ([scaling.get("rope_type"), scaling.get("type")] if scaling else [])
            if any(k not in (None, "default") for k in kinds) or (scaling and not any(kinds)):
                raise NotImplementedError(f"rope scaling {scaling or rp!r}: only the default RoPE is implemented on this path")
            self.vocab_size = t["vocab_size"]          # configuration_llava.py:125
        if t["num_key_value_heads"] is None:
            t["num_key_value_heads"] = t["num_attention_heads"]
        if t["head_dim"] is None:
            t["head_dim"] = t["hidden_size"] // t["num_attention_heads"]
        if t["model_type"] != "llama":
            raise NotImplementedError(f"text backbone {t['model_type']!r}: only the Llama family is built (SURVEY.md section 8)")
        if t.get("attention_bias") or t.get("mlp_bias") or t.get("tie_word_embeddings"):
            raise NotImplementedError("attention_bias / mlp_bias / tied embeddings are not used by the Mantis Llama-3 path")
        self.text_config = SubConfig(**t)
        self.use_return_dict = kwargs.pop("use_return_dict", True)
        self.output_attentions = False
        self.output_hidden_states = False
        for k, val in kwargs.items():
            setattr(self, k, val)

    def to_dict(self):
        d = {k: v for k, v in self.__dict__.items() if not isinstance(v, SubConfig)}
        d["vision_config"] = self.vision_config.to_dict()
        d["text_config"] = self.text_config.to_dict()
        return d

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2)

    @classmethod
    def from_oracle_meta(cls, meta):
        """Build from the json blob stored in tests/golden/weights_*.npz."""
        return cls(vision_config=dict(meta["vision"]), text_config=dict(meta["text"]),
                   image_token_index=meta["image_token_index"], pad_token_id=meta["pad_token_id"],
                   vocab_size=meta["vocab_size"], vision_feature_select_strategy=meta["vision_feature_select_strategy"],
                   vision_feature_layer=meta.get("vision_feature_layer", -2),
                   projector_hidden_act=meta.get("projector_hidden_act", "gelu"), ignore_index=meta.get("ignore_index", -100))


# Named geometries used by bench.py / tests (SURVEY.md section 8 "config shorthand").
def mantis_8b_siglip_llama3():
    """cfg2/cfg3: SigLIP-so400m/14 geometry at 336^2 + Llama-3-8B (+2 added tokens: <image>=128256, <|pad|>=128257,
    /root/reference/mantis/train/train_mllava.py:158-166)."""
    return LlavaConfig(
        vision_config=dict(model_type="siglip_vision_model", hidden_size=1152, intermediate_size=4304, num_hidden_layers=27,
                           num_attention_heads=16, image_size=336, patch_size=14),
        text_config=dict(model_type="llama", hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                         num_attention_heads=32, num_key_value_heads=8, vocab_size=128258, rope_theta=500000.0,
                         rms_norm_eps=1e-5),
        image_token_index=128256, pad_token_id=128257, vocab_size=128258, vision_feature_select_strategy="full")


def mantis_8b_clip_llama3():
    """The scripts' default tower (/root/reference/mantis/train/scripts/pretrain_mllava.sh:34, openai/clip-vit-large-patch14-336:
    CLS token, pre-LN, quick_gelu, eps 1e-5, "default" select strategy = drop CLS) + Llama-3-8B; SURVEY appendix B."""
    return LlavaConfig(
        vision_config=dict(model_type="clip_vision_model", hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                           num_attention_heads=16, image_size=336, patch_size=14),
        text_config=dict(model_type="llama", hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                         num_attention_heads=32, num_key_value_heads=8, vocab_size=128258, rope_theta=500000.0,
                         rms_norm_eps=1e-5),
        image_token_index=128256, pad_token_id=128257, vocab_size=128258, vision_feature_select_strategy="default")


def mantis_tiny():
    """cfg1: SigLIP-base/16-224 + Llama-68M (V = 32000 + 2)."""
    return LlavaConfig(
        vision_config=dict(model_type="siglip_vision_model", hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                           num_attention_heads=12, image_size=224, patch_size=16),
        text_config=dict(model_type="llama", hidden_size=768, intermediate_size=3072, num_hidden_layers=2,
                         num_attention_heads=12, num_key_value_heads=12, vocab_size=32002, rope_theta=10000.0,
                         rms_norm_eps=1e-6),
        image_token_index=32000, pad_token_id=32001, vocab_size=32002, vision_feature_select_strategy="full")
