"""Config contract of the Idefics2 path, mirroring HF `Idefics2Config` as the reference fork consumes it
(/root/reference/mantis/models/idefics2/modeling_idefics2.py:166-183 vision, :816-821 perceiver, :1488-1504 model): same field names
and nesting (vision_config / perceiver_config / text_config dicts), without the transformers dependency."""
from .configuration_llava import SubConfig

_VISION = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, num_channels=3, image_size=224,
               patch_size=32, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6, attention_dropout=0.0)
_PERCEIVER = dict(hidden_act="silu", resampler_n_latents=64, resampler_depth=3, resampler_n_heads=16, resampler_head_dim=96,
                  num_key_value_heads=4, attention_dropout=0.0)
_TEXT = dict(model_type="mistral", hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
             num_key_value_heads=8, vocab_size=32000, rope_theta=10000.0, rms_norm_eps=1e-5, hidden_act="silu", head_dim=None,
             sliding_window=None, initializer_range=0.02, tie_word_embeddings=False, attention_bias=False, mlp_bias=False)


class Idefics2Config:
    model_type = "idefics2"

    def __init__(self, vision_config=None, perceiver_config=None, text_config=None, image_token_id=32001, tie_word_embeddings=False,
                 use_cache=False, **kwargs):
        def sub(defaults, given):
            d = dict(defaults)
            if isinstance(given, SubConfig):
                given = given.to_dict()
            d.update(given or {})
            return d
        v, p, t = sub(_VISION, vision_config), sub(_PERCEIVER, perceiver_config), sub(_TEXT, text_config)
        if t["model_type"] not in ("mistral", "llama"):
            raise NotImplementedError(f"text backbone {t['model_type']!r}: the Idefics2 path is built for the Mistral / Llama block")
        if t.get("attention_bias") or t.get("mlp_bias") or t.get("tie_word_embeddings") or tie_word_embeddings:
            raise NotImplementedError("attention_bias / mlp_bias / tied embeddings are not used by Idefics2-8B")
        if t.get("num_key_value_heads") is None:
            t["num_key_value_heads"] = t["num_attention_heads"]
        if t.get("head_dim") is None:
            t["head_dim"] = t["hidden_size"] // t["num_attention_heads"]
        rp = t.get("rope_parameters") or {}
        if "rope_theta" in rp:
            t["rope_theta"] = rp["rope_theta"]
        if t.get("rope_scaling") or rp.get("rope_type") not in (None, "default"):
            raise NotImplementedError("only the default RoPE is implemented on this path")
        if p["attention_dropout"] or v["attention_dropout"]:
            raise NotImplementedError("attention dropout is not implemented (Idefics2-8B uses 0.0)")
        if t.get("sliding_window") not in (None, 0) and t["sliding_window"] < 1 << 30:
            # Mistral's 4096-token window: identical to full causal attention for sequences up to the window
            self.sliding_window = int(t["sliding_window"])
        else:
            self.sliding_window = None
        self.vision_config, self.perceiver_config, self.text_config = SubConfig(**v), SubConfig(**p), SubConfig(**t)
        self.image_token_id = image_token_id
        self.vocab_size = t["vocab_size"]
        self.use_return_dict = kwargs.pop("use_return_dict", True)
        for k, val in kwargs.items():
            setattr(self, k, val)

    def to_dict(self):
        return dict(vision_config=self.vision_config.to_dict(), perceiver_config=self.perceiver_config.to_dict(),
                    text_config=self.text_config.to_dict(), image_token_id=self.image_token_id)

    @classmethod
    def from_oracle_meta(cls, meta):
        """Build from the json blob stored in tests/golden/weights_idefics2.npz."""
        return cls(vision_config=dict(meta["vision"]), perceiver_config=dict(meta["perceiver"]), text_config=dict(meta["text"]),
                   image_token_id=meta["image_token_id"])


def mantis_8b_idefics2():
    """BASELINE.json configs[3]: Mantis-8B-Idefics2 = SigLIP-so400m/14 NaViT (980-px canvas: 70 x 70 position buckets) + perceiver
    resampler (64 latents, depth 3, 16/4 heads x 96) + Mistral-7B (+3 added tokens: <fake_token_around_image> 32000, <image> 32001,
    <end_of_utterance> 32002)."""
    return Idefics2Config(
        vision_config=dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16, image_size=980,
                           patch_size=14),
        perceiver_config=dict(resampler_n_latents=64, resampler_depth=3, resampler_n_heads=16, resampler_head_dim=96, num_key_value_heads=4),
        text_config=dict(model_type="mistral", hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                         num_key_value_heads=8, vocab_size=32003, rope_theta=10000.0, rms_norm_eps=1e-5),
        image_token_id=32001)
