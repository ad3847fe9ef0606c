"""Config contract of the Qwen2-VL path, mirroring HF `Qwen2VLConfig` as the reference consumes it
(/root/reference/mantis/models/qwen2_vl/modeling_qwen2_vl.py:1 star-imports the HF model; /root/reference/mantis/train/
train_qwen2_vl.py:126-128 pixel budget, :209-212 frozen `visual`): same field names and nesting (vision_config / text_config dicts),
without the transformers dependency."""
from .configuration_llava import SubConfig

_VISION = dict(depth=32, embed_dim=1280, hidden_size=3584, hidden_act="quick_gelu", mlp_ratio=4, num_heads=16, in_channels=3,
               patch_size=14, spatial_merge_size=2, temporal_patch_size=2)
_TEXT = dict(model_type="qwen2_vl_text", hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
             num_key_value_heads=4, vocab_size=152064, rms_norm_eps=1e-6, hidden_act="silu", initializer_range=0.02,
             tie_word_embeddings=False, sliding_window=None, use_sliding_window=False,
             rope_parameters=dict(rope_type="default", rope_theta=1000000.0, mrope_section=[16, 24, 24]))


class Qwen2VLConfig:
    model_type = "qwen2_vl"

    def __init__(self, vision_config=None, text_config=None, image_token_id=151655, video_token_id=151656, vision_start_token_id=151652,
                 vision_end_token_id=151653, tie_word_embeddings=False, use_cache=False, **kwargs):
        def sub(defaults, given):
            d = dict(defaults)
            if isinstance(given, SubConfig):
                given = given.to_dict()
            d.update(given or {})
            return d
        v, t = sub(_VISION, vision_config), sub(_TEXT, text_config)
        if t.get("tie_word_embeddings") or tie_word_embeddings:
            raise NotImplementedError("tied embeddings (Qwen2-VL-2B) are not implemented; the 7B model of BASELINE configs[4] unties them")
        if t.get("use_sliding_window"):
            raise NotImplementedError("sliding-window attention (Qwen2-VL ships with use_sliding_window=False)")
        rp = dict(t.get("rope_parameters") or {})
        rs = t.get("rope_scaling") or {}
        if "mrope_section" in rs and "mrope_section" not in rp:     # transformers 4.x spelling
            rp["mrope_section"] = rs["mrope_section"]
        if "rope_theta" in t and "rope_theta" not in rp:
            rp["rope_theta"] = t["rope_theta"]
        if rp.get("rope_type", "default") not in ("default", "mrope"):
            raise NotImplementedError(f"rope_type {rp.get('rope_type')!r}: only the default multimodal RoPE is implemented")
        hd = t["hidden_size"] // t["num_attention_heads"]
        if sum(rp["mrope_section"]) != hd // 2:
            raise ValueError(f"mrope_section {rp['mrope_section']} must sum to head_dim / 2 = {hd // 2}")
        t["rope_parameters"] = rp
        t["rope_theta"] = float(rp["rope_theta"])
        t["head_dim"] = hd
        if v["hidden_size"] != t["hidden_size"]:
            raise ValueError("vision_config.hidden_size (merger output width) must equal text_config.hidden_size")
        if v["embed_dim"] % v["num_heads"] or (v["embed_dim"] // v["num_heads"]) % 4:
            raise ValueError("vision head_dim must be a multiple of 4 (2-D rotary embedding)")
        self.vision_config, self.text_config = SubConfig(**v), SubConfig(**t)
        self.image_token_id, self.video_token_id = image_token_id, video_token_id
        self.vision_start_token_id, self.vision_end_token_id = vision_start_token_id, vision_end_token_id
        self.vocab_size = t["vocab_size"]
        self.use_return_dict = kwargs.pop("use_return_dict", True)
        for k, val in kwargs.items():
            setattr(self, k, val)

    def to_dict(self):
        return dict(vision_config=self.vision_config.to_dict(), text_config=self.text_config.to_dict(), image_token_id=self.image_token_id,
                    video_token_id=self.video_token_id, vision_start_token_id=self.vision_start_token_id,
                    vision_end_token_id=self.vision_end_token_id)

    @classmethod
    def from_oracle_meta(cls, meta):
        """Build from the json blob stored in tests/golden/weights_qwen2vl.npz."""
        return cls(vision_config=dict(meta["vision"]), text_config=dict(meta["text"]), image_token_id=meta["image_token_id"],
                   video_token_id=meta["video_token_id"], vision_start_token_id=meta["vision_start_token_id"],
                   vision_end_token_id=meta["vision_end_token_id"])


def qwen2_vl_7b():
    """BASELINE.json configs[4]: Qwen2-VL-7B = 32-block ViT (1280 wide, 16 heads x 80, 2-D RoPE, 2x2 patch merger to 3584) + Qwen2-7B
    decoder (28 layers, 28/4 heads x 128, q/k/v bias, M-RoPE sections 16/24/24, vocabulary 152064)."""
    return Qwen2VLConfig()
