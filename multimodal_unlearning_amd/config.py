"""Architecture + run configuration.  Mirrors the flag surface of the reference's config/mm/forget.yaml:1-57
and the per-family constants of config/model_config.yaml:59-65 (llava) that the forget step depends on."""
from __future__ import annotations

from dataclasses import asdict, dataclass, field


@dataclass
class VisionConfig:  # CLIP-ViT (transformers CLIPVisionConfig)
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_layers: int = 24
    num_heads: int = 16
    image_size: int = 336
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-5

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads

    @property
    def num_patches(self):
        return (self.image_size // self.patch_size) ** 2

    @property
    def patch_k(self):
        return self.num_channels * self.patch_size * self.patch_size

    @property
    def patch_k_padded(self):  # 16-byte rows for the im2col GEMM
        return (self.patch_k + 7) // 8 * 8


@dataclass
class TextConfig:  # Llama-2 (transformers LlamaConfig)
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_layers: int = 32
    num_heads: int = 32
    vocab_size: int = 32064
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_position: int = 4096

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


@dataclass
class LlavaConfig:
    vision: VisionConfig = field(default_factory=VisionConfig)
    text: TextConfig = field(default_factory=TextConfig)
    image_token_index: int = 32000
    vision_feature_layer: int = -2  # hidden_states[-2]  (configuration_llava.py:61)
    name: str = "llava-1.5-7b"

    @property
    def vision_layers_used(self) -> int:
        # hidden_states has L+1 entries; entry k is the output of k layers.
        L = self.vision.num_layers
        k = self.vision_feature_layer if self.vision_feature_layer >= 0 else L + 1 + self.vision_feature_layer
        return k

    def to_dict(self):
        return asdict(self)


def llava_1_5_7b() -> LlavaConfig:
    return LlavaConfig()


def llava_1_5_13b() -> LlavaConfig:
    return LlavaConfig(text=TextConfig(hidden_size=5120, intermediate_size=13824, num_layers=40, num_heads=40),
                       name="llava-1.5-13b")


def tiny_test_config(text_layers: int = 2, vision_layers: int = 3, vocab: int = 512) -> LlavaConfig:
    """Smallest config that still has the production head dims (Llama 128, CLIP 64): used by the parity tests
    and golden fixtures (28x28 image, patch 14 -> 4 image tokens)."""
    return LlavaConfig(
        vision=VisionConfig(hidden_size=128, intermediate_size=256, num_layers=vision_layers, num_heads=2,
                            image_size=28, patch_size=14),
        text=TextConfig(hidden_size=256, intermediate_size=512, num_layers=text_layers, num_heads=2,
                        vocab_size=vocab, max_position=512),
        image_token_index=vocab - 1, name="tiny")


@dataclass
class ForgetConfig:
    """config/mm/forget.yaml:1-57 (defaults copied from there)."""
    model_family: str = "llava"
    model_path: str = ""
    lora_r: int = 0
    lora_alpha: int = 32
    lora_dropout: float = 0.0
    lr: float = 1e-5
    batch_size: int = 2
    gradient_accumulation_steps: int = 4
    num_epochs: int = 5
    forget_loss: str = "grad_diff_forget_ce_retain_ce"
    loss_beta: float = 1.0
    l_norm_from: str = "zero"
    l1_lambda: float = 0.0
    l0_lambda: float = 0.0
    max_length: int = 2048
    weight_decay: float = 0.01
    seed: int = 42
    freeze_vision_module: bool = True
    max_grad_norm: float = 1.0  # HF TrainingArguments default (reference leaves it commented, mm/forget.py:151)
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_eps: float = 1e-8
