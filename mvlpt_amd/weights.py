"""Architecture table and deterministic synthetic frozen-CLIP weights.

Pretrained CLIP checkpoints cannot be downloaded here (clip/clip.py:41-70 needs network), and
neither parity nor throughput depends on trained values, so tests and `bench.py` use seeded
random weights with the reference's own initialisation scales (clip/model.py:209-217, 295-322)
under the reference's own ``state_dict`` key names.  The same function runs in the build
container (to feed the real reference when golden fixtures are generated) and on the GPU box.
"""
from __future__ import annotations

import hashlib
import math
from dataclasses import dataclass
from typing import Dict

import torch


@dataclass(frozen=True)
class ClipArch:
    name: str
    embed_dim: int
    image_resolution: int
    vision_layers: int
    vision_width: int
    vision_patch_size: int
    context_length: int
    vocab_size: int
    transformer_width: int
    transformer_heads: int
    transformer_layers: int

    @property
    def vision_heads(self) -> int:  # clip/model.py:268
        return self.vision_width // 64

    @property
    def grid(self) -> int:
        return self.image_resolution // self.vision_patch_size

    def ctor_args(self):
        """Positional arguments of clip.model.CLIP.__init__ (clip/model.py:240-253)."""
        return (self.embed_dim, self.image_resolution, self.vision_layers, self.vision_width,
                self.vision_patch_size, self.context_length, self.vocab_size,
                self.transformer_width, self.transformer_heads, self.transformer_layers)


ARCHS: Dict[str, ClipArch] = {
    "ViT-B/32": ClipArch("ViT-B/32", 512, 224, 12, 768, 32, 77, 49408, 512, 8, 12),
    "ViT-B/16": ClipArch("ViT-B/16", 512, 224, 12, 768, 16, 77, 49408, 512, 8, 12),
    "ViT-L/14": ClipArch("ViT-L/14", 768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12),
    "ViT-L/14@336px": ClipArch("ViT-L/14@336px", 768, 336, 24, 1024, 14, 77, 49408, 768, 12, 12),
    # test-size architecture: head_dim stays 64 as in every CLIP ViT (heads = width // 64)
    "tiny": ClipArch("tiny", 128, 32, 3, 128, 16, 77, 49408, 128, 2, 2),
}


def arch_from_state_dict(sd: Dict[str, torch.Tensor], name: str = "custom") -> ClipArch:
    """Same shape inference as clip.model.build_model (clip/model.py:395-418), ViT only."""
    vw = sd["visual.conv1.weight"].shape[0]
    vl = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    ps = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    tw = sd["ln_final.weight"].shape[0]
    tl = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")})
    vocab = sd["token_embedding.weight"].shape[0] if "token_embedding.weight" in sd else 49408
    return ClipArch(name, sd["text_projection"].shape[1], ps * grid, vl, vw, ps,
                    sd["positional_embedding"].shape[0], vocab, tw, tw // 64, tl)


def _randn(name: str, seed: int, shape, std: float, fp16_exact: bool = False) -> torch.Tensor:
    h = int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:7], "little")
    g = torch.Generator(device="cpu").manual_seed(h)
    w = torch.randn(tuple(shape), generator=g, dtype=torch.float32) * std
    # Real CLIP checkpoints store conv/linear/attention/projection tensors in fp16 (clip/model.py:371-392
    # convert_weights; "CLIP's default precision is fp16", trainers/mvlpt.py:849) and the fp32/amp modes only
    # up-cast them: such values are exactly representable in fp16.  Mirror that so that parity measures the
    # kernels' arithmetic and not a weight quantisation real checkpoints never undergo.
    return w.half().float() if fp16_exact else w


def make_state_dict(arch: ClipArch, seed: int = 0, *, include_token_embedding: bool = False,
                    randomize_affine: bool = True) -> Dict[str, torch.Tensor]:
    """fp32 CPU tensors keyed like clip.model.CLIP.state_dict().  With ``randomize_affine`` the
    LayerNorm scales/shifts and all biases are non-trivial so a dropped bias or γ shows in parity."""
    sd: Dict[str, torch.Tensor] = {}

    def affine(prefix: str, d: int):
        if randomize_affine:
            sd[prefix + ".weight"] = 1.0 + _randn(prefix + ".weight", seed, (d,), 0.1)
            sd[prefix + ".bias"] = _randn(prefix + ".bias", seed, (d,), 0.1)
        else:
            sd[prefix + ".weight"] = torch.ones(d)
            sd[prefix + ".bias"] = torch.zeros(d)

    def bias(name: str, d: int):
        sd[name] = _randn(name, seed, (d,), 0.02, True) if randomize_affine else torch.zeros(d)

    def tower(prefix: str, width: int, layers: int, init_width: int, init_layers: int):
        # clip/model.py:312-319 uses the TEXT transformer's width/layers for the text tower;
        # the vision tower keeps nn defaults there — we use the same formula with its own dims.
        proj_std = (init_width ** -0.5) * ((2 * init_layers) ** -0.5)
        attn_std = init_width ** -0.5
        fc_std = (2 * init_width) ** -0.5
        for l in range(layers):
            p = f"{prefix}resblocks.{l}."
            sd[p + "attn.in_proj_weight"] = _randn(p + "attn.in_proj_weight", seed, (3 * width, width), attn_std, True)
            bias(p + "attn.in_proj_bias", 3 * width)
            sd[p + "attn.out_proj.weight"] = _randn(p + "attn.out_proj.weight", seed, (width, width), proj_std, True)
            bias(p + "attn.out_proj.bias", width)
            affine(p + "ln_1", width)
            sd[p + "mlp.c_fc.weight"] = _randn(p + "mlp.c_fc.weight", seed, (4 * width, width), fc_std, True)
            bias(p + "mlp.c_fc.bias", 4 * width)
            sd[p + "mlp.c_proj.weight"] = _randn(p + "mlp.c_proj.weight", seed, (width, 4 * width), proj_std, True)
            bias(p + "mlp.c_proj.bias", width)
            affine(p + "ln_2", width)

    vw, p = arch.vision_width, arch.vision_patch_size
    scale = vw ** -0.5
    sd["visual.conv1.weight"] = _randn("visual.conv1.weight", seed, (vw, 3, p, p), (3 * p * p) ** -0.5, True)
    sd["visual.class_embedding"] = _randn("visual.class_embedding", seed, (vw,), scale)
    sd["visual.positional_embedding"] = _randn("visual.positional_embedding", seed, (arch.grid ** 2 + 1, vw), scale)
    affine("visual.ln_pre", vw)
    tower("visual.transformer.", vw, arch.vision_layers, vw, arch.vision_layers)
    affine("visual.ln_post", vw)
    sd["visual.proj"] = _randn("visual.proj", seed, (vw, arch.embed_dim), scale, True)

    tw = arch.transformer_width
    tower("transformer.", tw, arch.transformer_layers, tw, arch.transformer_layers)
    if include_token_embedding:
        sd["token_embedding.weight"] = _randn("token_embedding.weight", seed, (arch.vocab_size, tw), 0.02)
    sd["positional_embedding"] = _randn("positional_embedding", seed, (arch.context_length, tw), 0.01)
    affine("ln_final", tw)
    sd["text_projection"] = _randn("text_projection", seed, (tw, arch.embed_dim), tw ** -0.5, True)
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07), dtype=torch.float32)  # clip/model.py:291
    return sd
