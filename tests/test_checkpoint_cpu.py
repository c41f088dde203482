import os

import torch

from multimodal_unlearning_amd.checkpoint import load_pretrained, save_pretrained
from multimodal_unlearning_amd.config import tiny_test_config
from multimodal_unlearning_amd.params import init_params


def test_safetensors_round_trip_sharded(tmp_path):
    cfg = tiny_test_config()
    P = init_params(cfg, seed=2)
    for style in ("hf4", "hf5_flat"):
        d = str(tmp_path / style)
        files = save_pretrained(cfg, P, d, style=style, max_shard_bytes=1 << 20)
        assert len(files) > 1 and all(f.endswith(".safetensors") for f in files)
        assert os.path.exists(os.path.join(d, "model.safetensors.index.json"))
        back = load_pretrained(cfg, d)
        for k in P:
            assert torch.equal(back[k].float(), P[k]), (style, k)  # init_params values are bf16-representable
