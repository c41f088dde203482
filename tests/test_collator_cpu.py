"""Collator / pairing / chat-format parity against goldens generated from the reference's own
mm/dataset.py functions on a real offline LlavaProcessor (oracle/make_golden_collator.py).  CPU: token ids, masks,
labels and the uint8 resize/crop (through the numpy restatement of the processor's rescale+normalise);
the HIP normalise kernel is checked against the same goldens in tests/test_kernels_gpu.py."""
import json
import os
import random

import numpy as np
import pytest
import torch

from multimodal_unlearning_amd import collator as C
from oracle.tiny_tokenizer import adapter, pixel_digest, test_image as make_image

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "collator.npz")), json.load(open(os.path.join(G, "collator_pairs.json")))


def normalize_ref(u8):
    """numpy restatement of transformers' rescale (float64 product -> fp32) + normalize (fp32 subtract, divide)."""
    r = (u8.astype(np.float64) * (1 / 255)).astype(np.float32)
    m = np.array(C.OPENAI_CLIP_MEAN, dtype=np.float32).reshape(1, 3, 1, 1)
    s = np.array(C.OPENAI_CLIP_STD, dtype=np.float32).reshape(1, 3, 1, 1)
    return (r - m) / s


def realise(meta, v):
    if isinstance(v, list):
        return [realise(meta, x) for x in v]
    img = v.get("image")
    return dict(v, image=make_image(*meta["images"][img]) if img is not None else None)


def check(npz, tag, out):
    for k in ("input_ids", "attention_mask", "labels", "answers", "indices"):
        if f"{tag}.{k}" in npz.files:
            assert np.array_equal(out[k].numpy(), npz[f"{tag}.{k}"]), (tag, k)
        else:
            assert k not in out, (tag, k)
    if f"{tag}.pixel_values.shape" in npz.files:
        u8 = out["pixel_values_u8"].numpy()
        assert list(u8.shape) == list(npz[f"{tag}.pixel_values.shape"])
        d = pixel_digest(normalize_ref(u8.reshape(-1, *u8.shape[-3:])))
        assert np.array_equal(d["sub"], npz[f"{tag}.pixel_values.sub"]), tag  # bit-exact fp32
        for kk in ("sum", "sumsq", "wsum"):
            assert d[kk] == float(npz[f"{tag}.pixel_values.{kk}"]), (tag, kk)
    else:
        assert "pixel_values_u8" not in out


def test_chat_template_rendering(gold):
    """render_llava_chat == the reference's jinja template (config/model_config.yaml:63) rendered by transformers."""
    _, meta = gold
    assert len(meta["chat"]) >= 3
    for c in meta["chat"]:
        assert C.render_llava_chat(c["conversation"], c["add_generation_prompt"]) == c["rendered"]


@pytest.mark.parametrize("case", ["basic_left", "basic_right", "trunc", "text_answers", "nested"])
def test_collator_matches_reference(gold, case):
    npz, meta = gold
    c = meta["cases"][case]
    tok = adapter(c["padding_side"])
    out = C.mm_data_collator_preprocessor(realise(meta, c["samples"]), tok, c["max_length"],
                                          return_indices=c.get("return_indices", False),
                                          return_answers=c.get("return_answers", False))
    check(npz, case, out)


def test_label_mask_counts_the_unexpanded_prompt(gold):
    """Reference quirk (mm/dataset.py:41 vs :325-333): with an image, 575 prompt tokens stay supervised."""
    _, meta = gold
    c = meta["cases"]["basic_left"]
    out = C.mm_data_collator_preprocessor(realise(meta, c["samples"]), adapter("left"), 700)
    row = out["labels"][0]
    assert int((row == 4).sum()) > 500  # <image> ids left unmasked in the labels of the image row


@pytest.mark.parametrize("loss", ["grad_diff_forget_ce_retain_ce", "dpo", "idk", "sku", "LLMU"])
def test_pairing_matches_reference(gold, loss):
    _, meta = gold
    p = meta["pairs"]
    run = p["runs"][loss]
    ds = C.MixedForgetPairs(p["forget"], p["retain"], loss, idk_answers=p["idk_answers"], rng=random)
    random.seed(run["seed"])
    items = [ds[i] for i in range(len(ds))]
    assert items == run["items"]
    rng = random.Random(run["seed"])  # a private generator gives the same stream
    ds2 = C.MixedForgetPairs(p["forget"], p["retain"], loss, idk_answers=p["idk_answers"], rng=rng)
    assert [ds2[i] for i in range(len(ds2))] == run["items"]


def test_pairing_requires_idk_answers(gold):
    _, meta = gold
    with pytest.raises(ValueError):
        C.MixedForgetPairs(meta["pairs"]["forget"], meta["pairs"]["retain"], "dpo")


def test_forget_collator_matches_reference(gold):
    npz, meta = gold
    items = meta["pairs"]["runs"]["grad_diff_forget_ce_retain_ce"]["items"]
    batch = [{k: realise(meta, v) for k, v in it.items()} for it in items]
    out = C.mm_forget_data_collator_preprocessor(batch, adapter("left"), 700)
    assert list(out.keys()) == ["retain", "forget"]  # split order = dict order of the dataset item
    for split, o in out.items():
        check(npz, f"forget_collate.{split}", o)


def test_all_padding_row_is_fully_masked():
    tok = adapter("left")
    tok2 = C.TokenizerAdapter(encode=lambda s: [3, 3, 3], pad_token_id=3, image_token_id=4)  # degenerate: only pads
    out = C.mm_data_collator_preprocessor([{"image": None, "question": "x", "answer": "y"}], tok2, 16)
    assert (out["labels"] == -100).all()
    assert tok.pad_token_id == 3
