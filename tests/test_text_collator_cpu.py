"""The text trainer's data path (BASELINE config 1) against the reference's own functions: tests/golden/text_collator.npz
holds the outputs of /root/reference/nlp/data_module.py `convert_raw_data_to_model_format`,
`TextForgetDatasetQA.__getitem__` and /root/reference/nlp/dataloader.py `custom_data_collator_forget` on a tiny offline
tokenizer (oracle/make_golden_text_collator.py)."""
import os

import numpy as np
import pytest
import torch

from multimodal_unlearning_amd.collator import TokenizerAdapter
from multimodal_unlearning_amd.text_collator import (TextForgetPairs, TextForgetTriples, convert_raw_data_to_model_format,
                                                      custom_data_collator_forget)
from oracle.make_golden_text_collator import EXTRA, IDK, ROWS_F, ROWS_R
from oracle.tiny_tokenizer import build_tokenizers_object

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "text_collator.npz"))
EOS = 2


@pytest.fixture(scope="module")
def tok():
    t = build_tokenizers_object(EXTRA)
    return TokenizerAdapter(encode=lambda s: t.encode(s).ids, pad_token_id=3, image_token_id=4, padding_side="right")


@pytest.mark.parametrize("L", [24, 12])
def test_convert_raw_data_matches_reference(tok, L):
    for i, r in enumerate(ROWS_F + ROWS_R):
        ids, lab, msk = convert_raw_data_to_model_format(tok, EOS, L, r["question"], r["answer"])
        got = np.stack([ids.numpy(), lab.numpy(), msk.numpy()])
        assert np.array_equal(got, G[f"conv.{L}.{i}"]), (L, i)


@pytest.mark.parametrize("loss_type", ["grad_ascent", "idk"])
def test_pairing_and_collator_match_reference(tok, loss_type):
    ds = TextForgetPairs(ROWS_F, ROWS_R, tok, EOS, max_length=24, loss_type=loss_type, idk_answers=IDK)
    torch.manual_seed(1234)  # same global-generator draws as the reference's torch.randint calls
    items = [ds[i] for i in range(len(ds))]
    for i, it in enumerate(items):
        got = np.stack([np.stack([t.numpy() for t in split]) for split in it])
        assert np.array_equal(got, G[f"item.{loss_type}.{i}"]), (loss_type, i)
    col = custom_data_collator_forget(items)
    got = np.stack([np.stack([t.numpy() for t in split]) for split in col])
    assert np.array_equal(got, G[f"collated.{loss_type}"])
    assert len(col) == 2 and col[0][0].shape == (len(ROWS_F), 24)


def test_dpo_triples_match_reference(tok):
    """TextForgetDatasetDPOQA (dpo / LLMU): items [idk, forget, retain], collated position-wise into three splits."""
    ds = TextForgetTriples(ROWS_F, ROWS_R, tok, EOS, max_length=24, idk_answers=IDK)
    torch.manual_seed(4321)
    items = [ds[i] for i in range(len(ds))]
    for i, it in enumerate(items):
        got = np.stack([np.stack([t.numpy() for t in split]) for split in it])
        assert np.array_equal(got, G[f"item.dpo.{i}"]), i
    col = custom_data_collator_forget(items)
    got = np.stack([np.stack([t.numpy() for t in split]) for split in col])
    assert np.array_equal(got, G["collated.dpo"])
    assert len(col) == 3
    with pytest.raises(ValueError):
        TextForgetTriples(ROWS_F, ROWS_R, tok, EOS)


def test_collator_keeps_an_optional_third_split(tok):
    a = convert_raw_data_to_model_format(tok, EOS, 16, "What is her job ?", "She works as a marine biologist .")
    col = custom_data_collator_forget([[a, a, a], [a, a, a]])
    assert len(col) == 3 and all(t.shape == (2, 16) for split in col for t in split)
    with pytest.raises(ValueError):
        TextForgetPairs(ROWS_F, ROWS_R, tok, EOS, loss_type="idk")
