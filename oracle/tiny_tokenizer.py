"""ORACLE helper (test infrastructure): a tiny offline WordLevel tokenizer + deterministic test images shared by
oracle/make_golden_collator.py (reference side: wrapped in a real transformers LlavaProcessor) and
tests/test_collator_cpu.py (our side: raw `tokenizers` object behind TokenizerAdapter).  No files, no network."""
from __future__ import annotations

import numpy as np

SPECIALS = ["<unk>", "<s>", "</s>", "<pad>", "<image>"]

CORPUS = [
    "USER: ASSISTANT:",
    "Who is the person in the photo ? Where was Jane Doe born ? What is her job ?",
    "Jane Doe was born in Lisbon in 1987 . She works as a marine biologist .",
    "A woman with short dark hair standing on a pier next to a small boat at sunset .",
    "A man in a grey coat reading a newspaper on a bench .",
    "He is a retired pilot who lives in Oslo .",
    "I'm not sure. I don't have that information. That's beyond my current knowledge base.",
]


def vocab(extra_texts=()):
    from multimodal_unlearning_amd.collator import IMAGE_CAPTION_QUESTIONS

    v = {t: i for i, t in enumerate(SPECIALS)}
    for text in list(CORPUS) + list(IMAGE_CAPTION_QUESTIONS) + list(extra_texts):
        for w in text.split():
            v.setdefault(w, len(v))
    return v


def build_tokenizers_object(extra_texts=()):
    from tokenizers import AddedToken, Tokenizer, models, pre_tokenizers, processors

    tok = Tokenizer(models.WordLevel(vocab(extra_texts), unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])
    tok.add_special_tokens([AddedToken(t, special=True, normalized=False) for t in SPECIALS])
    return tok


def adapter(padding_side="left", extra_texts=()):
    from multimodal_unlearning_amd.collator import TokenizerAdapter

    tok = build_tokenizers_object(extra_texts)
    return TokenizerAdapter(encode=lambda s: tok.encode(s).ids, pad_token_id=3, image_token_id=4,
                            padding_side=padding_side)


def test_image(h, w, k=0):
    """Closed-form uint8 HWC image (no RNG): smooth ramps + a coarse checker so bicubic resampling matters."""
    y, x = np.mgrid[0:h, 0:w]
    ch = []
    for c in range(3):
        v = (x * (3 + c) + y * (5 - c) + 70 * c + 37 * k) // 2 + 40 * (((x // 23) + (y // 17) + c) % 2) + (x * y) // (97 + 10 * c)
        ch.append(v % 256)
    return np.stack(ch, axis=-1).astype(np.uint8)


def pixel_digest(pv):
    """Size-independent fingerprint of an fp32 pixel tensor [n,3,H,W]: strided subsample + float64 moments."""
    pv = np.asarray(pv, dtype=np.float32)
    pv = pv.reshape(-1, *pv.shape[-3:])
    n = pv.size
    wts = (np.arange(n, dtype=np.float64) % 1009 + 1.0).reshape(pv.shape)
    d = pv.astype(np.float64)
    return {"sub": pv[:, :, ::7, ::5].copy(), "sum": d.sum(), "sumsq": (d * d).sum(), "wsum": (d * wts).sum()}
