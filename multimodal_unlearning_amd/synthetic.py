"""Synthetic (image + QA) batches with the tensor layout of the reference collator.

Counterpart of mm/dataset.py:283-354 (`mm_forget_data_collator_preprocessor` -> nested dict
{"forget": {...}, "retain": {...}[, "idk": {...}]}; each split = input_ids / attention_mask / pixel_values /
labels with LEFT padding and `labels[: first_non_pad + n_question_tokens] = -100`, :325-339).  No dataset,
tokenizer or image files are available offline, so token ids and pixels are drawn from seeded generators
(SURVEY.md §8d "Synthetic inputs").
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

from .config import LlavaConfig


def make_split(cfg: LlavaConfig, batch: int, seq_len: int, seed: int, n_question: int = 0,
               left_pad: Optional[List[int]] = None, device="cpu") -> Dict[str, torch.Tensor]:
    """One split: [pad]* BOS <image>*num_patches question... answer...  (ids drawn from U{3..V-2}).
    seq_len counts every position including the expanded image tokens."""
    rng = np.random.default_rng(seed)
    V, npatch = cfg.text.vocab_size, cfg.vision.num_patches
    ids = np.zeros((batch, seq_len), dtype=np.int64)
    mask = np.ones((batch, seq_len), dtype=np.int64)
    labels = np.full((batch, seq_len), -100, dtype=np.int64)
    hi = min(V - 1, cfg.image_token_index) if cfg.image_token_index >= 3 else V
    for b in range(batch):
        pad = 0 if left_pad is None else int(left_pad[b])
        body = seq_len - pad
        assert body >= 1 + npatch + n_question + 1, "sequence too short for image + question + answer"
        row = np.empty(body, dtype=np.int64)
        row[0] = 1  # BOS
        row[1:1 + npatch] = cfg.image_token_index
        row[1 + npatch:] = rng.integers(3, hi, size=body - 1 - npatch)
        ids[b, :pad] = 0
        mask[b, :pad] = 0
        ids[b, pad:] = row
        first_answer = pad + 1 + npatch + n_question
        labels[b, first_answer:] = ids[b, first_answer:]
    img = cfg.vision.image_size
    pix = rng.standard_normal((batch, cfg.vision.num_channels, img, img)).astype(np.float32)
    out = OrderedDict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                      pixel_values=torch.from_numpy(pix), labels=torch.from_numpy(labels))
    return {k: v.to(device) for k, v in out.items()}


def make_text_split(cfg: LlavaConfig, batch: int, seq_len: int, seed: int, n_question: int = 0,
                    right_pad: Optional[List[int]] = None, device="cpu") -> Dict[str, torch.Tensor]:
    """Text-only split with the layout of nlp/data_module.py:10-43 (`convert_raw_data_to_model_format`): fixed
    length, RIGHT padding with the EOS id, attention_mask 0 on the padding, labels = ids with the question span
    and every padded position but the first set to -100 (the first EOS stays supervised, :33-36)."""
    rng = np.random.default_rng(seed)
    V = cfg.text.vocab_size
    eos = 2
    hi = min(V - 1, cfg.image_token_index) if cfg.image_token_index >= 3 else V
    ids = np.full((batch, seq_len), eos, dtype=np.int64)
    mask = np.zeros((batch, seq_len), dtype=np.int64)
    labels = np.full((batch, seq_len), -100, dtype=np.int64)
    for b in range(batch):
        pad = 0 if right_pad is None else int(right_pad[b])
        body = seq_len - pad
        ids[b, 0] = 1
        ids[b, 1:body] = rng.integers(3, hi, size=body - 1)
        mask[b, :body] = 1
        labels[b, n_question:body] = ids[b, n_question:body]
        if pad > 0:
            labels[b, body] = eos
    out = OrderedDict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), labels=torch.from_numpy(labels))
    return {k: v.to(device) for k, v in out.items()}


def make_forget_batch(cfg: LlavaConfig, batch: int, seq_len: int, seed: int, n_question: int,
                      splits=("forget", "retain"), device="cpu") -> Dict[str, Dict[str, torch.Tensor]]:
    """Nested dict exactly as MMTrainerForgetting.compute_loss consumes it (mm/trainer.py:170-177)."""
    return {name: make_split(cfg, batch, seq_len, seed * 1000 + i, n_question, device=device)
            for i, name in enumerate(splits)}


def bench_batch(cfg: LlavaConfig, batch: int, seed: int = 1234, text_len: int = 512, supervised: int = 256,
                splits=("forget", "retain"), device="cpu"):
    """1 BOS + num_patches <image> + (text_len-1) text ids, the last `supervised` positions labelled."""
    S = cfg.vision.num_patches + text_len
    n_question = S - 1 - cfg.vision.num_patches - supervised
    return make_forget_batch(cfg, batch, S, seed, n_question, splits, device)


def teacher_params_from(params, seed: int = 7, rel: float = 0.25):
    """Frozen 'teacher' = student + a deterministic perturbation (else every KL term is identically 0)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for k, v in params.items():
        noise = torch.from_numpy(rng.standard_normal(tuple(v.shape)).astype(np.float32)) * (rel * 0.02)
        t = (v.float() + noise).to(torch.bfloat16).to(v.dtype)
        if k == "vision.patch_embed.weight":
            t[v == 0] = 0  # keep the zero-padded im2col tail at zero
        out[k] = t
    return out
