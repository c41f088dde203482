"""Host-side data formats of the TEXT trainer (BASELINE config 1: nlp/forget.py on TOFU): what
`CustomTrainerForgetting.compute_loss` receives.

Counterparts of /root/reference/nlp/data_module.py:10-43 (`convert_raw_data_to_model_format`: question / answer tags,
fixed length, RIGHT padding with the EOS id, labels = ids with the question span and all padding but the first EOS
ignored), :46-94 (`TextForgetDatasetQA.__getitem__`: each forget item is paired with the retain item
`(idx + randint(len(retain))) % len(retain)`; the "idk" split answers a forget question with a random line of
idontknow.jsonl), :97-130 (`TextForgetDatasetDPOQA`, the dpo / LLMU dataset) and /root/reference/nlp/dataloader.py:497-517 (`custom_data_collator_forget`: per split, stack the
(input_ids, labels, attention_mask) tuples).  No dataset or tokenizer is fetched: rows and the tokenizer come from the
caller (`TokenizerAdapter`, collator.py).  Pinned by tests/golden/text_collator.npz, generated from the reference's own
functions (oracle/make_golden_text_collator.py).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from .collator import TokenizerAdapter

# config/model_config.yaml:1-8 (llama2-7b): the tags the text trainer wraps questions and answers in
LLAMA2_TAGS = {"question_start_tag": "[INST] ", "question_end_tag": " [/INST]", "answer_tag": ""}


def convert_raw_data_to_model_format(tok: TokenizerAdapter, eos_token_id: int, max_length: int, question: str, answer: str,
                                     tags: Dict[str, str] = LLAMA2_TAGS):
    """-> (input_ids, labels, attention_mask), each int64 [max_length] (nlp/data_module.py:10-43)."""
    new_question = tags["question_start_tag"] + question + tags["question_end_tag"]
    n_q = len(tok.encode(new_question))                      # special tokens (BOS) included, as the reference counts
    ids = tok.encode(new_question + tags["answer_tag"] + answer)[:max_length]   # truncation=True, right side
    n = len(ids)
    input_ids = torch.full((max_length,), eos_token_id, dtype=torch.int64)
    input_ids[:n] = torch.tensor(ids, dtype=torch.int64)
    attention_mask = torch.zeros(max_length, dtype=torch.int64)
    attention_mask[:n] = 1
    labels = torch.full((max_length,), -100, dtype=torch.int64)
    labels[:n] = input_ids[:n]
    if n < max_length:
        labels[n] = eos_token_id                             # the first padded EOS stays supervised
    labels[:min(n_q, max_length)] = -100                     # question tokens carry no loss
    return input_ids, labels, attention_mask


class TextForgetPairs:
    """`TextForgetDatasetQA` (nlp/data_module.py:46-94) over in-memory rows ({"question", "answer"} dicts).  Random
    draws use torch's global generator in the reference's order, so after `torch.manual_seed(s)` the items match."""

    def __init__(self, forget_rows: Sequence[dict], retain_rows: Sequence[dict], tok: TokenizerAdapter, eos_token_id: int,
                 max_length: int = 500, loss_type: str = "grad_ascent", idk_answers: Sequence[str] = (),
                 tags: Dict[str, str] = LLAMA2_TAGS):
        self.forget_data, self.retain_data = forget_rows, retain_rows
        self.tok, self.eos, self.max_length, self.tags = tok, eos_token_id, max_length, tags
        self.splits = ("idk", "retain") if loss_type == "idk" else ("forget", "retain")
        if loss_type == "idk" and not idk_answers:
            raise ValueError("loss_type 'idk' needs the idk answers (the reference reads data/idontknow.jsonl)")
        self.idk = list(idk_answers)

    def __len__(self):
        return len(self.forget_data)

    def __getitem__(self, idx: int):
        rets = []
        for data_type in self.splits:
            data = self.retain_data if data_type == "retain" else self.forget_data
            if data_type == "retain":  # NB the shifted index is kept for later splits, as in the reference's loop
                idx = (idx + torch.randint(0, len(self.retain_data), (1,)).item()) % len(self.retain_data)
            question, answer = data[idx]["question"], data[idx]["answer"]
            if data_type == "idk":
                answer = self.idk[torch.randint(0, len(self.idk), (1,)).item()].strip()
            rets.append(convert_raw_data_to_model_format(self.tok, self.eos, self.max_length, question, answer, self.tags))
        return rets


class TextForgetTriples:
    """`TextForgetDatasetDPOQA` (nlp/data_module.py:97-130), what nlp/forget.py:63-70 builds for forget_loss "dpo" and
    "LLMU": every item is [idk, forget, retain] IN THAT ORDER — the forget question answered by a random line of
    idontknow.jsonl, the forget pair, a retain pair at `(idx + randint) % len(retain)`.  The collator below names the
    POSITIONS (forget, retain, idk), so downstream the first tuple the trainer calls "forget" holds the idk answers; that
    is the reference's behaviour and is reproduced, not corrected (tests/golden/text_collator.npz: item.dpo.*)."""

    def __init__(self, forget_rows: Sequence[dict], retain_rows: Sequence[dict], tok: TokenizerAdapter, eos_token_id: int,
                 max_length: int = 500, idk_answers: Sequence[str] = (), tags: Dict[str, str] = LLAMA2_TAGS):
        if not idk_answers:
            raise ValueError("dpo / LLMU need the idk answers (the reference reads data/idontknow.jsonl)")
        self.forget_data, self.retain_data, self.idk = forget_rows, retain_rows, list(idk_answers)
        self.tok, self.eos, self.max_length, self.tags = tok, eos_token_id, max_length, tags

    def __len__(self):
        return len(self.forget_data)

    def __getitem__(self, idx: int):
        rets = []
        for data_type in ("idk", "forget", "retain"):
            data = self.retain_data if data_type == "retain" else self.forget_data
            if data_type == "retain":
                idx = (idx + torch.randint(0, len(self.retain_data), (1,)).item()) % len(self.retain_data)
            question = data[idx]["question"]
            if data_type == "idk":
                answer = self.idk[torch.randint(0, len(self.idk), (1,)).item()].strip()
            else:
                answer = data[idx]["answer"]
            rets.append(convert_raw_data_to_model_format(self.tok, self.eos, self.max_length, question, answer, self.tags))
        return rets


def custom_data_collator_forget(samples: List[list]):
    """nlp/dataloader.py:497-517: batch_size x [split] x (ids, labels, mask) -> [split] x [ids, labels, mask] stacked;
    a third split is present only when the samples carry one."""
    n_splits = 3 if any(len(s) > 2 and s[2] is not None for s in samples) else 2
    return [[torch.stack([s[j][k] for s in samples]) for k in range(3)] for j in range(n_splits)]
