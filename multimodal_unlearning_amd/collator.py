"""Host-side data formats either side of the forget step: the forget/retain pairing of
`MMMixedForgetDataset` (/root/reference/mm/dataset.py:182-280), the LLaVA chat rendering
(/root/reference/config/model_config.yaml:63, /root/reference/mm/dataset.py:22-51) and the collator
`mm_data_collator_preprocessor` / `mm_forget_data_collator_preprocessor` (/root/reference/mm/dataset.py:283-354).

MI355X-first split of the work: the host only tokenises, pads and resizes (PIL, uint8); it ships **uint8** pixels
(4x fewer PCIe bytes than the reference's fp32 `pixel_values`) and the rescale + CLIP normalisation runs on the
device in `mmu_clip_normalize_u8` when the batch is moved (`finalize_on_device`).  No model weights, tokenizer or
dataset is fetched here: the tokenizer is whatever object the caller holds (duck-typed, see `TokenizerAdapter`).

Reference behaviours kept on purpose (pinned by tests/golden/collator.npz, generated from the reference's own
functions with a real offline `LlavaProcessor`, oracle/make_golden_collator.py):
  * the number of masked question tokens is counted on the UNEXPANDED prompt (`<image>` = 1 token,
    mm/dataset.py:41) while `input_ids` carry the 576 expanded `<image>` ids, so the label mask ends 575 tokens
    before the answer starts when an image is present (mm/dataset.py:325-333);
  * the "idk" split of `MMMixedForgetDataset.__getitem__` formats the FORGET item again (the item with the
    "I don't know" answer is built and dropped, mm/dataset.py:258-263);
  * the "random" split overwrites `question` on K=3 sampled forget items but image items still draw a fresh
    caption question (mm/dataset.py:265-276).
"""
from __future__ import annotations

import random as _random
from dataclasses import dataclass
from typing import Callable, Sequence

import numpy as np
import torch

# mm/dataset.py:9-19 — prompt pool for image-caption items (data, not code)
IMAGE_CAPTION_QUESTIONS = [
    "What can you see in this picture?",
    "Tell me about the content of this image",
    "Can you give a description of the image?",
    "What is depicted in the image?",
    "Explain what you observe in the picture.",
    "Describe the image in detail.",
    "What is the main subject of this image?",
    "Can you describe the scene or objects in the image?",
    "What is happening in this image?",
]

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ------------------------------------------------------------------------------------------------ chat format
def render_llava_chat(messages, add_generation_prompt: bool) -> str:
    """The `llava` chat template of config/model_config.yaml:63 as straight Python: system text verbatim, other roles
    as `ROLE: `, all images of a message first (`<image>\\n` each), then every text followed by one space."""
    out = []
    for m in messages:
        if m["role"] == "system":
            out.append(m["content"][0]["text"])
        else:
            out.append(m["role"].upper() + ": ")
        for c in m["content"]:
            if c["type"] == "image":
                out.append("<image>\n")
        for c in m["content"]:
            if c["type"] == "text":
                out.append(c["text"] + " ")
    if add_generation_prompt:
        out.append("ASSISTANT:")
    return "".join(out)


@dataclass
class TokenizerAdapter:
    """What the collator needs from a tokenizer.  `encode(text)` returns ids INCLUDING special tokens (BOS), with
    `<image>` mapped to the single id `image_token_id`."""

    encode: Callable[[str], list]
    pad_token_id: int
    image_token_id: int
    padding_side: str = "left"

    @classmethod
    def from_hf(cls, tok, image_token="<image>"):
        """A transformers tokenizer object (PreTrainedTokenizer[Fast]); nothing is downloaded."""
        return cls(encode=lambda s: list(tok(s, add_special_tokens=True)["input_ids"]), pad_token_id=tok.pad_token_id,
                   image_token_id=tok.convert_tokens_to_ids(image_token), padding_side=tok.padding_side)


def convert_mm_data_to_model_format(tok: TokenizerAdapter, sample):
    """mm/dataset.py:22-51 -> (full_text, num_question_tokens)."""
    conv = [{"role": "user", "content": [{"type": "text", "text": sample["question"]}]}]
    if sample.get("image", None) is not None:
        conv[0]["content"].insert(0, {"type": "image"})
    question = render_llava_chat(conv, add_generation_prompt=True)
    n_q = len(tok.encode(question))  # unexpanded: <image> counts once
    conv.append({"role": "assistant", "content": [{"type": "text", "text": sample["answer"]}]})
    return render_llava_chat(conv, add_generation_prompt=False), n_q


# ------------------------------------------------------------------------------------------------ images (host half)
def clip_resize_crop_u8(image, size: int = 336) -> np.ndarray:
    """CLIP preprocessing up to (not including) rescale/normalise, as transformers' PIL CLIP image processor does it:
    RGB, bicubic resize of the SHORTEST edge to `size` (long edge int(size*long/short)), centre crop size x size.
    Returns uint8 [3, size, size].  Accepts a PIL image or a uint8 HWC array."""
    from PIL import Image

    if not isinstance(image, Image.Image):
        image = Image.fromarray(np.asarray(image, dtype=np.uint8))
    image = image.convert("RGB")
    w, h = image.size
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    image = image.resize((nw, nh), resample=Image.BICUBIC)
    a = np.asarray(image, dtype=np.uint8)  # [nh, nw, 3]
    top, left = (nh - size) // 2, (nw - size) // 2
    a = a[top:top + size, left:left + size]
    return np.ascontiguousarray(a.transpose(2, 0, 1))


def finalize_on_device(batch: dict, device) -> dict:
    """Move one split to the GPU; uint8 pixels become the fp32 `pixel_values` the reference's processor would have
    produced ((x/255 - mean)/std) in the HIP kernel `mmu_clip_normalize_u8`.  No CPU fallback."""
    from . import hip

    out = {}
    for k, v in batch.items():
        if k == "pixel_values_u8":
            u8 = v.to(device, non_blocking=True).contiguous()
            pv = torch.empty(u8.shape, device=device, dtype=torch.float32)
            # the nested "random" split of sku carries [B, K, 3, H, W]: the kernel sees a flat list of images
            hip.clip_normalize_u8(u8.view(-1, *u8.shape[-3:]), pv.view(-1, *u8.shape[-3:]), OPENAI_CLIP_MEAN, OPENAI_CLIP_STD)
            out["pixel_values"] = pv
        elif torch.is_tensor(v):
            out[k] = v.to(device, non_blocking=True)
            if k in ("attention_mask", "labels") and v.device.type == "cpu":
                from .llava import attach_host
                attach_host(out[k], v)  # lets the model see padding / supervised rows without a device sync (trainer.to_device)
        else:
            out[k] = v
    return out


# ------------------------------------------------------------------------------------------------ collator
def mm_data_collator_preprocessor(samples, tok: TokenizerAdapter, max_length: int, return_indices=False,
                                  return_answers=False, image_size: int = 336, patch_size: int = 14, device=None):
    """mm/dataset.py:283-354.  `samples`: list of {"question","answer","image"} or list of lists of those (nested
    return, used by the sku "random" split).  Returns input_ids / attention_mask / labels (int64) and, when any
    sample has an image, `pixel_values_u8` uint8 [n_img,3,H,W] — or fp32 `pixel_values` on `device` if given."""
    nested, list_size = False, 0
    if isinstance(samples[0], list):
        nested, list_size = True, len(samples[0])
        samples = [s for group in samples for s in group]

    n_img_tokens = (image_size // patch_size) ** 2  # "default" feature strategy: CLS dropped -> 576
    rows, n_questions = [], []
    for s in samples:
        text, n_q = convert_mm_data_to_model_format(tok, s)
        ids = []
        for t in tok.encode(text):
            ids.extend([t] * n_img_tokens if t == tok.image_token_id else [t])
        rows.append(ids[:max_length])  # truncation=True, right side
        n_questions.append(n_q)
    L = max(len(r) for r in rows)
    input_ids = torch.full((len(rows), L), tok.pad_token_id, dtype=torch.int64)
    attn = torch.zeros((len(rows), L), dtype=torch.int64)
    for i, r in enumerate(rows):
        sl = slice(L - len(r), L) if tok.padding_side == "left" else slice(0, len(r))
        input_ids[i, sl] = torch.tensor(r, dtype=torch.int64)
        attn[i, sl] = 1

    # Labels (mm/dataset.py:323-340), all rows at once: everything up to and including the question is ignored, counted
    # from the first token that differs from the pad id (so a pad id that also occurs as a real token shifts the span,
    # exactly as in the reference); right padding also ignores what follows the last non-pad token; an all-pad row is
    # ignored entirely.
    non_pad = input_ids != tok.pad_token_id
    first = non_pad.to(torch.int64).argmax(dim=1, keepdim=True)
    last = L - 1 - non_pad.flip(1).to(torch.int64).argmax(dim=1, keepdim=True)
    t = torch.arange(L, dtype=torch.int64)[None]
    q_end = first + torch.tensor(n_questions, dtype=torch.int64)[:, None]
    if tok.padding_side == "left":
        ignore = t < q_end
    else:
        ignore = ((t >= first) & (t < q_end)) | (t > last)
    ignore |= ~non_pad.any(dim=1, keepdim=True)
    labels = torch.where(ignore, torch.full_like(input_ids, -100), input_ids)

    out = {"input_ids": input_ids, "attention_mask": attn}
    images = [s["image"] for s in samples if s.get("image", None) is not None]
    if images:
        out["pixel_values_u8"] = torch.from_numpy(np.stack([clip_resize_crop_u8(im, image_size) for im in images]))
    out["labels"] = labels
    if return_answers:
        ans = [tok.encode(s["answer"]) for s in samples]
        La = max(len(a) for a in ans)
        at = torch.full((len(ans), La), tok.pad_token_id, dtype=torch.int64)
        for i, a in enumerate(ans):
            at[i, (La - len(a) if tok.padding_side == "left" else 0):][: len(a)] = torch.tensor(a, dtype=torch.int64)
        out["answers"] = at
    if return_indices:
        out["indices"] = torch.tensor([s["idx"] for s in samples])
    if nested:
        out = {k: torch.stack([v[i: i + list_size] for i in range(0, len(v), list_size)]) for k, v in out.items()}
    if device is not None:
        out = finalize_on_device(out, device)
    return out


def mm_forget_data_collator_preprocessor(samples, tok: TokenizerAdapter, max_length: int, device=None, image_size: int = 336,
                                         patch_size: int = 14):
    """mm/dataset.py:276-281: one collated dict per split present in the samples."""
    return {split: mm_data_collator_preprocessor([s[split] for s in samples], tok, max_length, device=device,
                                                 image_size=image_size, patch_size=patch_size)
            for split in samples[0].keys()}


# ------------------------------------------------------------------------------------------------ pairing
class MixedForgetPairs:
    """`MMMixedForgetDataset` (mm/dataset.py:182-280) over in-memory item lists (the reference loads them with
    `datasets.load_dataset` from the hub; there is no network here, so the caller supplies the rows).  The calls into
    `rng` happen in the reference's order, so with `rng = random` after `random.seed(s)` the draws are identical."""

    K_RANDOM = 3  # mm/dataset.py:266

    def __init__(self, forget_items: Sequence[dict], retain_items: Sequence[dict], forget_loss: str,
                 idk_answers: Sequence[str] = (), caption_key="caption", image_key="image", question_key="question",
                 answer_key="answer", rng=_random):
        self.forget_data, self.retain_data = forget_items, retain_items
        self.retain_len = len(retain_items)
        fl = forget_loss.upper()
        if fl in ("DPO", "LLMU"):
            self.return_pairs = ("forget", "retain", "idk")
        elif fl == "IDK":
            self.return_pairs = ("idk", "retain")
        elif fl == "SKU":
            self.return_pairs = ("forget", "retain", "random")
        else:
            self.return_pairs = ("forget", "retain")
        if "idk" in self.return_pairs and not idk_answers:
            raise ValueError("idk answers are required for this loss (the reference reads data/idontknow.jsonl)")
        self.idk_answers = list(idk_answers)
        self.image_key, self.caption_key, self.question_key, self.answer_key = image_key, caption_key, question_key, answer_key
        self.rng = rng

    def __len__(self):
        return len(self.forget_data)

    def _format_pair(self, pair):
        if pair.get(self.image_key, None) is None:
            return {"image": None, "question": pair[self.question_key], "answer": pair[self.answer_key]}
        return {"image": pair[self.image_key], "question": self.rng.choice(IMAGE_CAPTION_QUESTIONS),
                "answer": pair[self.caption_key]}

    def __getitem__(self, idx: int):
        retain_idx = self.rng.randint(0, self.retain_len - 1)
        forget_itm, retain_itm = self.forget_data[idx], self.retain_data[retain_idx]
        res = {}
        if "retain" in self.return_pairs:
            res["retain"] = self._format_pair(retain_itm)
        if "forget" in self.return_pairs:
            res["forget"] = self._format_pair(forget_itm)
        if "idk" in self.return_pairs:
            self.rng.choice(self.idk_answers)  # the two draws the reference makes for the item it then drops
            self.rng.choice(self.idk_answers)
            res["idk"] = self._format_pair(forget_itm)
        if "random" in self.return_pairs:
            question = res["forget"]["question"]
            idxs = self.rng.sample(range(len(self.forget_data)), self.K_RANDOM)
            items = [dict(self.forget_data[i]) for i in idxs]
            for it in items:
                it["question"] = question
            res["random"] = [self._format_pair(it) for it in items]
        return res
