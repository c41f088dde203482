"""ORACLE fixture generator #4 (build container only; needs /root/reference + transformers + PIL):
the reference's collator and forget/retain pairing (/root/reference/mm/dataset.py:22-51, 182-354) run on a REAL
transformers `LlavaProcessor` assembled offline (tiny WordLevel tokenizer from oracle/tiny_tokenizer.py, PIL CLIP
image processor at 336, patch 14, the reference's chat template from config/model_config.yaml:63)
-> tests/golden/collator.npz + tests/golden/collator_pairs.json.

Usage: PYTHONPATH=/root/reference:/root/repo python oracle/make_golden_collator.py
`natsort` (imported by the reference's utils.py) is absent and stubbed for the import only.  The reference pins
transformers 4.45.1; the processor here is 5.15's (same expansion of <image> into 576 ids when patch_size is set).
"""
import importlib.machinery
import json
import os
import random
import sys
import types

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from oracle.tiny_tokenizer import build_tokenizers_object, pixel_digest, test_image  # noqa: E402


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m


def build_processor(padding_side):
    from PIL import Image  # noqa: F401
    from transformers import CLIPImageProcessor, LlavaProcessor, PreTrainedTokenizerFast

    t = PreTrainedTokenizerFast(tokenizer_object=build_tokenizers_object(), bos_token="<s>", eos_token="</s>",
                                unk_token="<unk>", pad_token="<pad>", padding_side=padding_side,
                                additional_special_tokens=["<image>"])
    ip = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}, do_convert_rgb=True)
    tmpl = yaml.safe_load(open("/root/reference/config/model_config.yaml"))["llava"]["chat_template"]
    return LlavaProcessor(image_processor=ip, tokenizer=t, patch_size=14, vision_feature_select_strategy="default",
                          chat_template=tmpl, num_additional_image_tokens=1)


IMAGES = {"land": (400, 500, 0), "port": (351, 301, 1), "sq": (336, 336, 2), "small": (120, 200, 3)}


def pil(name):
    from PIL import Image

    return Image.fromarray(test_image(*IMAGES[name]))


def samples_basic():
    return [
        {"image": "land", "question": "What is depicted in the image?",
         "answer": "A woman with short dark hair standing on a pier next to a small boat at sunset ."},
        {"image": None, "question": "Where was Jane Doe born ?", "answer": "Jane Doe was born in Lisbon in 1987 ."},
        {"image": "port", "question": "Describe the image in detail.", "answer": "A man in a grey coat reading a newspaper on a bench ."},
        {"image": None, "question": "What is her job ?", "answer": "She works as a marine biologist ."},
    ]


def realise(samples):
    return [dict(s, image=pil(s["image"]) if s["image"] is not None else None) for s in samples]


def put(res, tag, out):
    for k, v in out.items():
        if k == "pixel_values":
            for kk, vv in pixel_digest(v.numpy()).items():
                res[f"{tag}.pixel_values.{kk}"] = np.asarray(vv)
            res[f"{tag}.pixel_values.shape"] = np.asarray(v.shape)
        else:
            res[f"{tag}.{k}"] = v.numpy()


def main():
    os.makedirs(OUT, exist_ok=True)
    stub("natsort", natsorted=sorted)
    sys.path.insert(0, "/root/reference")
    from mm import dataset as ref  # /root/reference/mm/dataset.py

    res, meta = {}, {"images": IMAGES, "cases": {}}
    for side in ("left", "right"):
        proc = build_processor(side)
        s = samples_basic()
        meta["cases"][f"basic_{side}"] = {"samples": s, "max_length": 700, "padding_side": side}
        put(res, f"basic_{side}", ref.mm_data_collator_preprocessor(realise(s), proc, 700))
    proc = build_processor("left")
    # truncation: the answers of the image rows are cut; text rows are shorter than the limit
    s = samples_basic()
    meta["cases"]["trunc"] = {"samples": s, "max_length": 590, "padding_side": "left"}
    put(res, "trunc", ref.mm_data_collator_preprocessor(realise(s), proc, 590))
    # text only (no pixel_values key), return_answers + return_indices
    s = [dict(x, idx=10 + i) for i, x in enumerate(samples_basic()) if x["image"] is None]
    meta["cases"]["text_answers"] = {"samples": s, "max_length": 64, "padding_side": "left", "return_answers": True,
                                     "return_indices": True}
    put(res, "text_answers", ref.mm_data_collator_preprocessor(realise(s), proc, 64, return_indices=True, return_answers=True))
    # nested (list of lists): the sku "random" split layout, all-image rows so pixel_values nest too
    s = [[{"image": "sq", "question": "What is happening in this image?", "answer": "A man in a grey coat ."},
          {"image": "small", "question": "What is happening in this image?", "answer": "He is a retired pilot who lives in Oslo ."}],
         [{"image": "land", "question": "Describe the image in detail.", "answer": "A small boat at sunset ."},
          {"image": "port", "question": "Describe the image in detail.", "answer": "A woman with short dark hair ."}]]
    meta["cases"]["nested"] = {"samples": s, "max_length": 700, "padding_side": "left"}
    put(res, "nested", ref.mm_data_collator_preprocessor([realise(g) for g in s], proc, 700))

    # ---- pairing: MMMixedForgetDataset.__getitem__ without its hub-loading constructor
    forget = [{"image": "land", "caption": "A small boat at sunset .", "question": None, "answer": None},
              {"image": None, "caption": None, "question": "Where was Jane Doe born ?", "answer": "Jane Doe was born in Lisbon in 1987 ."},
              {"image": "port", "caption": "A man in a grey coat .", "question": None, "answer": None},
              {"image": None, "caption": None, "question": "What is her job ?", "answer": "She works as a marine biologist ."}]
    retain = [{"image": None, "caption": None, "question": "Who is the person in the photo ?", "answer": "He is a retired pilot who lives in Oslo ."},
              {"image": "sq", "caption": "A woman with short dark hair .", "question": None, "answer": None},
              {"image": None, "caption": None, "question": "Where was Jane Doe born ?", "answer": "She works as a marine biologist ."}]
    class Rows(list):  # datasets.Dataset.__getitem__ hands out a fresh dict per access (the reference mutates it)
        def __getitem__(self, i):
            return dict(list.__getitem__(self, i))

    import copy
    meta_forget, meta_retain = copy.deepcopy(forget), copy.deepcopy(retain)
    forget, retain = Rows(forget), Rows(retain)
    idk = ["I'm not sure.\n", "I don't have that information.\n", "That's beyond my current knowledge base.\n"]
    meta["pairs"] = {"forget": meta_forget, "retain": meta_retain, "idk_answers": idk, "runs": {}}
    for loss in ("grad_diff_forget_ce_retain_ce", "dpo", "idk", "sku", "LLMU"):
        ds = ref.MMMixedForgetDataset.__new__(ref.MMMixedForgetDataset)
        ds.forget_loss = loss
        fl = loss.upper()
        ds.return_pairs = (("forget", "retain", "idk") if fl in ("DPO", "LLMU") else ("idk", "retain") if fl == "IDK"
                           else ("forget", "retain", "random") if fl == "SKU" else ("forget", "retain"))  # mm/dataset.py:204-211
        ds.forget_data, ds.retain_data, ds.retain_len = forget, retain, len(retain)
        ds.idk_answers = idk
        ds.image_key, ds.caption_key, ds.question_key, ds.answer_key = "image", "caption", "question", "answer"
        random.seed(1234)
        items = [ds[i] for i in range(len(forget))]
        meta["pairs"]["runs"][loss] = {"seed": 1234, "items": items}
        if loss in ("grad_diff_forget_ce_retain_ce", "sku"):
            def real(v):
                return [real(x) for x in v] if isinstance(v, list) else dict(v, image=pil(v["image"]) if v["image"] is not None else None)
            batch = [{k: real(v) for k, v in it.items()} for it in items[:2]] if loss == "sku" else \
                    [{k: real(v) for k, v in it.items()} for it in items]
            if loss == "sku":  # nested split needs all-image or all-text rows to stack; keep the text-only forget rows' K lists
                batch = None
            if batch is not None:
                out = ref.mm_forget_data_collator_preprocessor(batch, proc, 700)
                for split, o in out.items():
                    put(res, f"forget_collate.{split}", o)
    # ---- chat template through the real jinja path
    convs = [
        ([{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "hello there"}]}], True),
        ([{"role": "system", "content": [{"type": "text", "text": "SYS"}]},
          {"role": "user", "content": [{"type": "text", "text": "a"}, {"type": "image"}, {"type": "text", "text": "b"}, {"type": "image"}]},
          {"role": "assistant", "content": [{"type": "text", "text": "c d"}]}], False),
        ([{"role": "user", "content": [{"type": "text", "text": "q"}]},
          {"role": "assistant", "content": [{"type": "text", "text": "ans"}]}], True),
    ]
    meta["chat"] = [{"conversation": c, "add_generation_prompt": g,
                     "rendered": proc.apply_chat_template(c, add_generation_prompt=g)} for c, g in convs]
    np.savez_compressed(os.path.join(OUT, "collator.npz"), **res)
    json.dump(meta, open(os.path.join(OUT, "collator_pairs.json"), "w"), indent=1)
    print({k: v.shape for k, v in res.items() if hasattr(v, "shape") and v.ndim > 0})
    print(os.path.getsize(os.path.join(OUT, "collator.npz")))


if __name__ == "__main__":
    main()
