"""ORACLE fixture generator #6 (build container only; needs /root/reference + transformers):
the text trainer's data path — /root/reference/nlp/data_module.py:10-43 `convert_raw_data_to_model_format`, :46-94
`TextForgetDatasetQA.__getitem__`, :97-130 `TextForgetDatasetDPOQA.__getitem__` (called unbound on in-memory rows: the
classes themselves load TOFU from the hub) and
/root/reference/nlp/dataloader.py:497-517 `custom_data_collator_forget` — on a tiny offline tokenizer
-> tests/golden/text_collator.npz.
Usage: PYTHONPATH=/root/reference:/root/reference/nlp:/root/repo python oracle/make_golden_text_collator.py
Absent third-party modules (natsort, deepspeed, wandb, ...) are stubbed for the import only."""
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.tiny_tokenizer import build_tokenizers_object  # noqa: E402

ROWS_F = [{"question": "Where was Jane Doe born ?", "answer": "Jane Doe was born in Lisbon in 1987 ."},
          {"question": "What is her job ?", "answer": "She works as a marine biologist ."},
          {"question": "Who is the person in the photo ?", "answer": "He is a retired pilot who lives in Oslo ."}]
ROWS_R = [{"question": "What is her job ?", "answer": "A man in a grey coat reading a newspaper on a bench ."},
          {"question": "Where was Jane Doe born ?", "answer": "A woman with short dark hair standing on a pier next to a small boat at sunset ."},
          {"question": "Who is the person in the photo ?", "answer": "She works as a marine biologist ."},
          {"question": "What is her job ?", "answer": "He is a retired pilot who lives in Oslo ."}]
IDK = ["I'm not sure.", "I don't have that information.", "That's beyond my current knowledge base."]
EXTRA = ["[INST] [/INST]"]


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m


def main():
    from transformers import PreTrainedTokenizerFast, Trainer  # noqa: F401  resolve lazy imports first

    stub("natsort", natsorted=sorted)
    for n in ("deepspeed", "wandb"):
        stub(n)
    os.chdir("/root/reference")
    import data_module as dm  # /root/reference/nlp/data_module.py

    tok = PreTrainedTokenizerFast(tokenizer_object=build_tokenizers_object(EXTRA), bos_token="<s>", eos_token="</s>",
                                  unk_token="<unk>", pad_token="<pad>")
    cfgs = {"question_start_tag": "[INST] ", "question_end_tag": " [/INST]", "answer_tag": ""}
    res = {}
    for L in (24, 12):  # 12 truncates the longer rows
        for i, r in enumerate(ROWS_F + ROWS_R):
            ids, lab, msk = dm.convert_raw_data_to_model_format(tok, L, r["question"], r["answer"], cfgs)
            res[f"conv.{L}.{i}"] = np.stack([ids.numpy(), lab.numpy(), msk.numpy()])
    for loss_type in ("grad_ascent", "idk"):
        fake = types.SimpleNamespace(tokenizer=tok, max_length=24, loss_type=loss_type, forget_data=ROWS_F, retain_data=ROWS_R,
                                     model_configs=cfgs, idk=[x + "\n" for x in IDK],
                                     split1="idk" if loss_type == "idk" else "forget", split2="retain")
        torch.manual_seed(1234)
        items = [dm.TextForgetDatasetQA.__getitem__(fake, i) for i in range(len(ROWS_F))]
        for i, it in enumerate(items):
            res[f"item.{loss_type}.{i}"] = np.stack([np.stack([t.numpy() for t in split]) for split in it])
        import importlib
        try:
            dl = importlib.import_module("dataloader")  # /root/reference/nlp/dataloader.py
            col = dl.custom_data_collator_forget(items)
            res[f"collated.{loss_type}"] = np.stack([np.stack([t.numpy() for t in split]) for split in col])
        except Exception as e:  # the module imports the whole trainer stack; record why if that fails
            print("custom_data_collator_forget not importable:", repr(e))
    # nlp/forget.py:63-70: dpo / LLMU runs use TextForgetDatasetDPOQA, whose items are [idk, forget, retain] IN THAT ORDER
    # while custom_data_collator_forget names the positions (forget, retain, idk)
    fake = types.SimpleNamespace(tokenizer=tok, max_length=24, forget_data=ROWS_F, retain_data=ROWS_R, model_configs=cfgs,
                                 idk=[x + "\n" for x in IDK])
    torch.manual_seed(4321)
    items = [dm.TextForgetDatasetDPOQA.__getitem__(fake, i) for i in range(len(ROWS_F))]
    for i, it in enumerate(items):
        res[f"item.dpo.{i}"] = np.stack([np.stack([t.numpy() for t in split]) for split in it])
    import importlib
    col = importlib.import_module("dataloader").custom_data_collator_forget(items)
    res["collated.dpo"] = np.stack([np.stack([t.numpy() for t in split]) for split in col])
    path = os.path.join(ROOT, "tests", "golden", "text_collator.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, os.path.getsize(path), sorted(res)[:4], "...")


if __name__ == "__main__":
    main()
