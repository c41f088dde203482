"""Forget entry point: `python -m multimodal_unlearning_amd.forget [--gpus N] key=value ...`

Counterpart of the reference's launcher + entry script for the forget step — `accelerate launch mm/forget.py k=v ...`
(README.md:64, run_exps.py:36) and `mm/forget.py:20-204` — on this repo's own pieces:

    ForgetConfig (config/mm/forget.yaml flag surface, Hydra-style `k=v` overrides)
      -> rank discovery from WORLD_SIZE / LOCAL_RANK (mm/forget.py:23-31); `--gpus N` starts the N ranks itself
      -> save_dir / overwrite_dir handling, resolved config written to save_dir/config.yaml (:40-51)
      -> MixedForgetPairs + collator (mm/forget.py:57-63,170; mm/dataset.py) or the synthetic generator
      -> max_steps / warm-up formulae (:65-68,147-148)
      -> student (+ frozen teacher when the loss or the L1/L0 regulariser needs it, :91-99), vision tower frozen
         (:117-118), optional LoRA on every Linear (:126-137)
      -> MMTrainerForgetting plug-in + ForgetEngine (replaces TrainingArguments + transformers.Trainer.train, :139-190)
      -> save_pretrained in HF layout on rank 0 (:193-195)

There is no network in this environment and the reference ships no weights or data: `model_path` must be a local
directory of HF safetensors shards (else the named architecture is random-initialised and the run says so), and the
data are either local JSON-lines files (`{"question", "answer", "image": path | null}` per line, or
`{"caption", "image"}`), tokenised by a local `tokenizer_path`, or `forget_data_path=synthetic:<n_items>`.
"""
from __future__ import annotations

import json
import os
import sys
import time
from dataclasses import asdict, dataclass, fields
from typing import List

from .config import ForgetConfig


@dataclass
class ForgetRunConfig(ForgetConfig):
    """ForgetConfig plus what the entry script itself reads (config/mm/forget.yaml:1-57)."""
    model_family: str = "llava-1.5-7b"          # "llava-1.5-7b" | "llava-1.5-13b" | "tiny" (tests)
    forget_data_path: str = "synthetic:64"
    retain_data_path: str = "synthetic:64"
    forget_split: str = "forget10"
    retain_split: str = "retain90"
    tokenizer_path: str = ""
    save_dir: str = ""
    overwrite_dir: bool = True
    save_model: bool = True
    optimizer: str = "adamw"                    # "adamw" (fp32 moments) | "adamw8bit" (the reference's adamw_bnb_8bit)
    dp_mode: str = "allreduce"                  # "allreduce" | "rs_ag"
    seq_len: int = 0                            # synthetic data: 0 = 576 image + 512 text tokens (text: 500)
    max_steps: int = 0                          # 0 = the reference's formula
    gradient_checkpointing: bool = False        # config/model_config.yaml `gradient_checkpointing` (mm/forget.py:123-124): recompute
                                                # the decoder layers' activations in backward instead of keeping them
    modality: str = "mm"                        # "mm": mm/forget.py (image + QA pairs, MMTrainerForgetting);
                                                # "text": nlp/forget.py (TOFU-style QA, CustomTrainerForgetting, tuple inputs)


def parse_overrides(argv: List[str], cfg: ForgetRunConfig) -> ForgetRunConfig:
    """Hydra-style `key=value` overrides (run_exps.py:17-18); `LoRA.r=8` style keys map to lora_r / lora_alpha /
    lora_dropout.  Booleans accept the reference's string forms ("true" / "false")."""
    types = {f.name: f.type for f in fields(cfg)}
    for a in argv:
        if "=" not in a:
            raise SystemExit(f"expected key=value, got {a!r}")
        k, v = a.split("=", 1)
        k = {"LoRA.r": "lora_r", "LoRA.alpha": "lora_alpha", "LoRA.dropout": "lora_dropout",
             "freeze_vision_module": "freeze_vision_module"}.get(k, k)
        if k not in types:
            raise SystemExit(f"unknown config key {k!r}")
        cur = getattr(cfg, k)
        if isinstance(cur, bool):
            val = v.strip().lower() in ("1", "true", "yes")
        elif isinstance(cur, int):
            val = int(float(v))
        elif isinstance(cur, float):
            val = float(v)
        else:
            val = v
        setattr(cfg, k, val)
    if not cfg.save_dir:  # config/mm/forget.yaml:50
        cfg.save_dir = f"{cfg.model_path or cfg.model_family}/{cfg.forget_loss}_{cfg.forget_split}_L1_{cfg.l1_lambda}_from{cfg.l_norm_from}"
    return cfg


def _architecture(name: str):
    from .config import llava_1_5_7b, llava_1_5_13b, tiny_test_config

    table = {"llava-1.5-7b": llava_1_5_7b, "llava": llava_1_5_7b, "llava-1.5-13b": llava_1_5_13b, "tiny": tiny_test_config}
    if name not in table:
        raise SystemExit(f"model_family {name!r}: this path covers LLaVA-1.5 ({', '.join(table)})")
    return table[name]()


def _has_checkpoint(path: str) -> bool:
    return bool(path) and os.path.isdir(path) and any(f.endswith(".safetensors") and not f.startswith("adapter_")
                                                      for f in os.listdir(path))


class _SyntheticPairs:
    """`synthetic:<n>`: n items whose collated micro-batches come from the seeded generator (synthetic.py)."""

    def __init__(self, n, cfg, splits, seq_len, seed):
        self.n, self.cfg, self.splits, self.seq_len, self.seed = n, cfg, splits, seq_len, seed

    def __len__(self):
        return self.n

    def batch(self, indices, device):
        from .synthetic import make_forget_batch

        npatch = self.cfg.vision.num_patches
        S = self.seq_len or npatch + 512
        nq = max(0, min(S - 2 - npatch, (S - 1 - npatch) // 2))
        return make_forget_batch(self.cfg, len(indices), S, seed=self.seed * 100003 + int(indices[0]), n_question=nq,
                                 splits=self.splits, device=device)


class _SyntheticText:
    """`synthetic:<n>` for modality=text: [B, 500] right-padded tuples (nlp/data_module.py:10-43 layout)."""

    def __init__(self, n, cfg, n_splits, seq_len, seed):
        self.n, self.cfg, self.n_splits, self.seq_len, self.seed = n, cfg, n_splits, seq_len or 500, seed

    def __len__(self):
        return self.n

    def batch(self, indices, device):
        from .synthetic import make_text_split

        S, B = self.seq_len, len(indices)
        out = []
        for j in range(self.n_splits):
            sp = make_text_split(self.cfg, B, S, seed=self.seed * 100003 + int(indices[0]) * 7 + j, n_question=S // 4,
                                 right_pad=[(7 * (int(i) + j)) % max(1, S // 5) for i in indices], device=device)
            out.append((sp["input_ids"], sp["labels"], sp["attention_mask"]))
        return out


class _JsonlText:
    """Local JSON-lines {"question", "answer"} rows through TextForgetPairs + custom_data_collator_forget."""

    def __init__(self, rc, tokenizer_path):
        from transformers import AutoTokenizer

        from .collator import TokenizerAdapter
        from .text_collator import TextForgetPairs, TextForgetTriples

        read = lambda p: [json.loads(line) for line in open(p) if line.strip()]
        tok = AutoTokenizer.from_pretrained(tokenizer_path, local_files_only=True)
        idk = []
        idk_path = os.path.join(os.path.dirname(rc.forget_data_path), "idontknow.jsonl")
        if os.path.exists(idk_path):
            idk = [line.strip() for line in open(idk_path) if line.strip()]
        if rc.forget_loss in ("dpo", "LLMU"):  # nlp/forget.py:63-70: TextForgetDatasetDPOQA ([idk, forget, retain] items)
            if not idk:
                raise SystemExit(f"forget_loss={rc.forget_loss} needs idontknow.jsonl next to {rc.forget_data_path}")
            self.pairs = TextForgetTriples(read(rc.forget_data_path), read(rc.retain_data_path), TokenizerAdapter.from_hf(tok),
                                           tok.eos_token_id, max_length=rc.seq_len or 500, idk_answers=idk)
        else:
            self.pairs = TextForgetPairs(read(rc.forget_data_path), read(rc.retain_data_path), TokenizerAdapter.from_hf(tok),
                                         tok.eos_token_id, max_length=rc.seq_len or 500, loss_type=rc.forget_loss,
                                         idk_answers=idk)

    def __len__(self):
        return len(self.pairs)

    def batch(self, indices, device):
        from .text_collator import custom_data_collator_forget

        col = custom_data_collator_forget([self.pairs[int(i)] for i in indices])
        return [tuple(t.to(device) for t in split) for split in col]


class _JsonlPairs:
    """Local JSON-lines forget / retain files through MixedForgetPairs + the reference-exact collator."""

    def __init__(self, rc: ForgetRunConfig, cfg, tokenizer_path):
        from transformers import AutoTokenizer

        from .collator import MixedForgetPairs, TokenizerAdapter

        read = lambda p: [json.loads(line) for line in open(p) if line.strip()]
        tok = AutoTokenizer.from_pretrained(tokenizer_path, local_files_only=True)
        tok.padding_side = "left"  # mm/forget.py:54
        self.tok = TokenizerAdapter.from_hf(tok)
        idk = []
        idk_path = os.path.join(os.path.dirname(rc.forget_data_path), "idontknow.jsonl")
        if os.path.exists(idk_path):
            idk = [line.strip() for line in open(idk_path) if line.strip()]
        self.pairs = MixedForgetPairs(read(rc.forget_data_path), read(rc.retain_data_path), rc.forget_loss, idk_answers=idk)
        self.rc, self.cfg = rc, cfg

    def __len__(self):
        return len(self.pairs)

    def batch(self, indices, device):
        from PIL import Image

        from .collator import mm_forget_data_collator_preprocessor

        samples = []
        for i in indices:
            s = self.pairs[int(i)]
            for split in s.values():
                for it in (split if isinstance(split, list) else [split]):
                    if isinstance(it.get("image"), str):
                        it["image"] = Image.open(it["image"])
            samples.append(s)
        v = self.cfg.vision  # CLIP-L/14-336 for LLaVA-1.5; the tiny test architecture has its own image size
        return mm_forget_data_collator_preprocessor(samples, self.tok, self.rc.max_length, device=device,
                                                    image_size=v.image_size, patch_size=v.patch_size)


def run(rc: ForgetRunConfig):
    import random

    import numpy as np
    import torch

    from . import trainer_utils as tu
    from .checkpoint import load_pretrained, save_pretrained
    from .engine import ForgetEngine, init_distributed, shard_indices, steps_from_reference_formula
    from .llava import LlavaHIP, LoraSpec
    from .trainer import CustomTrainerForgetting, MMTrainerForgetting

    # mm/forget.py:40-44 "Directory already exists": decided BEFORE the rendezvous from the launcher's env, so that every
    # rank leaves (a rank-0-only exit after init_process_group would strand the others in their first collective)
    if os.path.exists(rc.save_dir) and not rc.overwrite_dir:
        raise SystemExit("Directory already exists")
    rank, local, world = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("the forget step runs on MI355X GPUs (HIP kernels; there is no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    random.seed(rc.seed)
    np.random.seed(rc.seed)
    torch.manual_seed(rc.seed)  # transformers.set_seed (mm/forget.py:33)
    log = (lambda *a: print(*a, flush=True)) if rank == 0 else (lambda *a: None)
    log(f"num_devices: {world}")
    log("######################\nSaving to: ", rc.save_dir, "\n######################")
    if rank == 0:
        os.makedirs(rc.save_dir, exist_ok=True)
        import yaml

        yaml.safe_dump(asdict(rc), open(os.path.join(rc.save_dir, "config.yaml"), "w"))

    cfg = _architecture(rc.model_family)
    lt = rc.forget_loss
    splits = {"DPO": ("forget", "retain", "idk"), "LLMU": ("forget", "retain", "idk"), "IDK": ("idk", "retain"),
              "SKU": ("forget", "retain", "random")}.get(lt.upper(), ("forget", "retain"))
    text = rc.modality == "text"
    if rc.modality not in ("mm", "text"):
        raise SystemExit(f"modality {rc.modality!r}: expected 'mm' or 'text'")
    if text:
        n_splits = 3 if lt in ("dpo", "LLMU") else 2  # nlp/forget.py: TextForgetDatasetDPOQA adds the idk split
        if rc.forget_data_path.startswith("synthetic:"):
            data = _SyntheticText(int(rc.forget_data_path.split(":")[1]), cfg, n_splits, rc.seq_len, rc.seed)
        else:
            if not rc.tokenizer_path:
                raise SystemExit("JSON-lines data need tokenizer_path=<local tokenizer directory>")
            data = _JsonlText(rc, rc.tokenizer_path)
    elif rc.forget_data_path.startswith("synthetic:"):
        if "random" in splits:
            raise SystemExit("the synthetic generator has no nested 'random' split: use JSON-lines data for sku")
        data = _SyntheticPairs(int(rc.forget_data_path.split(":")[1]), cfg, splits, rc.seq_len, rc.seed)
    else:
        if not rc.tokenizer_path:
            raise SystemExit("JSON-lines data need tokenizer_path=<local LLaVA tokenizer directory>")
        data = _JsonlPairs(rc, cfg, rc.tokenizer_path)
    max_steps, warmup = steps_from_reference_formula(len(data), rc.batch_size, rc.gradient_accumulation_steps, world,
                                                     rc.num_epochs)
    if rc.max_steps:
        max_steps = rc.max_steps
    log(f"max_steps: {max_steps}")

    # mm/forget.py:126: adapters only when `LoRA.r != 0 and (l_norm_from != "zero" or l1_lambda == 0)`; an L1 penalty on
    # the distance from zero with LoRA.r set silently runs FULL fine-tuning in the reference, and so does this
    use_lora = rc.lora_r != 0 and (rc.l_norm_from != "zero" or rc.l1_lambda == 0)
    if rc.lora_r != 0 and not use_lora:
        log(f"LoRA.r={rc.lora_r} ignored: l1_lambda={rc.l1_lambda} with l_norm_from=zero runs full fine-tuning (mm/forget.py:126)")
    # LoRA.dropout (config/mm/forget.yaml:6, default 0.0): peft's nn.Dropout on the adapters' input, mask stream seeded by cfg.seed
    lora = LoraSpec(r=rc.lora_r, alpha=rc.lora_alpha, dropout=rc.lora_dropout, seed=rc.seed) if use_lora else None
    need_teacher = bool(rc.l1_lambda or rc.l0_lambda) or tu.loss_needs_teacher(lt)
    params = extra = None
    # What model_path holds is decided ONCE, by rank 0, before any rank can have written there (the merge below rewrites the
    # directory): a slower rank that looked for itself could find the adapter already retired, skip the merge, train on
    # unmerged weights and miss the barriers of the branch.
    has_ckpt = _has_checkpoint(rc.model_path)
    has_adapter = has_ckpt and os.path.exists(os.path.join(rc.model_path, "adapter_config.json"))
    if world > 1:
        import torch.distributed as dist

        seen = [has_ckpt, has_adapter]
        dist.broadcast_object_list(seen, src=0)
        has_ckpt, has_adapter = seen
    if has_ckpt:
        log("Loading from checkpoint")
        params, extra = load_pretrained(cfg, rc.model_path, return_extra=True)
    else:
        log(f"no safetensors under model_path={rc.model_path!r}: RANDOM-INIT weights of {cfg.name} (seed {rc.seed})")
    if has_adapter:
        # mm/forget.py:100-113 "Loading after merge and unload": model_path holds a LoRA run (this repo saves the base
        # weights next to the peft-layout adapter; the reference fetches the base from the hub).  Fold the adapter into
        # the base, keep the result as the new full checkpoint of that directory ("so that we can again do PEFT style
        # finetuning from scratch") and retire the adapter files so that the next load does not merge twice.
        from .checkpoint import load_adapter

        log("Loading after merge and unload")
        conf = json.load(open(os.path.join(rc.model_path, "adapter_config.json")))
        tmp = LlavaHIP(cfg, params, device=dev, lora=LoraSpec(r=int(conf["r"]), alpha=float(conf["lora_alpha"])), trainable=False)
        state, _, _ = load_adapter(cfg, rc.model_path, [n[:-len(".weight")] for n in tmp.lora])
        tmp.load_lora_state_(state)
        tmp.merge_lora_()
        params = {k: v.detach().clone() for k, v in tmp.w.items()}  # stays on the device (bf16): no 4-byte host copy of 7B
        if world > 1:
            dist.barrier()  # every rank has finished READING model_path (weights and adapter) before rank 0 rewrites it
        if rank == 0:
            # written under temporary names and moved into place file by file: a reader never sees a half-written shard
            stage = os.path.join(rc.model_path, ".merge_tmp")
            save_pretrained(cfg, tmp.w, stage, extra=extra)
            for f in sorted(os.listdir(stage)):
                os.replace(os.path.join(stage, f), os.path.join(rc.model_path, f))
            os.rmdir(stage)
            for f in ("adapter_config.json", "adapter_model.safetensors"):
                os.replace(os.path.join(rc.model_path, f), os.path.join(rc.model_path, f + ".merged"))
        del tmp
        if world > 1:
            dist.barrier()  # the directory is whole again before anyone goes on (all ranks took this branch: has_adapter is rank 0's)
    # mm/forget.py:117-118: `freeze_vision_module` "true" freezes the CLIP tower (the default, config/mm/forget.yaml:57); false
    # trains it with everything else in full fine-tuning (under LoRA peft freezes every base weight whatever the flag says)
    model = LlavaHIP(cfg, params, device=dev, lora=lora, seed=rc.seed, init="device" if params is None else "numpy",
                     freeze_vision=bool(rc.freeze_vision_module) or text)
    if rc.gradient_checkpointing:
        model.gradient_checkpointing_enable()
    if not rc.freeze_vision_module and not text:
        log("vision tower: " + ("TRAINABLE (freeze_vision_module=false)" if model.train_vision else
                                "base weights frozen by LoRA; its adapters train"))
    teacher = None
    if need_teacher:  # the reference loads the same checkpoint a second time (mm/forget.py:91-99)
        teacher = LlavaHIP(cfg, params, device=dev, trainable=False, seed=rc.seed, init="device" if params is None else "numpy")
    if text:  # nlp/forget.py:180-197 (oracle_model=, case-sensitive loss names, L1/L0 against the oracle)
        trainer = CustomTrainerForgetting(model=model, oracle_model=teacher, forget_loss=lt, loss_beta=rc.loss_beta,
                                          l1_lambda=rc.l1_lambda, l0_lambda=rc.l0_lambda)
    else:
        trainer = MMTrainerForgetting(model=model, teacher_model=teacher, forget_loss=lt, loss_beta=rc.loss_beta,
                                      l1_lambda=rc.l1_lambda, l0_lambda=rc.l0_lambda, l_norm_from=rc.l_norm_from,
                                      loss_args={"bad_weight": 2.5, "random_weight": 1.0, "normal_weight": 0.5})  # :186
    # overlap_optimizer: AdamW + W^T refresh of a step run on a side stream under the next step's forward (full fine-tuning,
    # fp32 moments, all-reduce mode; bit-identical); this loop only reads the parameters after the synchronize below
    engine = ForgetEngine(model, trainer, rc, total_steps=max_steps, warmup_steps=warmup, optimizer=rc.optimizer,
                          dp_mode=rc.dp_mode, overlap_optimizer=True)

    mine = shard_indices(len(data), rank, world)
    if not mine:
        raise SystemExit(f"rank {rank}: no samples ({len(data)} items on {world} ranks)")
    per_step = rc.batch_size * rc.gradient_accumulation_steps
    order_rng = random.Random(rc.seed)
    t0, seen, cursor, order = time.perf_counter(), 0, 0, []

    def next_micro_batches():
        nonlocal cursor, order
        mbs = []
        for _ in range(rc.gradient_accumulation_steps):
            if cursor + rc.batch_size > len(order):  # next epoch(s): reshuffle this rank's shard until a batch is full
                order = order[cursor:]
                cursor = 0
                while len(order) < rc.batch_size:
                    order = order + order_rng.sample(mine, len(mine))
            mbs.append(data.batch(order[cursor:cursor + rc.batch_size], dev))
            cursor += rc.batch_size
        return mbs

    mbs = next_micro_batches()
    for step in range(max_steps):
        st = engine.step(mbs, sync_loss="defer")
        # the step's kernels are queued: tokenising / resizing the NEXT step's micro-batches on the host runs under them,
        # and only then is the loss read back (the one sync of the step)
        mbs = next_micro_batches() if step + 1 < max_steps else None
        loss = float(st.loss_tensor)
        seen += per_step * world
        if step % max(1, max_steps // 100) == 0 or step == max_steps - 1:  # logging_steps=0.01 (mm/forget.py:153)
            log(json.dumps({"step": st.step, "loss": round(loss, 6), "learning_rate": st.lr,
                            "train_samples_per_second": round(seen / (time.perf_counter() - t0), 3)}))
    torch.cuda.synchronize()
    if rc.save_model and rank == 0:
        if lora is not None:
            save_pretrained(cfg, model.w, rc.save_dir, extra=extra, lora_state=model.lora_state(), lora_r=rc.lora_r,
                            lora_alpha=rc.lora_alpha, lora_dropout=rc.lora_dropout)
        else:
            save_pretrained(cfg, model.w, rc.save_dir, extra=extra)
        log("saved", rc.save_dir)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return engine


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    gpus = 1
    if "--gpus" in argv:
        i = argv.index("--gpus")
        gpus = int(argv[i + 1])
        del argv[i:i + 2]
    if gpus > 1 and "WORLD_SIZE" not in os.environ:
        # what `accelerate launch` does for the reference: one fresh process per GPU (this process has not touched a GPU)
        import socket
        import subprocess

        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), "-m", "multimodal_unlearning_amd.forget", *argv]
        return subprocess.run(cmd, env=env).returncode
    run(parse_overrides(argv, ForgetRunConfig()))
    return 0


if __name__ == "__main__":
    sys.exit(main())
