"""Drop-in counterpart of the reference's MMTrainerForgetting (mm/trainer.py:112-492).

Same constructor kwargs (`forget_loss, teacher_model, loss_beta, l1_lambda, l0_lambda, l_norm_from, loss_args`,
mm/trainer.py:113-123; call site mm/forget.py:171-187), same `compute_loss(model, inputs, return_outputs=False,
*args, **kwargs)` signature (mm/trainer.py:168), same nested-dict `inputs` (mm/dataset.py:283-288), same
exceptions (ValueError for unknown losses / bad l_norm_from / L1 without teacher / LoRA with "zero").
`model(**split)` must return an object with `.loss` and `.logits` (our LlavaHIP, or anything HF-shaped).

Differences on purpose: both "KL" and "kl" need/accept a teacher; `model.device = ...` is not
assigned; `return_outputs=True` returns the last student output for every branch; the dead forwards of the dpo
branch run only if loss_args["dpo_dead_compute"] is set (they never influence the loss, mm/trainer.py:394-396).

MI355X addition — paired forget/retain fusion: when both splits have the same sequence length, the branches that
only need CE terms (grad_diff_forget_ce_retain_ce, eco_ft, idk) run ONE forward over the concatenated batch and get
the per-split means from the segmented CE kernel.  Rows never interact, so each split's loss and gradient are
the same as with two separate forwards; it doubles the GEMM M dimension and halves launches.
"""
from __future__ import annotations



import torch

from . import trainer_utils as tu
from .llava import attach_host, cat_with_host, host_of, segment_ce_losses

_SPLIT_KEYS = ("input_ids", "attention_mask", "pixel_values", "labels")


class _L1L0Fn(torch.autograd.Function):
    """reg = l1 * ||d||_1 + l0 * ||d||_0 with d = bf16(flat - teacher_flat) (or flat), mm/trainer.py:451-481, as HIP
    kernels over the flat buffers.  backward() accumulates l1 * g * sign(d) into the model's flat gradient buffer and
    returns no tensor for it (same convention as the model's autograd edge)."""

    @staticmethod
    def forward(ctx, flat, model, tflat, l1, l0):
        from . import hip

        out = torch.zeros(2, device=flat.device, dtype=torch.float64)
        hip.l1l0_fwd(flat.detach(), tflat, out)
        ctx.model, ctx.tflat, ctx.l1 = model, tflat, l1
        return (l1 * out[0] + l0 * out[1]).float()

    @staticmethod
    def backward(ctx, gout):
        from . import hip

        if ctx.l1:
            model = ctx.model
            g = model.ensure_grad()
            hip.l1_bwd(model.train_flat.detach(), ctx.tflat, g, ctx.l1, gout.detach().float().reshape(1).contiguous())
        return None, None, None, None, None


class _MseFn(torch.autograd.Function):
    """mse_loss(act, ref) over a captured activation [B,S,H] bf16 with ref = another activation (RMU's retain term,
    mm/trainer.py:321-324) or a control vector [H] broadcast over every token (the text trainer's forget term,
    nlp/dataloader.py:355): HIP kernels mmu_mse_fwd / mmu_mse_bwd, fp32 math, deterministic reduction."""

    @staticmethod
    def forward(ctx, act, ref, cvec):
        from . import hip

        act = act.contiguous()
        H = act.shape[-1]
        out = torch.empty(hip.RED_FLOATS, device=act.device, dtype=torch.float32)
        hip.mse_fwd(act, ref, cvec, H, out)
        ctx.save_for_backward(act, ref, cvec)
        return out[0] / act.numel()

    @staticmethod
    def backward(ctx, g):
        from . import hip

        act, ref, cvec = ctx.saved_tensors
        coef = (g.detach().float() * (2.0 / act.numel())).reshape(1).contiguous()
        da = torch.empty_like(act)
        hip.mse_bwd(act, ref, cvec, act.shape[-1], coef, da)
        return da, None, None


class _RmuForgetFn(torch.autograd.Function):
    """mm RMU forget term: mse_loss(act.mean(dim=1) [B,H], control [Bc,1,H]) — which broadcasts to [Bc,B,H], a quirk of
    mm/trainer.py:316 kept on purpose — as mmu_seqmean_fwd + mmu_rmu_pair_fwd; backward writes the [B,S,H] gradient of
    the captured activation in one pass (mmu_rmu_pair_bwd)."""

    @staticmethod
    def forward(ctx, act, ctrl):
        from . import hip

        act = act.contiguous()
        B, S, H = act.shape
        mean = torch.empty(B, H, device=act.device, dtype=torch.float32)
        hip.seqmean_fwd(act, mean)
        out = torch.empty(hip.RED_FLOATS, device=act.device, dtype=torch.float32)
        hip.rmu_pair_fwd(mean, ctrl, out)
        ctx.save_for_backward(mean, ctrl)
        ctx.meta = (S, B * (ctrl.numel() // H) * H, act.dtype)
        return out[0] / ctx.meta[1]

    @staticmethod
    def backward(ctx, g):
        from . import hip

        mean, ctrl = ctx.saved_tensors
        S, n, dt = ctx.meta
        coef = (g.detach().float() * (2.0 / n)).reshape(1).contiguous()
        da = torch.empty(mean.shape[0], S, mean.shape[1], device=mean.device, dtype=dt)
        hip.rmu_pair_bwd(mean, ctrl, S, coef, da)
        return da, None


def _inner(model):
    """The module behind a wrapper (torch DDP keeps it in `.module`): attributes are read there, CALLS go through the wrapper."""
    return getattr(model, "module", model)


def _slice_k(v, i):
    """v[:, i] of a nested [B, K, ...] split tensor, carrying the host view of labels / masks along (a fresh view has
    none, and looking at it would cost a device sync on every call)."""
    out = v[:, i]
    h = host_of(v)
    if h is not None and out.device.type != "cpu":
        attach_host(out, h[:, i])
    return out


# losses whose per-micro-batch value is a mean over equal-sized sets (KL batchmean over B*S rows, NPO mean over B*S*V,
# dpo's mean over sequences), a per-segment token mean (every CE term, through the segmented CE kernel) or a plain sum
# (entropy, divided by g): the mean over g micro-batches equals ONE evaluation on their concatenation.  rmu (control
# vector of batch shape, [B,B,H] broadcast) and sku (nested [B,K] split) are evaluated micro-batch by micro-batch.
_MERGEABLE = ("retain_ft", "grad_ascent", "scrub", "kl", "llmu", "npo", "idk", "eco_ft", "dpo")
_CE_ONLY = ("retain_ft", "grad_ascent", "grad_diff_forget_ce_retain_ce", "idk", "eco_ft")


class MMTrainerForgetting:
    def __init__(self, *args, **kwargs):
        self._pop_forget_kwargs(kwargs)
        self._init_host(*args, **kwargs)

    def _pop_forget_kwargs(self, kwargs):
        """The plug-in's own constructor arguments (mm/trainer.py:113-123); what is left belongs to the host loop."""
        self.loss_type = kwargs.pop("forget_loss").lower()
        self.teacher_model = kwargs.pop("teacher_model", None)
        self.loss_beta = kwargs.pop("loss_beta", 1.0)
        self.l1_lambda = kwargs.pop("l1_lambda", 0.0)
        self.l0_lambda = kwargs.pop("l0_lambda", 0.0)
        self.l_norm_from = kwargs.pop("l_norm_from", "zero")
        self.loss_args = kwargs.pop("loss_args", {}) or {}
        if self.l1_lambda and self.l1_lambda != 0 and self.l_norm_from not in ("zero", "init"):
            raise ValueError(f"Invalid l_norm_from {self.l_norm_from}, should be 'zero' or 'init'")
        self.fuse_pairs = bool(self.loss_args.get("fuse_pairs", True))
        # MI355X addition: forwards of which only `.loss` is read run the lm_head on the supervised rows (A/B switch)
        self.loss_only_head = bool(self.loss_args.get("loss_only_head", True))
        self.control_vec = None
        self._merge = 1  # > 1 only inside compute_loss(..., _merge=g)
        self.is_deepspeed_enabled = False
        if tu.loss_needs_teacher(self.loss_type) and self.teacher_model is None:
            raise ValueError(f"forget_loss={self.loss_type} evaluates a frozen teacher: pass teacher_model")
        if self.teacher_model is not None and hasattr(self.teacher_model, "parameters"):
            for p in self.teacher_model.parameters():  # mm/trainer.py:157-160
                p.requires_grad = False

    def _init_host(self, *args, **kwargs):
        """Stand-alone use (our ForgetEngine drives the loop): keep the HF-Trainer style leftovers as attributes.
        `hosted_by_hf_trainer()` replaces this with transformers.Trainer.__init__ (mm/trainer.py:125)."""
        self.model = kwargs.pop("model", args[0] if args else None)
        self.args = kwargs.pop("args", None)
        self.train_dataset = kwargs.pop("train_dataset", None)
        self.data_collator = kwargs.pop("data_collator", None)
        self.tokenizer = kwargs.pop("tokenizer", None)

    @staticmethod
    def to_device(device, inputs):  # mm/trainer.py:164-166
        out = {}
        for k, v in inputs.items():
            d = v.to(device)
            # collator output lives on the host: the device copy remembers it, so the model can look at masks / labels
            # (padding? which rows are supervised?) without a device sync
            if d is not v and k in ("attention_mask", "labels") and v.device.type == "cpu":
                attach_host(d, v)
            out[k] = d
        return out

    # ------------------------------------------------------------------------ merged micro-batches (MI355X addition)
    def can_merge(self, model) -> bool:
        lt = self.loss_type
        return hasattr(model, "loss_segments") and (lt in _MERGEABLE or lt.startswith("grad_diff"))

    def merge_micro_batches(self, model, micro_batches, pad_token_id: int = 0):
        """ONE nested input dict that concatenates the gradient-accumulation micro-batches along the batch dimension, or
        None when they cannot be evaluated together (different keys / batch sizes, a loss outside `can_merge`).
        Micro-batches of different sequence length are LEFT-padded to the longest (pad id, mask 0, label -100) for the
        CE-only losses — padded keys are masked and RoPE scores depend on position differences only; losses that
        normalise over all B*S rows (KL batchmean, NPO, entropy) need equal shapes, as padding rows would be counted.
        `compute_loss(model, merged, _merge=len(micro_batches))` then returns the mean of the individual losses."""
        if len(micro_batches) < 2 or not self.can_merge(model) or not all(isinstance(mb, dict) for mb in micro_batches):
            return None
        first = micro_batches[0]
        if any(mb.keys() != first.keys() for mb in micro_batches):
            return None
        out = {}
        for split in first:
            parts = [mb[split] for mb in micro_batches]
            if any(p.keys() != parts[0].keys() for p in parts):
                return None
            ids = [p["input_ids"] for p in parts]
            if any(x.dim() != 2 or x.shape[0] != ids[0].shape[0] for x in ids):
                return None
            S = max(x.shape[1] for x in ids)
            ragged = any(x.shape[1] != S for x in ids)
            if ragged and (self.loss_type not in _CE_ONLY or "attention_mask" not in parts[0]):
                return None
            merged = {}
            for k in parts[0]:
                ts = [p[k] for p in parts]
                if k in ("input_ids", "attention_mask", "labels") and ragged:
                    fill = {"input_ids": pad_token_id, "attention_mask": 0, "labels": -100}[k]
                    ts = [t if t.shape[1] == S else torch.cat([t.new_full((t.shape[0], S - t.shape[1]), fill), t], dim=1)
                          for t in ts]
                elif any(t.shape[1:] != ts[0].shape[1:] for t in ts):
                    return None
                merged[k] = (cat_with_host if k in ("attention_mask", "labels") else torch.cat)(ts, dim=0)
            out[split] = merged
        return out

    # ------------------------------------------------------------------------------------------------
    def _teacher(self, split):
        with torch.no_grad():
            return self.teacher_model(**split)

    @staticmethod
    def _can_fuse(a, b):
        if a is None or b is None or a["input_ids"].shape != b["input_ids"].shape:
            return False
        pa, pb = a.get("pixel_values"), b.get("pixel_values")
        if (pa is None) != (pb is None):
            return False
        return pa is None or pa.shape[1:] == pb.shape[1:]

    def _paired_ce(self, model, a, b, need_outputs=False):
        """One forward over cat(a, b); returns (ce_a, ce_b, outputs).  When nobody asked for the outputs and the model
        can do it, the head runs on the supervised rows only (LlavaHIP.ce_losses) and `outputs` carries no logits."""
        g = self._merge
        cat = {k: (cat_with_host if k == "attention_mask" else torch.cat)([a[k], b[k]], dim=0)
               for k in _SPLIT_KEYS if k in a and k != "labels"}
        labels = cat_with_host([a["labels"], b["labels"]], dim=0)
        if not need_outputs and getattr(_inner(model), "supports_loss_only", False) and self.loss_only_head:
            ce = model(labels=labels, loss_only=True, ce_segments=2 * g, **cat).loss   # through forward(): a DDP wrapper sees it
            from .llava import LlavaOutput
            return ce[:g].mean(), ce[g:].mean(), LlavaOutput(loss=None, logits=None)
        out = model(**cat)
        ce = segment_ce_losses(out.logits, labels.to(out.logits.device), 2 * g)
        return ce[:g].mean(), ce[g:].mean(), out

    def _loss_of(self, model, split, need_outputs=False):
        """`model(**split)` when only `.loss` will be read: models that understand `loss_only=True` (LlavaHIP) skip the
        logits of unsupervised positions; `need_outputs` (the caller returns the outputs) or a foreign model keep the
        full forward."""
        if not need_outputs and getattr(_inner(model), "supports_loss_only", False) and self.loss_only_head and "labels" in split:
            return model(**split, loss_only=True)
        return model(**split)

    def compute_loss(self, model, inputs, return_outputs=False, *args, **kwargs):
        dev = next(model.parameters()).device
        td = MMTrainerForgetting.to_device  # (the text-trainer subclass overrides to_device for tuples)
        retain_inputs = td(dev, inputs["retain"]) if "retain" in inputs else None
        idk_inputs = td(dev, inputs["idk"]) if "idk" in inputs else None
        forget_inputs = td(dev, inputs["forget"]) if "forget" in inputs else None
        random_inputs = td(dev, inputs["random"]) if "random" in inputs else None
        lt, beta = self.loss_type, self.loss_beta
        outputs = None
        # `_merge=g`: `inputs` is the concatenation of g gradient-accumulation micro-batches (merge_micro_batches) and the
        # value returned is the MEAN of the g losses the host loop would have computed one by one
        g = int(kwargs.get("_merge", 1))
        if g > 1:
            if not self.can_merge(model):
                raise ValueError(f"forget_loss={self.loss_type} cannot evaluate merged micro-batches on this model")
            self._merge, keep_seg, model.loss_segments = g, model.loss_segments, g
            try:
                return self.compute_loss(model, inputs, return_outputs, *args, **{**kwargs, "_merge": 1, "_merged": g})
            finally:
                self._merge, model.loss_segments = 1, keep_seg
        # does anybody read the outputs?  (the text trainer calls this method with return_outputs=True for its own
        # bookkeeping and passes what ITS caller asked for in `_need_outputs`)
        need_out = bool(kwargs.get("_need_outputs", return_outputs))

        if lt == "retain_ft":
            outputs = self._loss_of(model, retain_inputs, need_out)
            loss = outputs.loss

        elif lt == "grad_ascent":
            outputs = self._loss_of(model, forget_inputs, need_out)
            loss = -1 * outputs.loss

        elif lt.startswith("grad_diff"):
            if "forget_ce" in lt and "retain_ce" in lt and self.fuse_pairs and self._can_fuse(forget_inputs, retain_inputs):
                forget_loss, retain_loss, outputs = self._paired_ce(model, forget_inputs, retain_inputs, need_out)
            else:
                forget_outputs = self._loss_of(model, forget_inputs, "forget_ce" not in lt)
                retain_outputs = self._loss_of(model, retain_inputs, need_out or "retain_ce" not in lt)
                outputs = retain_outputs
                if "forget_ce" in lt:
                    forget_loss = forget_outputs.loss
                elif "forget_entropy" in lt:
                    forget_loss = tu.entropy_term(forget_outputs.logits) / self._merge  # a plain SUM over all rows
                elif "forget_kl" in lt:
                    oracle = self._teacher(forget_inputs)
                    forget_loss = tu.kl_div_batchmean(forget_outputs.logits, oracle.logits, student_is_input=True)
                else:
                    raise ValueError(f"Invalid loss type on forget {self.loss_type}")
                if "retain_ce" in lt:
                    retain_loss = retain_outputs.loss
                elif "retain_kl" in lt:
                    oracle = self._teacher(retain_inputs)
                    retain_loss = tu.kl_div_batchmean(retain_outputs.logits, oracle.logits, student_is_input=True)
                else:
                    raise ValueError(f"Invalid loss type on retain {self.loss_type}")
            loss = -1 * beta * forget_loss + retain_loss

        elif lt == "scrub":
            forget_outputs = model(**forget_inputs)
            oracle_f = self._teacher(forget_inputs)
            kl_forget = tu.kl_div_batchmean(oracle_f.logits, forget_outputs.logits, student_is_input=False)
            retain_outputs = model(**retain_inputs)
            oracle_r = self._teacher(retain_inputs)
            kl_retain = tu.kl_div_batchmean(oracle_r.logits, retain_outputs.logits, student_is_input=False)
            loss = -1 * beta * kl_forget + kl_retain + retain_outputs.loss
            outputs = retain_outputs

        elif lt == "kl":
            forget_outputs = self._loss_of(model, forget_inputs)
            forget_loss = -1 * forget_outputs.loss
            oracle_r = self._teacher(retain_inputs)
            retain_outputs = model(**retain_inputs)
            retain_loss = tu.kl_div_batchmean(retain_outputs.logits, oracle_r.logits, student_is_input=True)
            loss = forget_loss + retain_loss
            outputs = retain_outputs

        elif lt == "llmu":
            forget_loss = -1 * self._loss_of(model, forget_inputs).loss
            random_loss = self._loss_of(model, idk_inputs).loss
            retain_outputs = model(**retain_inputs)
            oracle_r = self._teacher(retain_inputs)
            retain_loss = tu.kl_div_batchmean(oracle_r.logits, retain_outputs.logits, student_is_input=False)
            loss = forget_loss + retain_loss + random_loss
            outputs = retain_outputs

        elif lt == "npo":
            outputs = model(**forget_inputs)
            oracle_f = self._teacher(forget_inputs)
            loss = tu.npo_loss(outputs.logits, oracle_f.logits, beta)

        elif lt == "idk":
            if self.fuse_pairs and self._can_fuse(retain_inputs, idk_inputs):
                a, b, outputs = self._paired_ce(model, retain_inputs, idk_inputs, need_out)
                loss = a + b
            else:
                retain_outputs = self._loss_of(model, retain_inputs)
                outputs = self._loss_of(model, idk_inputs, need_out)
                loss = retain_outputs.loss + outputs.loss

        elif lt == "eco_ft":
            if self.fuse_pairs and self._can_fuse(forget_inputs, retain_inputs):
                a, b, outputs = self._paired_ce(model, forget_inputs, retain_inputs, need_out)
                loss = beta * a + b
            else:
                forget_outputs = self._loss_of(model, forget_inputs)
                outputs = self._loss_of(model, retain_inputs, need_out)
                loss = beta * forget_outputs.loss + outputs.loss

        elif lt == "dpo":
            # mm/trainer.py:364-398: the log-sigmoid DPO value is overwritten twice; what is returned (and
            # differentiated) is mean_b sum_t CE on the idk split.  The three other forwards are dead compute.
            idk_outputs = model(**idk_inputs)
            if self.loss_args.get("dpo_dead_compute", False):
                outputs = model(**forget_inputs)
                self._teacher(idk_inputs)
                self._teacher(forget_inputs)
            else:
                outputs = idk_outputs
            image_token_index = _inner(model).config.image_token_index
            idk_logits = tu.remove_image_tokens(idk_inputs["input_ids"], idk_outputs.logits, image_token_index)
            idk_loss_current = -1 * tu.get_batch_loss(idk_logits, idk_inputs["labels"])
            variant = self.loss_args.get("dpo_variant", "reference")
            if variant == "reference":      # the LAST of the three assignments at mm/trainer.py:394-396 is what the reference trains on
                loss = -idk_loss_current.mean()
            elif variant in ("logsigmoid", "pi_logratios"):
                # The two values the reference computes and overwrites (:390-395), offered behind a flag:
                # "logsigmoid" = DPO proper, -logsigmoid(0.1 * (pi_logratios - ref_logratios)).mean() with the frozen
                # teacher as reference policy; "pi_logratios" = -(log p(idk) - log p(forget)).mean().  Per-sequence
                # log-likelihoods are `get_batch_loss` sums (HIP CE kernels); the [B]-vector arithmetic on top is host glue.
                def seq_logp(m, split, grad):
                    with torch.enable_grad() if grad else torch.no_grad():
                        lg = tu.remove_image_tokens(split["input_ids"], m(**{k: v for k, v in split.items() if k != "labels"}).logits,
                                                    image_token_index)
                        return -1 * tu.get_batch_loss(lg, split["labels"])
                forget_loss_current = seq_logp(model, forget_inputs, True)
                pi_logratios = idk_loss_current - forget_loss_current
                if variant == "pi_logratios":
                    loss = -pi_logratios.mean()
                else:
                    ref_logratios = seq_logp(self.teacher_model, idk_inputs, False) - seq_logp(self.teacher_model, forget_inputs, False)
                    loss = -torch.nn.functional.logsigmoid(0.1 * (pi_logratios - ref_logratios)).mean()   # beta = 0.1 (:393)
            else:
                raise ValueError(f"Invalid dpo_variant {variant!r}: expected 'reference', 'logsigmoid' or 'pi_logratios'")

        elif lt == "sku":
            bad_loss = self._loss_of(model, forget_inputs).loss
            # [B, K, ...] -> K batches; the reference transposes in place (mm/trainer.py:420-421); we do not mutate
            K = random_inputs["input_ids"].shape[1]
            random_loss = 0.0
            for i in range(K):
                random_loss = random_loss + self._loss_of(model, {k: _slice_k(v, i) for k, v in random_inputs.items()}).loss
            oracle_r = self._teacher(retain_inputs)
            outputs = model(**retain_inputs)
            normal_loss = -1 * tu.kl_div_batchmean(outputs.logits, oracle_r.logits, student_is_input=True)
            loss = (self.loss_args["bad_weight"] * bad_loss + self.loss_args["random_weight"] * random_loss
                    + self.loss_args["normal_weight"] * normal_loss)

        elif lt == "rmu":
            # mm/trainer.py:302-326: forward hooks on decoder layer 7; MSE towards a cached random control vector
            # (forget) + MSE to the teacher's activation (retain).  The hidden-state sweep stops at that layer.
            layer = int(self.loss_args.get("rmu_layer", 7))
            strip = lambda d: {k: v for k, v in d.items() if k != "labels"}
            hidden = (lambda m, split: m(**strip(split), hidden_layer=layer).hidden_states)   # via forward(): see LlavaHIP.forward
            forget_act = hidden(model, forget_inputs)
            if getattr(self, "control_vec", None) is None:
                rand_vec = torch.rand(forget_act.shape[0], 1, forget_act.shape[-1], dtype=torch.float32,
                                      device=forget_act.device)
                self.control_vec = rand_vec / torch.norm(rand_vec) * 300
            retain_act = hidden(model, retain_inputs)
            with torch.no_grad():
                oracle_act = hidden(self.teacher_model, retain_inputs)
            # fused mean-over-sequence + MSE kernels on the bf16 activations (no eager substitute: the kernels reject
            # anything but device bf16).  NOTE the reference's shapes: input [B,H] vs target [B,1,H] broadcast to [B,B,H]
            ctrl = self.control_vec.to(forget_act.device, torch.float32).reshape(-1, forget_act.shape[-1]).contiguous()
            forget_loss = _RmuForgetFn.apply(forget_act, ctrl) * beta
            retain_loss = _MseFn.apply(retain_act, oracle_act.contiguous(), None)
            loss = forget_loss + retain_loss

        else:
            raise ValueError(f"Invalid loss type {self.loss_type}")

        if not kwargs.get("_skip_reg", False):
            loss = loss + self._l_norm_regulariser(model)
        return (loss, outputs) if return_outputs else loss

    # mm/trainer.py:451-481
    def _l_norm_regulariser(self, model):
        l1, l0 = self.l1_lambda, self.l0_lambda
        if not ((l1 is not None and l1 != 0) or (l0 is not None and l0 != 0)):
            return 0.0
        model = _inner(model)
        lora = tu.has_lora_adapter(model)
        if self.l_norm_from == "init" and not lora and self.teacher_model is None:
            raise ValueError("Teacher model is required for L1\\L0 regularization during training without LORA!")
        if self.l_norm_from == "zero" and lora:
            raise ValueError("L1\\L0 regularization is not supported for LORA!")
        flat = getattr(model, "train_flat", None)
        against_teacher = self.l_norm_from == "init" and not lora
        tflat = getattr(self.teacher_model, "train_flat", None) if against_teacher else None
        if flat is not None and flat.is_cuda and (not against_teacher or (tflat is not None and tflat.shape == flat.shape)):
            # LlavaHIP keeps every trainable tensor in ONE flat parameter: one streaming HIP pass over it (and over the
            # teacher's) instead of the reference's `p - tp` temporaries + 13.5 GB torch.cat; the backward adds
            # lambda1 * sign(d) straight into the flat gradient.  Nothing of parameter size is allocated here.
            if tflat is not None:
                tflat = tflat.detach()
                if tflat.device != flat.device:
                    tflat = tflat.to(flat.device)
            return _L1L0Fn.apply(flat, model, tflat, float(l1 or 0.0), float(l0 or 0.0))
        params = []
        if self.l_norm_from == "init":
            if lora:
                params = [p.view(-1) for p in model.parameters() if p.requires_grad]
            else:
                for p, tp in zip(model.parameters(), self.teacher_model.parameters()):
                    assert p.shape == tp.shape
                    if p.requires_grad:
                        params.append((p - tp.to(p.device)).view(-1))
        elif self.l_norm_from == "zero":
            params = [p.view(-1) for p in model.parameters() if p.requires_grad]
        reg = 0.0
        cat = torch.cat(params).float()
        if l1 is not None and l1 != 0:
            reg = reg + l1 * torch.norm(cat, p=1)
        if l0 is not None and l0 != 0:
            reg = reg + l0 * torch.norm(cat, p=0)
        return reg

    def prediction_step(self, model, inputs, prediction_loss_only: bool, ignore_keys=None):  # mm/trainer.py:485-492
        input_ids, labels, attention_mask = inputs
        with torch.no_grad():
            outputs = model(input_ids=input_ids, labels=labels, attention_mask=attention_mask)
        return (outputs.loss, outputs.logits, labels)


class CustomTrainerForgetting(MMTrainerForgetting):
    """Counterpart of the text trainer nlp/dataloader.py:93-485 (BASELINE config 1): same dispatcher on TUPLE inputs
    `[forget, retain(, idk)]`, each `(input_ids, labels, attention_mask)` as custom_data_collator_forget builds them
    (nlp/dataloader.py:497-517), `oracle_model` instead of `teacher_model`, CASE-SENSITIVE loss names as in the
    reference ("KL", "LLMU", "RMU", "grad_diff_forget_KL_retain_ce"), and the text trainer's own formulas where they
    differ from mm: NPO = 2/beta * mean(log(1 + log(p/p_T)^beta)) (:396-401), idk = ONE forward over cat(forget,
    retain) with a single mean CE (:403-410), RMU with a fresh [1,1,H] control vector per call (:339-367), L1/L0
    always against the oracle unless LoRA (:460-481).  The model is the same LlavaHIP run without pixel_values."""

    def __init__(self, *args, **kwargs):
        self.loss_type_raw = kwargs.get("forget_loss")
        kwargs.setdefault("teacher_model", kwargs.pop("oracle_model", None))
        kwargs.setdefault("l_norm_from", "init")
        kwargs.setdefault("loss_args", {})
        super().__init__(*args, **kwargs)
        self.oracle_model = self.teacher_model

    @staticmethod
    def to_device(device, inputs):  # nlp/dataloader.py:140-142
        out = tuple(t.to(device) for t in inputs[:3])
        for d, v in zip(out[1:], inputs[1:3]):  # labels, attention_mask: the device copies remember the collator's tensors
            if d is not v and v.device.type == "cpu":
                attach_host(d, v)
        return out

    @staticmethod
    def _split(t):
        ids, labels, mask = t
        return {"input_ids": ids, "attention_mask": mask, "labels": labels}

    def can_merge(self, model) -> bool:
        lt = self.loss_type_raw
        return hasattr(model, "loss_segments") and (lt in ("retain_ft", "grad_ascent", "scrub", "eco_ft", "dpo", "KL", "LLMU", "npo")
                                                    or lt.startswith("grad_diff"))

    def merge_micro_batches(self, model, micro_batches, pad_token_id: int = 0):
        """Tuple form of MMTrainerForgetting.merge_micro_batches: [(ids, labels, mask) x (forget, retain[, idk])] per
        micro-batch -> one such list over the concatenated batch (equal shapes only: the text collator pads every batch
        to max_length, nlp/data_module.py:10-43).  `idk` (one mean over cat(forget, retain)) and RMU are not merged."""
        if len(micro_batches) < 2 or not self.can_merge(model):
            return None
        first = micro_batches[0]
        if any(len(mb) != len(first) for mb in micro_batches):
            return None
        out = []
        for j in range(len(first)):
            parts = [mb[j][:3] for mb in micro_batches]
            if any(t.shape != parts[0][i].shape for p in parts for i, t in enumerate(p)):
                return None
            out.append(tuple((torch.cat if i == 0 else cat_with_host)([p[i] for p in parts], dim=0) for i in range(3)))
        return out

    def compute_loss(self, model, inputs, return_outputs=False, *args, **kwargs):
        g = int(kwargs.pop("_merge", 1))
        if g > 1:
            if not self.can_merge(model):
                raise ValueError(f"forget_loss={self.loss_type_raw} cannot evaluate merged micro-batches on this model")
            self._merge, keep_seg, model.loss_segments = g, model.loss_segments, g
            try:
                return self.compute_loss(model, inputs, return_outputs, *args, **kwargs)
            finally:
                self._merge, model.loss_segments = 1, keep_seg
        forget_t, retain_t, *idk_t = inputs
        dev = next(model.parameters()).device
        f, r = self._split(self.to_device(dev, forget_t)), self._split(self.to_device(dev, retain_t))
        nested = {"forget": f, "retain": r}
        if idk_t:
            nested["idk"] = self._split(self.to_device(dev, idk_t[0]))
        lt, beta = self.loss_type_raw, self.loss_beta
        outputs = None
        if lt == "npo":
            outputs = model(**f)
            with torch.no_grad():
                oracle = self.oracle_model(**f)
            n = outputs.logits.numel()
            from . import hip
            from .llava import row_loss_sum
            loss = 2 / beta * row_loss_sum(outputs.logits, oracle.logits, hip.NPO_NLP, beta) / n
        elif lt == "idk":
            cat = {k: (torch.cat if k == "input_ids" else cat_with_host)((f[k], r[k]), dim=0)
                   for k in ("input_ids", "attention_mask", "labels")}
            outputs = self._loss_of(model, cat, return_outputs)
            loss = outputs.loss
        elif lt == "RMU":
            layer = int(self.loss_args.get("rmu_layer", 7))
            strip = lambda d: {k: v for k, v in d.items() if k != "labels"}
            hidden = (lambda m, split: m(**strip(split), hidden_layer=layer).hidden_states)
            fa = hidden(model, f)
            cv = self.control_vec
            if cv is None:  # the text trainer draws a new vector on every call (nlp/dataloader.py:347-354)
                rand_vec = torch.rand(1, 1, fa.shape[-1], dtype=torch.float32, device=fa.device)
                cv = rand_vec / torch.norm(rand_vec) * 300
            ra = hidden(model, r)
            with torch.no_grad():
                oa = hidden(self.oracle_model, r)
            cvec = cv.to(fa.device, torch.float32).reshape(-1).contiguous()
            loss = _MseFn.apply(fa, None, cvec) * beta + _MseFn.apply(ra, oa.contiguous(), None)
        elif lt in ("retain_ft", "grad_ascent", "scrub", "eco_ft", "dpo", "KL", "LLMU") or lt.startswith("grad_diff"):
            # identical arithmetic to the mm dispatcher (whose names are the lower-cased ones; "forget_KL" -> "forget_kl")
            keep, self.loss_type = self.loss_type, lt.lower()
            try:
                if lt.startswith("grad_diff") and "forget_ce" not in lt and "forget_entropy" not in lt and "forget_KL" not in lt:
                    raise ValueError(f"Invalid loss type on forget {lt}")
                res = MMTrainerForgetting.compute_loss(self, model, nested, return_outputs=True, _skip_reg=True,
                                                       _need_outputs=return_outputs)
            finally:
                self.loss_type = keep
            loss, outputs = res
        else:
            raise ValueError(f"Invalid loss type {lt}")
        loss = loss + self._l_norm_regulariser(model)
        return (loss, outputs) if return_outputs else loss


def hosted_by_hf_trainer(plugin=MMTrainerForgetting):
    """The plug-in as a real `transformers.Trainer` subclass — the reference's arrangement (`class
    MMTrainerForgetting(Trainer)`, mm/trainer.py:112; built at mm/forget.py:171-187): HF's loop (`training_step`,
    gradient accumulation, clipping, optimiser, scheduler, logging) calls this class's `compute_loss` /
    `prediction_step`.  Usage: `hosted_by_hf_trainer()(model=..., args=TrainingArguments(...), train_dataset=...,
    data_collator=..., forget_loss=..., teacher_model=..., ...)`.  transformers is imported here, not at module import:
    the stand-alone path (ForgetEngine) does not need it."""
    import transformers

    class Hosted(plugin, transformers.Trainer):
        def _init_host(self, *args, **kwargs):
            import os

            m = kwargs.get("model", args[0] if args else None)
            if int(os.environ.get("WORLD_SIZE", "1")) > 1 and hasattr(m, "train_flat") and not getattr(m, "ddp_compatible", False):
                # transformers.Trainer wraps the model in torch DDP (mm/forget.py:161 ddp_find_unused_parameters), whose reducer
                # listens on the parameter's AccumulateGrad node; a model that accumulates into a flat gradient buffer by hand
                # and returns no gradient would never be reduced and the replicas would drift apart silently.  LlavaHIP can
                # hand its gradient to autograd instead (one more parameter-sized tensor); anything else must say it is safe.
                if hasattr(m, "enable_autograd_gradients"):
                    m.enable_autograd_gradients()
                else:
                    raise RuntimeError(
                        "hosted_by_hf_trainer() at WORLD_SIZE > 1: torch DDP cannot see this model's hand-written gradient "
                        "accumulation.  Run data-parallel training through multimodal_unlearning_amd.engine.ForgetEngine "
                        "(python -m multimodal_unlearning_amd.forget --gpus N ...), which owns the bucketed RCCL exchange.")
            transformers.Trainer.__init__(self, *args, **kwargs)
            # the reference's pinned transformers (4.45.1) divides every micro-batch loss by the accumulation steps;
            # 5.x skips that for models whose forward takes **kwargs unless this is False
            self.model_accepts_loss_kwargs = False

        def create_optimizer(self, *a, **kw):
            """transformers.Trainer.create_optimizer for a model that keeps its trainable tensors in ONE flat parameter
            (LlavaHIP): HF would build torch AdamW over a single parameter and decay everything, norm weights and biases
            included; the fused HIP AdamW with HF's no-decay ranges takes its place (`optim` names with "8bit", the
            reference's adamw_bnb_8bit, select the block-wise 8-bit moments).  Other models get HF's optimiser."""
            m = getattr(self, "model", None)
            if self.optimizer is None and m is not None and hasattr(m, "train_flat") and hasattr(m, "range_of"):
                from .engine import FlatAdamW

                args = self.args
                name = getattr(args, "optim", "")
                kind = "adamw8bit" if "8bit" in str(getattr(name, "value", name)) else "adamw"  # OptimizerNames enum or str
                self.optimizer = FlatAdamW(m, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2),
                                           eps=args.adam_epsilon, weight_decay=args.weight_decay, optimizer=kind)
                return self.optimizer
            return transformers.Trainer.create_optimizer(self, *a, **kw)

        def _save(self, output_dir=None, state_dict=None):
            """transformers.Trainer._save writes `model.state_dict()` for anything that is not a PreTrainedModel — for
            LlavaHIP that is one flat tensor no loader understands.  Checkpoints (`save_strategy="steps"`,
            mm/forget.py:158-160) and `trainer.save_model()` go through the model's own HF-layout writer instead."""
            m = getattr(self, "model", None)
            if m is None or not (hasattr(m, "train_flat") and hasattr(m, "save_pretrained")):
                return transformers.Trainer._save(self, output_dir, state_dict)
            import os

            out = output_dir if output_dir is not None else self.args.output_dir
            os.makedirs(out, exist_ok=True)
            m.save_pretrained(out)
            torch.save(self.args, os.path.join(out, "training_args.bin"))

    Hosted.__name__ = Hosted.__qualname__ = plugin.__name__ + "HF"
    return Hosted
