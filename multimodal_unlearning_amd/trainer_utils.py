"""Host-side mirror of /root/reference/mm/trainer_utils.py (same names, argument meaning and results), with the
[N,V] arithmetic routed to the HIP loss kernels (csrc/loss.hip) instead of ATen."""
from __future__ import annotations

import torch

from . import hip
from .llava import row_loss_sum, sequence_ce_sums

LOSSES_WITH_TEACHER = ("DPO", "SCRUB", "RMU", "LLMU", "SKU", "NPO")  # mm/trainer_utils.py:5


def loss_needs_teacher_reference(loss_type: str) -> bool:
    """Bit-for-bit behaviour of mm/trainer_utils.py:8-9 — case-sensitive "KL" test (a latent defect of the
    reference: the trainer lower-cases loss_type)."""
    return "KL" in loss_type or loss_type.upper() in LOSSES_WITH_TEACHER


def loss_needs_teacher(loss_type: str) -> bool:
    """What the dispatcher actually needs: a teacher iff the branch evaluates one (either case accepted)."""
    lt = loss_type.lower()
    return "kl" in lt or lt.upper() in LOSSES_WITH_TEACHER


def has_lora_adapter(model) -> bool:  # mm/trainer_utils.py:12-13
    """peft names its adapter parameters `...lora_A...` / `...lora_B...`; LlavaHIP keeps all trainable tensors in ONE
    flat parameter, so it says so itself (`model.lora`: name -> adapter, empty for full fine-tuning)."""
    if getattr(model, "lora", None):
        return True
    return any(True for name, _ in model.named_parameters() if "lora" in name)


class _LogProbs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits2d, log):
        M, V = logits2d.shape
        out = torch.empty(M, V, device=logits2d.device, dtype=torch.float32)
        lse = torch.empty(M, device=logits2d.device, dtype=torch.float32)
        hip.logprobs_fwd(logits2d, out, lse, log)
        ctx.save_for_backward(logits2d, lse)
        ctx.log = log
        return out

    @staticmethod
    def backward(ctx, g):
        logits2d, lse = ctx.saved_tensors
        dl = torch.empty_like(logits2d)
        hip.logprobs_bwd(logits2d, lse, g.contiguous().float(), dl, ctx.log)
        return dl, None


def logits2probs(logits, log=True):
    """mm/trainer_utils.py:16-21 by name: (log-)softmax over the vocabulary, flattened to [N, V], fp32 as under the
    reference's autocast, differentiable — one HIP kernel each way (`mmu_logprobs_fwd/bwd`).  The loss branches of this
    package never call it (they read the logits once through `mmu_rowloss_*` and materialise nothing, 4 V bytes per row
    saved each time); it exists so that code written against the reference's helper keeps working."""
    V = logits.shape[-1]
    x = logits.reshape(-1, V)
    if x.is_cuda and x.dtype == torch.bfloat16 and V % 8 == 0:
        return _LogProbs.apply(x.contiguous(), bool(log))
    raise TypeError("logits2probs: expected bf16 logits on the GPU with a vocabulary that is a multiple of 8 "
                    "(the HIP path has no eager substitute)")


class DecoderLayerHandle:
    """What `model.language_model.model.layers[i]` is to the reference's forward_with_cache (mm/trainer.py:302-310): names
    decoder layer `index` of a LlavaHIP, which has no nn.Module per layer.  `LlavaHIP.decoder_layers[i]` builds one."""

    def __init__(self, index: int):
        self.index = int(index)


def forward_with_cache(model, inputs, module, no_grad=True):
    """mm/trainer_utils.py:24-45 by name: the output of `module` ([B, S, H]) during `model(**inputs)`.  For a LlavaHIP,
    `module` is a DecoderLayerHandle (or a plain layer index) and the forward sweep stops at that layer
    (`LlavaHIP.hidden_states`: the layers above cannot influence the activation or its gradient); for any other
    nn.Module the reference's forward hook is used as is."""
    if hasattr(model, "hidden_states") and isinstance(module, (int, DecoderLayerHandle)):
        layer = module if isinstance(module, int) else module.index
        kw = {k: v for k, v in inputs.items() if k != "labels"}
        if no_grad:
            with torch.no_grad():
                return model.hidden_states(layer, **kw)
        return model.hidden_states(layer, **kw)
    cache = []

    def hook(_m, _inp, output):
        cache.append(output[0] if isinstance(output, tuple) else output)
        return None

    handle = module.register_forward_hook(hook)
    try:
        if no_grad:
            with torch.no_grad():
                model(**inputs)
        else:
            model(**inputs)
    finally:
        handle.remove()
    return cache[0]


def kl_div_batchmean(input_logits, target_logits, student_is_input: bool):
    """F.kl_div(logits2probs(input), logits2probs(target), reduction="batchmean", log_target=True) as the reference
    calls it (mm/trainer.py:203-208 etc.): divides by ALL B*S rows.  Exactly one side is the student (the side
    that carries gradient); `student_is_input` says which."""
    V = input_logits.shape[-1]
    n_rows = input_logits.numel() // V
    if student_is_input:
        s = row_loss_sum(input_logits, target_logits, hip.KL_STUDENT_INPUT)
    else:
        s = row_loss_sum(target_logits, input_logits, hip.KL_TEACHER_INPUT)
    return s / n_rows


def npo_loss(student_logits, teacher_logits, beta: float):
    """(2/beta) * mean_{n,v} log(1 + (p/(p_T + 1e-8))^beta)   (mm/trainer.py:348-353)."""
    n = student_logits.numel()
    return (2.0 / beta) * row_loss_sum(student_logits, teacher_logits, hip.NPO, beta) / n


def entropy_term(student_logits):
    """torch.sum(p * log p) over every row and vocab entry (mm/trainer.py:195-196)."""
    return row_loss_sum(student_logits, None, hip.ENTROPY)


def get_batch_loss(output, labels):
    """mm/trainer_utils.py:73-81: per-sequence SUM of the shifted CE -> [B]."""
    return sequence_ce_sums(output, labels.to(output.device))


def image_token_source_rows(input_ids, n_logit_rows: int, image_token_id):
    """Index arithmetic of mm/trainer_utils.py:48-70 for the whole batch at once, on the device of `input_ids`: for every
    sequence and every kept position t < T, the position inside its L = n_logit_rows logits rows that the reference
    keeps — the first `start` rows (start = column of the sequence's first <image> id) and the last T - start rows; a
    sequence without an image keeps its last T rows.  Returns int64 [B, T]."""
    B, T = input_ids.shape
    is_img = input_ids == image_token_id
    start = torch.where(is_img.any(dim=1), is_img.to(torch.int64).argmax(dim=1), torch.zeros(B, dtype=torch.int64,
                                                                                               device=input_ids.device))
    t = torch.arange(T, device=input_ids.device, dtype=torch.int64)[None]
    return torch.where(t < start[:, None], t, n_logit_rows - T + t)


def remove_image_tokens(input_ids, logits, image_token_id):
    """mm/trainer_utils.py:48-70: align logits [B, L, V] of a model that expands ONE <image> id into its patch
    positions (transformers 4.45.1) with the un-expanded `input_ids` [B, T].  With pre-expanded ids (transformers >=
    4.47 processors, and every batch this repo builds) L == T and the function is the identity, as in the reference.
    The general case has no host loop and no device sync: the kept rows are computed for the whole batch on the
    device and copied by one row-gather kernel (bf16 device logits; other tensors — the helper's own fp32 CPU
    fixture — go through the same indices with torch indexing)."""
    B, T = input_ids.shape
    L = logits.shape[1]
    if L == T:
        return logits
    pos = image_token_source_rows(input_ids.to(logits.device), L, image_token_id)  # [B, T]
    if logits.is_cuda and logits.dtype == torch.bfloat16 and logits.shape[-1] % 8 == 0:
        V = logits.shape[-1]
        rows = (pos + torch.arange(B, device=pos.device, dtype=torch.int64)[:, None] * L).reshape(-1).contiguous()
        out = torch.empty(B, T, V, device=logits.device, dtype=logits.dtype)
        hip.gather_rows(logits.contiguous().view(B * L, V), rows, out.view(B * T, V))
        return out
    return logits[torch.arange(B, device=logits.device)[:, None], pos]
