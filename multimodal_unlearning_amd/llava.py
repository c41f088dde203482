"""LLaVA-1.5 forward + backward on the hand-written gfx950 kernels (libmmu_hip.so).

This is the `model(**inputs)` that the reference's MMTrainerForgetting.compute_loss calls
(mm/trainer.py:180-433): it returns an object with `.loss` (HF shifted CE, mean over supervised tokens)
and `.logits`, attached to autograd so `loss.backward()` fills the gradients.  PyTorch supplies device
memory, the stream and the autograd edge; every arithmetic step is a C-ABI kernel launch:

  CLIP tower (frozen): im2col -> patch GEMM -> cls/pos -> pre-LN -> [LN, qkv GEMM+bias, attention(64, full),
      out GEMM+bias+residual, LN, fc1 GEMM+bias+quick_gelu, fc2 GEMM+bias+residual] x (L-1) -> drop CLS
  projector: GEMM+bias -> GELU -> GEMM+bias;  token gather + image-feature scatter
  Llama decoder: [RMSNorm, fused qkv GEMM, RoPE in place, causal attention(128), o GEMM+residual, RMSNorm,
      fused gate|up GEMM, SwiGLU, down GEMM+residual] x L -> RMSNorm -> lm_head GEMM -> bf16 logits
  backward: explicit reverse sweep (dgrad = NN GEMM, wgrad = TN GEMM accumulating straight into the flat bf16
      gradient buffer), RMSNorm / SwiGLU / RoPE / attention backward kernels, embedding scatter-add.

Memory layout (MI355X, 288 GB): all trainable parameters live in ONE flat bf16 buffer (`train_flat`, a single
nn.Parameter; projector then decoder, layer by layer) with ONE flat bf16 gradient buffer of the same shape —
the AdamW kernel, the grad-norm kernel and the RCCL all-reduce each see a few large contiguous ranges.
Frozen CLIP weights live in `frozen_flat`.  Activations are [tokens, features] row-major bf16; q/k/v stay
fused in one [tokens, 3*H*D] buffer that the attention kernels index by head (no transposes anywhere).
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch
import torch.nn as nn

from . import hip
from .config import LlavaConfig
from .params import init_params, is_vision, param_specs

BF = torch.bfloat16


class LoraSpec:
    """peft LoraConfig subset used by the reference (mm/forget.py:126-137; config/mm/forget.yaml:3-6): rank r,
    alpha, dropout 0, bias "none".  Targets: decoder q|k|v, o, gate|up, down, lm_head (utils.py:10-19 only strips
    the top-level name "lm_head", so under transformers 4.45.1 LLaVA's `language_model.lm_head` IS adapted), the two
    projector Linears and — target "vision" — the q|k|v, out, fc1, fc2 Linears of the CLIP layers that feed the
    projector (peft wraps every nn.Linear of the model, frozen tower included; the patch conv is not a Linear).
    `targets` makes the set explicit, as SURVEY.md §4 asks."""

    def __init__(self, r: int = 8, alpha: float = 32.0,
                 targets=("qkv", "o", "gate_up", "down", "lm_head", "projector", "vision"), dropout: float = 0.0, seed: int = 0):
        self.r, self.alpha, self.targets = int(r), float(alpha), tuple(targets)
        self.scaling = self.alpha / self.r
        # peft's lora_dropout (mm/forget.py:128-136, config/mm/forget.yaml:6): nn.Dropout on the ADAPTER's input in training
        # forwards, y = W x + (alpha/r) B A drop(x); `seed` starts the stream of per-Linear mask seeds
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f"LoRA dropout {dropout}: expected 0 <= p < 1")
        self.dropout, self.seed = float(dropout), int(seed)


class _Adapter:
    """One (possibly fused) LoRA module: y = x W^T + s * (x A^T) B^T.  A32 [kl,in] and BL [out,kl] are views of the
    trainable flat buffer (kl = k*r rounded up to 32, zero padded; fused q|k|v / gate|up adapters are block-diagonal in
    BL, enforced by `mask`).  Derived, refreshed after every optimiser step: A32s = s*A32, A32T = A32^T, BLs = s*BL."""
    __slots__ = ("name", "k", "r", "kl", "A32", "BL", "gA", "gB", "mask", "A32s", "A32T", "BLs")



@dataclass
class LlavaOutput:
    """Attribute + key access like transformers' ModelOutput (`.loss`, `.logits`).  A dataclass on purpose: torch DDP looks for
    the tensors of a forward's output (to find the autograd graph when `find_unused_parameters=True`, mm/forget.py:161) in
    tensors, lists, dicts and dataclasses only — an opaque object would read as "no parameter was used"."""
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    hidden_states: Optional[torch.Tensor] = None

    def __getitem__(self, k):
        return getattr(self, k)


def _align8(n):
    return (n + 7) // 8 * 8


# --------------------------------------------------------------------------------------------------------
# host views of the small integer tensors (labels, attention masks) the host logic has to look at
# --------------------------------------------------------------------------------------------------------
def attach_host(dev_t: torch.Tensor, host_t: torch.Tensor) -> torch.Tensor:
    """Remember on the device tensor the host tensor it was copied from (collator output), with both version counters:
    an in-place edit of either side afterwards (a host loop masking extra tokens, a reused host buffer) invalidates the
    shortcut instead of silently selecting rows / dropping the key mask from stale data."""
    dev_t._mmu_host = (host_t, host_t._version, dev_t._version)
    return dev_t


def host_of(t: Optional[torch.Tensor], fetch: bool = False) -> Optional[torch.Tensor]:
    """Host copy of a [B, S] label / mask tensor: the tensor itself when it lives on the host, the tensor recorded by
    `attach_host` while neither side has been written since, else None — or, with fetch=True, a fresh device->host copy
    (ONE sync) that is recorded on the tensor object, so a batch that stays in HBM and is stepped over repeatedly
    (bench.py) pays it once."""
    if t is None:
        return None
    if t.device.type == "cpu":
        return t
    rec = getattr(t, "_mmu_host", None)
    if rec is not None:
        host, hv, dv = rec
        if host._version == hv and t._version == dv:
            return host
    if not fetch:
        return None
    host = t.detach().cpu()
    attach_host(t, host)
    return host


def cat_with_host(tensors, dim: int = 0) -> torch.Tensor:
    """torch.cat that carries the host views along (paired fusion, merged micro-batches): the concatenation is a new
    tensor object every call, so whatever the parts already know about their contents must travel with it."""
    out = torch.cat(list(tensors), dim=dim)
    if out.device.type != "cpu":
        hosts = [host_of(t, fetch=True) for t in tensors]
        attach_host(out, torch.cat(hosts, dim=dim))
    return out


# --------------------------------------------------------------------------------------------------------
# loss autograd functions (kernels in csrc/loss.hip)
# --------------------------------------------------------------------------------------------------------
class _SegCE(torch.autograd.Function):
    """HF ForCausalLMLoss per segment of rows: out[seg] = mean over valid shifted targets of CE (or the SUM when
    normalize=False, as get_batch_loss needs).  logits2d [M,V] bf16, labels [M] int64 (unshifted), S = seq len."""

    @staticmethod
    def forward(ctx, logits2d, labels, S, seg_rows, normalize, shift=True, grad_in_place=False):
        M, V = logits2d.shape
        dev = logits2d.device
        lse = torch.empty(M, device=dev, dtype=torch.float32)
        lrow = torch.empty(M, device=dev, dtype=torch.float32)
        hip.ce_fwd(logits2d, labels, lse, lrow, S, shift=shift)
        nseg = (M + seg_rows - 1) // seg_rows
        sums = torch.empty(nseg, device=dev, dtype=torch.float32)
        cnts = torch.empty(nseg, device=dev, dtype=torch.float32)
        hip.seg_reduce(lrow, labels, seg_rows, M, V, S, shift, sums, cnts)
        ctx.save_for_backward(logits2d, labels, lse, cnts)
        ctx.meta = (S, seg_rows, normalize, shift, grad_in_place)
        return sums / cnts.clamp(min=1.0) if normalize else sums

    @staticmethod
    def backward(ctx, g):
        logits2d, labels, lse, cnts = ctx.saved_tensors
        S, seg_rows, normalize, shift, grad_in_place = ctx.meta
        # the logits of a loss-only forward belong to this edge alone: their gradient overwrites them in place (the
        # kernel reads every element before the same lane writes it) instead of taking another [rows, V] buffer
        if grad_in_place:
            # the kernel writes through the C ABI, which autograd's version counters do not see: a second backward through
            # this edge (retain_graph=True) would differentiate the gradient instead of the logits
            if getattr(ctx, "consumed", False):
                raise RuntimeError("backward through a loss-only forward twice: its logits were overwritten by their "
                                   "gradient (loss_args={'loss_only_head': False} keeps them)")
            ctx.consumed = True
        dl = logits2d if grad_in_place else torch.empty_like(logits2d)
        hip.ce_bwd(logits2d, labels, lse, g.contiguous().float(), cnts, seg_rows, normalize, dl, S, shift=shift)
        return dl, None, None, None, None, None, None


class _RowLoss(torch.autograd.Function):
    """sum over rows of a dual-logit row loss (mode: hip.KL_STUDENT_INPUT / KL_TEACHER_INPUT / NPO / ENTROPY)."""

    @staticmethod
    def forward(ctx, slog, tlog, mode, beta):
        M, V = slog.shape
        stats = torch.empty(M, 4, device=slog.device, dtype=torch.float32)
        hip.rowloss_fwd(mode, slog, tlog, beta, stats)
        out = torch.empty(1, device=slog.device, dtype=torch.float32)
        # row values sit at stride 4 in `stats`; reduce them with the segment kernel on a compact copy
        vals = stats[:, 2].contiguous()
        hip.seg_reduce(vals, None, M, M, V, 1, False, out, None)
        ctx.save_for_backward(slog, tlog, stats)
        ctx.meta = (mode, beta)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        slog, tlog, stats = ctx.saved_tensors
        mode, beta = ctx.meta
        dl = torch.empty_like(slog)
        hip.rowloss_bwd(mode, slog, tlog, beta, stats, g.reshape(1).float().contiguous(), dl)
        return dl, None, None, None


def supervised_rows(labels: torch.Tensor, vocab: int, device):
    """Row selection for a loss-only forward.  HF's causal-LM loss pairs position s with labels[s + 1] and ignores a pair
    whose label is -100 (the collator masks padding, image and question tokens, mm/dataset.py:325-339), so only the
    hidden states of the supervised positions ever reach the lm_head.  Returns (rows [B * cap] int64 device: flat row
    b * S + s per kept position, -1 in the padding slots; labels_c [B * cap] int64 device: the UNshifted target of every
    slot, -100 in the padding slots; cap = slots per sequence, a multiple of 8).  The index arithmetic runs where the
    labels live: on the host for collator output (no device sync), on the device otherwise (one sync for `cap`)."""
    B, S = labels.shape
    lab = labels.to(torch.int64)
    valid = (lab[:, 1:] >= 0) & (lab[:, 1:] < vocab)                       # position s < S - 1 predicts lab[s + 1]
    cnt = valid.sum(dim=1)
    cap = max(8, (int(cnt.max()) + 7) // 8 * 8) if B else 8
    cap = min(cap, (S - 1 + 7) // 8 * 8)
    order = torch.argsort(valid.to(torch.int8), dim=1, descending=True, stable=True)   # kept positions first, in order
    if order.shape[1] < cap:
        order = torch.cat([order, order.new_zeros(B, cap - order.shape[1])], dim=1)
    order = order[:, :cap]
    live = torch.arange(cap, device=lab.device).unsqueeze(0) < cnt.unsqueeze(1)
    base = (torch.arange(B, device=lab.device) * S).unsqueeze(1)
    rows = torch.where(live, base + order, torch.full_like(order, -1))
    tgt = torch.where(live, torch.gather(lab, 1, (order + 1).clamp(max=S - 1)), torch.full_like(order, -100))
    return rows.reshape(-1).contiguous().to(device), tgt.reshape(-1).contiguous().to(device), cap


def causal_lm_loss(logits, labels):
    """`outputs.loss` of the HF model: logits [B,S,V] bf16, labels [B,S] int64 -> 0-dim fp32."""
    B, S, V = logits.shape
    return _SegCE.apply(logits.view(B * S, V), labels.reshape(-1).contiguous(), S, B * S, True)[0]


def segment_ce_losses(logits, labels, n_segments):
    """Per-split mean CE for a batch that concatenates `n_segments` equal splits along dim 0 (paired fusion)."""
    B, S, V = logits.shape
    assert B % n_segments == 0
    return _SegCE.apply(logits.view(B * S, V), labels.reshape(-1).contiguous(), S, (B // n_segments) * S, True)


def sequence_ce_sums(logits, labels):
    """mm/trainer_utils.py:73-81 get_batch_loss: per-sequence SUM of shifted CE -> [B]."""
    B, S, V = logits.shape
    return _SegCE.apply(logits.view(B * S, V), labels.reshape(-1).contiguous(), S, S, False)


def row_loss_sum(student_logits, teacher_logits, mode, beta=1.0):
    V = student_logits.shape[-1]
    s2 = student_logits.reshape(-1, V)
    t2 = None if teacher_logits is None else teacher_logits.reshape(-1, V)
    return _RowLoss.apply(s2, t2, mode, float(beta))


# --------------------------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------------------------
class LlavaHIP(nn.Module):
    def __init__(self, cfg: LlavaConfig, params: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0",
                 seed: int = 0, freeze_vision: bool = True, trainable: bool = True, init: str = "numpy",
                 lora: Optional[LoraSpec] = None):
        super().__init__()
        hip.load()  # fail loudly if the HIP library is missing
        # `freeze_vision_module: false` (config/mm/forget.yaml:57, mm/forget.py:117-118): in full fine-tuning the CLIP tower's
        # weights, LayerNorms, class / position embeddings and the patch conv train too (their tensors then live in the
        # trainable flat buffer, in front of the projector).  Under LoRA peft freezes every base weight whatever the flag says.
        self.train_vision = (not freeze_vision) and lora is None
        self.cfg = cfg
        self.config = cfg  # HF-style alias (reference reads model.config.image_token_index)
        dev = torch.device(device)
        specs = param_specs(cfg)
        self._specs = specs
        # flat layout
        self._off: Dict[str, tuple] = {}
        nv = nt = 0
        for name, shape in specs.items():
            n = _align8(math.prod(shape))
            if is_vision(name) and not self.train_vision:
                self._off[name] = ("v", nv, shape)
                nv += n
            else:
                self._off[name] = ("t", nt, shape)
                nt += n
        self.register_buffer("frozen_flat", torch.zeros(nv, device=dev, dtype=BF), persistent=True)
        self.lora_spec = lora
        self.lora: Dict[str, _Adapter] = {}
        self._base_flat = torch.zeros(nt, device=dev, dtype=BF)
        self.w: Dict[str, torch.Tensor] = OrderedDict()
        for name, (kind, off, shape) in self._off.items():
            base = self.frozen_flat if kind == "v" else self._base_flat
            self.w[name] = base[off:off + math.prod(shape)].view(shape)
        # trainable flat buffer + name -> (offset, shape) of every trainable tensor inside it
        self._toff: Dict[str, tuple] = OrderedDict()
        if lora is None:
            self.train_flat = nn.Parameter(self._base_flat, requires_grad=trainable)
            for name, (kind, off, shape) in self._off.items():
                if kind == "t":
                    self._toff[name] = (off, shape)
        else:
            self.train_flat = nn.Parameter(self._build_lora(lora, dev), requires_grad=trainable)
        self._grad_flat: Optional[torch.Tensor] = None
        self.g: Dict[str, torch.Tensor] = OrderedDict()
        self.wT: Dict[str, torch.Tensor] = {}
        self._wT_flat: Optional[torch.Tensor] = None
        self.transposed_wgrad = True
        self.transposed_wgrad_min_m = 1024
        self.fuse_rope = os.environ.get("MMU_NO_ROPE_FUSION", "0") != "1"  # A/B switch
        self.fuse_dqkv_t = os.environ.get("MMU_NO_DQKV_T", "0") != "1"     # A/B switch: dqkv^T written by attention backward
        self.fuse_swiglu = os.environ.get("MMU_NO_SWIGLU_FUSION", "0") != "1"  # A/B switch: SwiGLU fwd / bwd in the GEMM epilogues
        self.attn_out_t = os.environ.get("MMU_NO_ATTN_OT", "0") != "1"        # A/B switch: a^T written by the attention forward
        # activation recompute = the reference's `model.gradient_checkpointing_enable()` (mm/forget.py:123-124): the forward
        # keeps one [tokens, d] tensor per decoder layer instead of ~92-114 KB per token and layer, the backward pays one more
        # forward of every layer (+33 % FLOPs).  Buys micro-batch where HBM is the limit (13B + teacher + fp32 moments).
        self.recompute_activations = False
        # keep act^T of every layer from forward to backward (+ layers*M*I*2 B, 23 GB at 7B / 30 sequences) instead of
        # recomputing it from gate|up in backward; the owner of the step turns it on when HBM allows (bench.py does)
        self.keep_act_t = False
        self._fwd_U = None  # LoRA: x A^T of every adapted Linear of the forward in flight (-> saved["U"])
        self._bwd_U = None
        self._drop_seeds = None  # lora_dropout: wname -> mask seed of the forward / backward in flight (-> saved["drop"])
        self._drop_ctr = 0
        if torch.device(device).type == "cuda":
            hip.ensure_gemm_workspace(torch.device(device))  # split-K tail partials (128 MiB, once per process)
        self.lora_splitk_min_m = 2048
        if params is not None:
            self.load_params(params)
        elif init == "device":
            self.init_random_(seed)
        else:
            self.load_params(init_params(cfg, seed=seed))
        # rope tables + scratch
        t = cfg.text
        self._cos = torch.empty(t.max_position, t.head_dim // 2, device=dev, dtype=BF)
        self._sin = torch.empty_like(self._cos)
        hip.rope_table(self._cos, self._sin, t.max_position, t.head_dim, t.rope_theta)
        self._embed_scratch = None
        self._embed_flag = None
        # hook: called as fn(lo, hi) with the [lo,hi) element range of the flat gradient that is final for this
        # backward (used by the DP engine to start the RCCL all-reduce of a bucket while backward continues)
        self.grad_ready_hook: Optional[Callable[[int, int], None]] = None
        # Autograd edges (forwards with a saved graph) whose backward sweep has not run yet.  A loss with several student
        # forwards (kl, scrub, llmu, dpo, un-fused grad_diff ...) runs several sweeps in one loss.backward(): the
        # gradient buckets may only be handed to grad_ready_hook by the LAST of them (an earlier sweep's all-reduce would
        # be summed over the ranks again after the later sweeps added to it).  The owner of the step resets the counter
        # per micro-batch (a forward whose output is never used, as in the reference's dpo branch, never sweeps; the
        # engine then reduces what no hook covered).
        self.live_sweeps = 0
        self.capture_layer: Optional[int] = None
        self._derived_at = -1
        self.position_ids_mode = "arange"  # or "from_mask" (transformers 4.45.1 left-padding positions); see _positions
        # > 1 while the plug-in evaluates several gradient-accumulation micro-batches as ONE concatenated batch
        # (MMTrainerForgetting.compute_loss(..., _merge=g)): `.loss` is then the mean over the g equal row segments of each
        # segment's own mean CE — what the host loop would have averaged — instead of one mean over all tokens
        self.loss_segments = 1
        self.param_events = None      # see _wait_params
        self.autograd_grads = False   # see enable_autograd_gradients
        self.ddp_compatible = False
        # torch DDP broadcasts module buffers from rank 0 at every forward; the frozen CLIP weights never change
        self._ddp_params_and_buffers_to_ignore = ["frozen_flat"]
        if self.lora:
            self.init_lora_(seed)

    # ------------------------------------------------------------------------------------------ params
    @property
    def device(self):
        return self.train_flat.device

    # ------------------------------------------------------------------------------------------- LoRA
    def _lora_targets(self, spec: LoraSpec):
        t = self.cfg.text
        out = []  # (weight name, k fused sub-modules)
        if "vision" in spec.targets:
            for i in range(self.cfg.vision_layers_used):
                p = f"vision.layers.{i}."
                out += [(p + "qkv.weight", 3), (p + "out.weight", 1), (p + "fc1.weight", 1), (p + "fc2.weight", 1)]
        if "projector" in spec.targets:
            out += [("projector.linear_1.weight", 1), ("projector.linear_2.weight", 1)]
        for i in range(t.num_layers):
            p = f"text.layers.{i}."
            for key, k in (("qkv", 3), ("o", 1), ("gate_up", 2), ("down", 1)):
                if key in spec.targets:
                    out.append((p + key + ".weight", k))
        if "lm_head" in spec.targets:
            out.append(("lm_head.weight", 1))
        return out

    def _build_lora(self, spec: LoraSpec, dev):
        off = 0
        plan = []
        for wname, k in self._lora_targets(spec):
            n_out, n_in = self._specs[wname]
            # padded inner width of the (possibly fused) adapter: 32 or 64 ride in the GEMM epilogue; wider ones (e.g. r = 32
            # on the fused q|k|v: 96) are applied as a second, accumulating GEMM — which is also how peft adds its branch
            kl = (k * spec.r + 31) // 32 * 32
            base = wname[: -len(".weight")]
            self._toff[base + ".lora_A"] = (off, (kl, n_in))
            off += kl * n_in
            self._toff[base + ".lora_B"] = (off, (n_out, kl))
            off += n_out * kl
            plan.append((wname, k, kl, n_out, n_in))
        flat = torch.zeros(off, device=dev, dtype=BF)
        for wname, k, kl, n_out, n_in in plan:
            base = wname[: -len(".weight")]
            ad = _Adapter()
            ad.name, ad.k, ad.r, ad.kl = wname, k, spec.r, kl
            oa, sa = self._toff[base + ".lora_A"]
            ob, sb = self._toff[base + ".lora_B"]
            ad.A32 = flat[oa:oa + kl * n_in].view(sa)
            ad.BL = flat[ob:ob + n_out * kl].view(sb)
            mask = torch.zeros(n_out, kl, device=dev, dtype=BF)
            seg = n_out // k
            for j in range(k):
                mask[j * seg:(j + 1) * seg, j * spec.r:(j + 1) * spec.r] = 1
            ad.mask = mask
            ad.A32s = torch.zeros(kl, n_in, device=dev, dtype=BF)
            ad.A32T = torch.zeros(n_in, kl, device=dev, dtype=BF)
            ad.BLs = torch.zeros(n_out, kl, device=dev, dtype=BF)
            ad.gA = ad.gB = None
            self.lora[wname] = ad
        return flat

    def init_lora_(self, seed: int = 0, b_std: float = 0.0):
        """peft init: A ~ kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in)) on the real rows, B = 0 (b_std > 0
        draws B ~ N(0, b_std^2) inside its blocks: tests need a non-trivial adapter)."""
        gen = torch.Generator(device=self.device).manual_seed(seed + 1000)
        with torch.no_grad():
            for ad in self.lora.values():
                kl, n_in = ad.A32.shape
                bound = 1.0 / math.sqrt(n_in)
                ad.A32.zero_()
                ad.A32[: ad.k * ad.r].uniform_(-bound, bound, generator=gen)
                ad.BL.zero_()
                if b_std > 0:
                    ad.BL.normal_(0.0, b_std, generator=gen)
                    ad.BL.mul_(ad.mask)
        self.refresh_derived()

    def refresh_derived(self):
        """Recompute what depends on the trainable weights: LoRA's scaled / transposed adapter copies, or the
        resident W^T copies in full fine-tuning.  The owner of the optimiser step calls this after every update."""
        if self.lora:
            s = self.lora_spec.scaling
            with torch.no_grad():
                for ad in self.lora.values():
                    ad.A32s.copy_(ad.A32 * s)
                    ad.A32T.copy_(ad.A32.t())
                    ad.BLs.copy_(ad.BL * s)
        elif self.wT:
            self.refresh_transposed_weights()
        self._derived_at = self.train_flat._version

    def _sync_derived(self):
        """A foreign optimiser (transformers.Trainer's, when the plug-in is hosted by it) updates `train_flat` in place
        and knows nothing of the derived copies: every forward checks the parameter's version counter and refreshes them
        when it has moved.  (ForgetEngine's fused AdamW writes through the C ABI, which the counter does not see, and
        calls refresh_derived itself.)"""
        if (self.lora or self.wT) and self._derived_at != self.train_flat._version:
            self.refresh_derived()

    def lora_state(self) -> "OrderedDict[str, torch.Tensor]":
        """peft-shaped adapter tensors: <module>.lora_A [k*r, in] / lora_B [out, r] (compact, per fused module)."""
        out = OrderedDict()
        for wname, ad in self.lora.items():
            base = wname[: -len(".weight")]
            out[base + ".lora_A"] = ad.A32[: ad.k * ad.r].detach().float().cpu()
            seg = ad.BL.shape[0] // ad.k
            out[base + ".lora_B"] = torch.cat([ad.BL[j * seg:(j + 1) * seg, j * ad.r:(j + 1) * ad.r]
                                               for j in range(ad.k)]).detach().float().cpu()
        return out

    def save_pretrained(self, save_dir: str, extra=None, **kw):
        """`model.save_pretrained(cfg.save_dir)` as the reference calls it after training (mm/forget.py:193-195): HF-layout
        safetensors + config.json, and for a LoRA model the peft adapter files next to the base weights
        (checkpoint.save_pretrained; `extra`: what load_pretrained(..., return_extra=True) set aside)."""
        from .checkpoint import save_pretrained

        if self.lora:
            kw.setdefault("lora_state", self.lora_state())
            kw.setdefault("lora_r", self.lora_spec.r)
            kw.setdefault("lora_alpha", self.lora_spec.alpha)
            kw.setdefault("lora_dropout", self.lora_spec.dropout)
        return save_pretrained(self.cfg, self.w, save_dir, extra=extra, **kw)

    def load_lora_state_(self, state):
        """Inverse of lora_state(): compact peft-shaped tensors -> the padded, block-masked adapter buffers."""
        with torch.no_grad():
            for wname, ad in self.lora.items():
                base = wname[: -len(".weight")]
                A, Bm = state[base + ".lora_A"], state[base + ".lora_B"]
                if tuple(A.shape) != (ad.k * ad.r, ad.A32.shape[1]) or tuple(Bm.shape) != (ad.BL.shape[0], ad.r):
                    raise ValueError(f"adapter shape mismatch for {base}: {tuple(A.shape)} / {tuple(Bm.shape)}")
                ad.A32.zero_()
                ad.A32[: ad.k * ad.r].copy_(A.to(self.device, BF))
                ad.BL.zero_()
                seg = ad.BL.shape[0] // ad.k
                for j in range(ad.k):
                    ad.BL[j * seg:(j + 1) * seg, j * ad.r:(j + 1) * ad.r].copy_(Bm[j * seg:(j + 1) * seg].to(self.device, BF))
        self.refresh_derived()

    def merge_lora_(self):
        """peft `merge_and_unload` (mm/forget.py:111-113) on the device: W += (alpha/r) B A for every adapted Linear as
        one accumulate-epilogue GEMM each (fp32 accumulation, ONE bf16 rounding of W + delta; peft rounds the delta to
        bf16 first — peft is absent here, so this step is "parity unpinned" against peft itself and pinned against the
        formula).  The adapters are then reset to the identity (B = 0) so the model keeps computing the same function."""
        if not self.lora:
            raise ValueError("merge_lora_ needs a model built with lora=LoraSpec(...)")
        self.refresh_derived()
        with torch.no_grad():
            for wname, ad in self.lora.items():
                hip.gemm(hip.NN, ad.BLs, ad.A32, self.w[wname], accumulate=True)  # [out,kl] x [kl,in]; off-block BLs = 0
                ad.BL.zero_()
        self.refresh_derived()
        if self.wT:
            self.refresh_transposed_weights()

    def _adapter_input(self, wname, x):
        """peft's lora_dropout: the adapter branch of an adapted Linear sees dropout(x) in TRAINING forwards (those that
        build a graph, in train() mode); the base branch sees x.  The mask is a hash of a per-(forward, Linear) seed kept in
        the forward's saved state, so the backward sweep — and a gradient-checkpointing recompute — regenerate it."""
        seeds = self._drop_seeds
        if seeds is None:
            return x
        seed = seeds.get(wname)
        if seed is None:
            self._drop_ctr += 1
            seed = seeds[wname] = (self.lora_spec.seed * 0x9E3779B97F4A7C15 + self._drop_ctr * 0xD1B54A32D192ED03) & (2 ** 64 - 1)
        return hip.dropout(x.contiguous(), torch.empty_like(x, memory_format=torch.contiguous_format), self.lora_spec.dropout, seed)

    def _lin_fwd(self, wname, x, out, bias=None, resid=None):
        """out = x W^T (+ bias) (+ resid), plus the fused LoRA term when `wname` carries an adapter."""
        ad = self.lora.get(wname)
        if ad is None:
            return hip.gemm(hip.NT, x, self.w[wname], out, bias=bias, resid=resid)
        U = torch.empty(x.shape[0], ad.kl, device=x.device, dtype=BF)
        hip.gemm(hip.NT, self._adapter_input(wname, x), ad.A32s, U)  # s * drop(x) A^T
        if self._fwd_U is not None:
            self._fwd_U[wname] = U  # [M,32]: kept for the adapter wgrad dB = s dy^T (x A^T) instead of re-reading x
        if ad.kl > 64:  # wider than the fused epilogue takes: base GEMM, then out += U B^T
            hip.gemm(hip.NT, x, self.w[wname], out, bias=bias, resid=resid)
            return hip.gemm(hip.NT, U, ad.BL, out, accumulate=True)
        return hip.gemm(hip.NT, x, self.w[wname], out, bias=bias, resid=resid, lora_u=U, lora_b=ad.BL)

    def _qkv_fwd(self, wname, x, qkv, pos, H, D, dt):
        """Fused q|k|v projection + RoPE on the q, k thirds: in the GEMM epilogue when the 256^2 kernel runs the shape
        (head_dim 128), else GEMM followed by the in-place RoPE kernel."""
        ad = self.lora.get(wname)
        U = None
        if ad is not None:
            U = torch.empty(x.shape[0], ad.kl, device=x.device, dtype=BF)
            hip.gemm(hip.NT, self._adapter_input(wname, x), ad.A32s, U)
            if self._fwd_U is not None:
                self._fwd_U[wname] = U
        wide = ad is not None and ad.kl > 64   # adapter too wide for the fused epilogue: added by a second GEMM, RoPE after it
        if D == 128 and self.fuse_rope and not wide and hip.gemm_rope(x, self.w[wname], qkv, pos, self._cos, self._sin, 2 * dt,
                                                                    lora_u=U, lora_b=None if ad is None else ad.BL):
            return
        if wide:
            hip.gemm(hip.NT, x, self.w[wname], qkv)
            hip.gemm(hip.NT, U, ad.BL, qkv, accumulate=True)
        else:
            hip.gemm(hip.NT, x, self.w[wname], qkv, lora_u=U, lora_b=None if ad is None else ad.BL)
        hip.rope(qkv, pos, self._cos, self._sin, H, H, D, dt)

    def _lora_scratch(self, n):
        if getattr(self, "_lora_scr", None) is None or self._lora_scr.numel() < n:
            self._lora_scr = torch.zeros(n, device=self.device, dtype=torch.float32)
        return self._lora_scr

    def _lin_bwd(self, wname, dy, x, dx):
        """Backward of out = x W^T (+LoRA): accumulates the weight (or adapter) gradients, writes dx if given."""
        ad = self.lora.get(wname)
        if ad is None:
            if not self.lora:  # full fine-tuning: the base weight is trainable
                self._wgrad(dy, x, wname)
            if dx is not None:
                self._dgrad(dy, wname, dx)
            return
        M = dy.shape[0]
        Vs = torch.empty(M, ad.kl, device=dy.device, dtype=BF)
        hip.gemm(hip.NN, dy, ad.BLs, Vs)  # s * dy B
        seed = self._drop_seeds.get(wname) if self._drop_seeds is not None else None
        if dx is not None:
            wT = wname in self.wT
            if seed is None and ad.kl > 64:  # wide adapter: dx = dy W, then dx += (s dy B) A
                hip.gemm(hip.NT if wT else hip.NN, dy, self.wT[wname] if wT else self.w[wname], dx)
                hip.gemm(hip.NT, Vs, ad.A32T, dx, accumulate=True)
            elif seed is None:   # the adapter's term s dy B A rides in the dgrad GEMM's epilogue
                hip.gemm(hip.NT if wT else hip.NN, dy, self.wT[wname] if wT else self.w[wname], dx, lora_u=Vs, lora_b=ad.A32T)
            else:              # lora_dropout: that term passes the forward's mask first, so it is formed on its own
                hip.gemm(hip.NT if wT else hip.NN, dy, self.wT[wname] if wT else self.w[wname], dx)
                dxa = torch.empty_like(dx)
                hip.gemm(hip.NT, Vs, ad.A32T, dxa)
                hip.dropout(dxa, dxa, self.lora_spec.dropout, seed)
                hip.add(dx, dxa, dx)
                del dxa
        if seed is not None:   # the adapter saw drop(x): its A gradient is taken against the same masked input
            x = hip.dropout(x.contiguous(), torch.empty_like(x, memory_format=torch.contiguous_format), self.lora_spec.dropout, seed)
        base = wname[: -len(".weight")]
        gA, gB = self.g[base + ".lora_A"], self.g[base + ".lora_B"]
        U = self._bwd_U.pop(wname, None) if self._bwd_U is not None else None
        if U is None:
            U = torch.empty(M, ad.kl, device=dy.device, dtype=BF)
            hip.gemm(hip.NT, x, ad.A32s, U)
        if M >= self.lora_splitk_min_m:
            # skinny outputs reduced over thousands of tokens: split-K over the tokens with fp32 accumulation
            scr = self._lora_scratch(max(gA.numel(), gB.numel()))
            hip.gemm_splitk_accumulate(hip.TN, Vs, x, scr[: gA.numel()].view(gA.shape), gA, None,
                                       splits=max(1, min(64, M // 2048 * 2)))       # dA = s (dy B)^T x
            hip.gemm_splitk_accumulate(hip.TN, dy, U, scr[: gB.numel()].view(gB.shape), gB, ad.mask,
                                       splits=max(1, min(16, M // 4096)))           # dB = s dy^T (x A^T), block mask
        else:
            hip.gemm(hip.TN, Vs, x, gA, accumulate=True)
            hip.gemm(hip.TN, dy, U, gB, accumulate=True)  # off-block entries are not parameters
            gB.mul_(ad.mask)

    def init_random_(self, seed: int = 0, std: float = 0.02):
        """Random-init weights of the architecture, generated ON DEVICE (7B/13B: no host copy): N(0, std^2) for
        matrices / embeddings / biases, 1 for every norm weight (SURVEY.md §8d synthetic weights)."""
        gen = torch.Generator(device=self.device).manual_seed(seed)
        with torch.no_grad():
            for flat in (self.frozen_flat, self._base_flat):
                if flat.numel():
                    flat.normal_(0.0, std, generator=gen)
            for name, t in self.w.items():
                if name.endswith(("norm.weight", "ln1.weight", "ln2.weight", "pre_ln.weight")):
                    t.fill_(1.0)
            self.w["vision.patch_embed.weight"][:, self.cfg.vision.patch_k:] = 0

    def load_params(self, params: Dict[str, torch.Tensor]):
        with torch.no_grad():
            for name in self._specs:
                self.w[name].copy_(params[name].to(self.device, dtype=BF))

    def export_params(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, v.detach().float().cpu()) for k, v in self.w.items())

    def trainable_names(self):
        return list(self._toff)

    def range_of(self, name):
        off, shape = self._toff[name]
        return off, off + _align8(math.prod(shape))

    def _span(self, prefix: str):
        """[lo, hi) of the flat trainable/gradient buffer covered by the tensors whose name starts with `prefix`."""
        r = [self.range_of(n) for n in self._toff if n.startswith(prefix)]
        return (min(a for a, _ in r), max(b for _, b in r)) if r else None

    def layer_grad_range(self, i: int):
        return self._span(f"text.layers.{i}.")

    @staticmethod
    def _is_above(name: str, layer: int) -> bool:
        """True for trainable tensors that sit after decoder layer `layer` in the sweep (upper layers, norm, head)."""
        if name.startswith("text.norm.") or name.startswith("lm_head."):
            return True
        if name.startswith("text.layers."):
            return int(name.split(".")[2]) > layer
        return False

    # -------------------------------------------------------------------------- transposed weight copies
    def _dgrad_weight_names(self):
        names = []
        if "vision.layers.0.qkv.weight" in self.lora or self.train_vision:  # adapted / unfrozen CLIP tower: its backward sweep needs dX = dY W too
            for i in range(self.cfg.vision_layers_used):
                p = f"vision.layers.{i}."
                names += [p + "qkv.weight", p + "out.weight", p + "fc1.weight", p + "fc2.weight"]
        for i in range(self.cfg.text.num_layers):
            p = f"text.layers.{i}."
            names += [p + "qkv.weight", p + "o.weight", p + "gate_up.weight", p + "down.weight"]
        return names + ["lm_head.weight"]

    def enable_transposed_weights(self):
        """Keep W^T of every decoder Linear + lm_head resident (+13.5 GB at 7B; 288 GB HBM) so that dgrad
        dX = dY W runs as an NT GEMM (both operands reduction-contiguous) instead of reading W reduction-major.
        The owner of the optimiser step must call refresh_transposed_weights() after every weight update
        (ForgetEngine does); a stale copy silently gives stale gradients, hence opt-in."""
        if self._wT_flat is None:
            names = [n for n in self._dgrad_weight_names() if n in self.w]
            total = sum(self.w[n].numel() for n in names)
            self._wT_flat = torch.empty(total, device=self.device, dtype=BF)
            off = 0
            for n in names:
                r, c = self.w[n].shape
                self.wT[n] = self._wT_flat[off:off + r * c].view(c, r)
                off += r * c
        self.refresh_transposed_weights()

    def refresh_transposed_weights(self, prefix: Optional[str] = None):
        """W^T copies from W; `prefix`: only the weights whose name starts with it (the pipelined optimiser refreshes a decoder
        layer's copies right behind that layer's AdamW launch)."""
        for n, t in self.wT.items():
            if prefix is None or n.startswith(prefix):
                hip.transpose(self.w[n], t)

    # ---- parameters updated asynchronously (ForgetEngine(overlap_optimizer=True)) ---------------------------------------------
    # The engine may run AdamW and the W^T refresh of step k on a side stream, chunk by chunk in forward order, while the
    # forward of step k + 1 already runs: `param_events` then maps a chunk ("pre" = everything in front of decoder layer 0,
    # i = decoder layer i, "post" = final norm + lm_head) to the event recorded behind that chunk's update, and every forward
    # waits for a chunk's event right before it first reads the chunk.
    def _wait_params(self, key):
        ev = self.param_events.get(key) if self.param_events else None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def param_chunks(self):
        """[(key, lo, hi)] in flat order: the ranges of `train_flat` a forward touches one after the other."""
        L = self.cfg.text.num_layers
        first = self.layer_grad_range(0)
        last = self.layer_grad_range(L - 1)
        out = [("pre", 0, first[0])] if first[0] > 0 else []
        out += [(i, *self.layer_grad_range(i)) for i in range(L)]
        n = self.train_flat.numel()
        if last[1] < n:
            out.append(("post", last[1], n))
        return out

    def _use_t(self, M):
        return self.transposed_wgrad and not self.lora and M % 8 == 0 and M >= self.transposed_wgrad_min_m

    def _wgrad_t(self, dy, dyT, xT, gname):
        """g[gname] += dy^T x given x^T (and optionally dy^T) already in the NT layout."""
        if dyT is None:
            dyT = torch.empty(dy.shape[1], dy.shape[0], device=dy.device, dtype=BF)
            hip.transpose(dy, dyT)
        return hip.gemm(hip.NT, dyT, xT, self.g[gname], accumulate=True)

    def _wgrad(self, dy, x, gname):
        """g[gname] += dy^T x  (dW of y = x W^T; reduction over tokens).  Token-major operands are reduction-MAJOR
        for this product; measured on MI355X the LDS-DMA GEMM streams reduction-contiguous operands from HBM ~35 %
        faster, so both operands are transposed first ([M,N] -> [N,M], HBM-bound, ~4.5 TB/s) and the NT kernel is
        used: net +10-15 % on every wgrad (tools/wgrad_exp.py)."""
        g = self.g[gname]
        M = dy.shape[0]
        if self.transposed_wgrad and M % 8 == 0 and M >= self.transposed_wgrad_min_m and dy.is_contiguous() and x.is_contiguous():
            dyT = torch.empty(dy.shape[1], M, device=dy.device, dtype=BF)
            xT = torch.empty(x.shape[1], M, device=x.device, dtype=BF)
            hip.transpose(dy, dyT)
            hip.transpose(x, xT)
            return hip.gemm(hip.NT, dyT, xT, g, accumulate=True)
        return hip.gemm(hip.TN, dy, x, g, accumulate=True)

    def _dgrad(self, dy, name, out):
        """out = dy @ W[name]  (dX of y = x W^T)."""
        if name in self.wT:
            return hip.gemm(hip.NT, dy, self.wT[name], out)
        return hip.gemm(hip.NN, dy, self.w[name], out)

    def enable_autograd_gradients(self):
        """Hand the gradient to autograd instead of writing `train_flat.grad` by hand — what a host that listens on the
        parameter's AccumulateGrad node needs, i.e. torch DDP under transformers.Trainer at world > 1 (the reference's
        arrangement: mm/trainer.py:112 under `accelerate launch`, mm/forget.py:161 ddp_find_unused_parameters=True).  The
        backward sweeps of a micro-batch still accumulate into the private flat buffer; the LAST sweep returns that buffer as
        the gradient of `train_flat` (earlier sweeps return None = zero), autograd adds it into `.grad`, DDP's reducer sees
        the parameter become ready and all-reduces its buckets.  The buffer is cleared by the first forward of the next
        micro-batch.  Costs one more parameter-sized tensor (`.grad`) and one add pass per micro-batch."""
        self.autograd_grads = True
        self.ddp_compatible = True
        if self.train_flat.grad is not None and self._grad_flat is not None and \
                self.train_flat.grad.data_ptr() == self._grad_flat.data_ptr():
            self.train_flat.grad = None
        return self

    def train(self, mode: bool = True):
        # a host loop calls model.train() at the start of every training step (transformers.Trainer.training_step): forwards
        # whose graph was dropped without a backward (the dead forwards of the reference's dpo branch) must not keep the
        # sweep counter above zero, or no later sweep would count as the last one
        self.live_sweeps = 0
        return super().train(mode)

    def ensure_grad(self):
        """Attach the persistent flat gradient buffer as train_flat.grad (zeroed when (re)attached).  In autograd-gradient
        mode (`enable_autograd_gradients`) the buffer stays private: `.grad` belongs to autograd / the host."""
        if self._grad_flat is None:
            self._grad_flat = torch.zeros_like(self.train_flat.data)
        if self.autograd_grads:
            if not self.g:
                for name, (off, shape) in self._toff.items():
                    self.g[name] = self._grad_flat[off:off + math.prod(shape)].view(shape)
            return self._grad_flat
        if self.train_flat.grad is None or self.train_flat.grad.data_ptr() != self._grad_flat.data_ptr():
            self._grad_flat.zero_()
            self.train_flat.grad = self._grad_flat
        if not self.g:
            for name, (off, shape) in self._toff.items():
                self.g[name] = self._grad_flat[off:off + math.prod(shape)].view(shape)
        return self._grad_flat

    def gradient_checkpointing_enable(self, *a, **kw):
        """transformers' name for it (mm/forget.py:124)."""
        self.recompute_activations = True

    def gradient_checkpointing_disable(self):
        self.recompute_activations = False

    def activation_bytes_per_token(self) -> int:
        """Upper estimate of the HBM one sequence position keeps alive through a training forward + backward (DESIGN.md
        §3): per decoder layer the saved h, a, h2 (2d each), q|k|v (6d), gate|up (4I) and, with keep_act_t, act^T (2I);
        the backward's working set (dgu, dgu^T, dact, dqkv, dqkv^T, y^T, dy, dh, dh2); student + teacher logits, their
        gradient and row statistics.  The step loop sizes merged micro-batches with it (ForgetEngine._groups)."""
        t = self.cfg.text
        d, I, V, L = t.hidden_size, t.intermediate_size, t.vocab_size, t.num_layers
        per_layer = 12 * d + 4 * I + (2 * I + 2 * d if self.keep_act_t else 0)   # keep_act_t: act^T and a^T
        if self.lora:
            per_layer += 2 * I + 2 * d  # act and the x A^T products kept for the adapter wgrads
        work = 12 * I + 24 * d
        if self.recompute_activations:  # one layer's tensors live at a time, the rest of the depth keeps its input only
            work, per_layer = work + per_layer, 2 * d
        return L * per_layer + work + 8 * V

    # ----------------------------------------------------------------------------------------- forward
    # model(**split, loss_only=True) is understood (the plug-in asks before passing the keyword to a foreign model)
    supports_loss_only = True

    def ce_losses(self, input_ids, attention_mask=None, pixel_values=None, labels=None, position_ids=None,
                  n_segments: int = 1):
        """HF's causal-LM loss of the batch — per segment when the batch concatenates `n_segments` equal splits (paired
        forget / retain fusion) — WITHOUT the logits of positions that carry no loss: the final norm, the lm_head GEMM,
        the CE kernels and their backward run on the supervised rows only (`supervised_rows`).  Same value and gradients
        as `forward(...).loss` up to summation order (tests/test_model_gpu.py::test_loss_only_matches_full_logits); at
        the headline workload 256 of 1088 positions per sequence are supervised: -76 % of the lm_head's three GEMMs."""
        B = input_ids.shape[0]
        assert B % n_segments == 0
        lc, tgt, cap = self.compact_logits(input_ids, attention_mask, pixel_values, labels, position_ids)
        return _SegCE.apply(lc, tgt, cap, (B // n_segments) * cap, True, False, True)

    def compact_logits(self, input_ids, attention_mask=None, pixel_values=None, labels=None, position_ids=None):
        """Logits of the supervised positions only: (logits [B * cap, V] bf16 attached to autograd, targets [B * cap] int64
        UNshifted — slot r of a sequence holds the position whose next label is targets[r], -100 in padding slots —, cap =
        slots per sequence).  The final norm and the lm_head run on those rows alone (`supervised_rows`)."""
        dev = self.device
        ids = input_ids.to(dev, dtype=torch.int64).contiguous()
        B, S = ids.shape
        # the row selection is index arithmetic on the HOST view of the labels (collator output carries one; a
        # device-resident tensor fetches it once and keeps it, see host_of); the result is remembered on the tensor
        # object together with its version, so stepping over the same batch again costs nothing
        rec = getattr(labels, "_mmu_rows", None)
        hit = rec[1] if (rec is not None and rec[0] == labels._version) else None
        if hit is None:
            hit = supervised_rows(host_of(labels, fetch=True), self.cfg.text.vocab_size, dev)
            labels._mmu_rows = (labels._version, hit)
        rows, tgt, cap = hit
        kmask = self._kmask(attention_mask)
        pos = self._positions(position_ids, attention_mask, B, S)
        pix = None
        if pixel_values is not None and pixel_values.numel() > 0:
            pix = pixel_values.to(dev, dtype=torch.float32).contiguous()
        if torch.is_grad_enabled() and self.train_flat.requires_grad:
            lc = _LlavaFn.apply(self.train_flat, self, ids, kmask, pix, pos, None, rows)
        else:
            with torch.no_grad():
                lc, _ = self._forward_impl(ids, kmask, pix, pos, save=False, rows=rows)
        return lc, tgt, cap

    def _kmask(self, attention_mask):
        """int32 key mask for the attention kernels, or None when nothing is padded (the kernels then skip the per-block
        mask staging and validity test altogether: forward 565 -> 502 us standalone at the step's shape, the dQ kernel
        likewise).  Host-resident masks (collator output) are inspected on the host; a device-resident mask is inspected
        ONCE per tensor object and version (one sync, as HF's flash-attention path does for every call) and remembered,
        so batches that stay in HBM and are stepped over repeatedly pay nothing afterwards."""
        if attention_mask is None:
            return None
        if attention_mask.device.type != "cpu" and os.environ.get("MMU_NO_MASK_INSPECT", "0") == "1":
            return attention_mask.to(dtype=torch.int32).contiguous()  # A/B switch: every device-resident mask goes to the kernels
        full = bool(host_of(attention_mask, fetch=True).all())
        return None if full else attention_mask.to(self.device, dtype=torch.int32).contiguous()

    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, labels=None, position_ids=None,
                loss_only: bool = False, ce_segments: Optional[int] = None, hidden_layer: Optional[int] = None, **_unused):
        """`ce_segments=n` (with labels, loss_only): `.loss` is the VECTOR of the n per-segment mean CEs of `ce_losses`;
        `hidden_layer=i`: `.hidden_states` is the output of decoder layer i (`hidden_states()`), nothing else is computed.
        Both exist so that every way the plug-in uses the model goes through `forward` — the one entry point a wrapper such
        as torch DDP intercepts (mm/forget.py:161)."""
        if hidden_layer is not None:
            return LlavaOutput(hidden_states=self.hidden_states(hidden_layer, input_ids, attention_mask, pixel_values, position_ids))
        if ce_segments is not None:
            return LlavaOutput(loss=self.ce_losses(input_ids, attention_mask, pixel_values, labels, position_ids,
                                                   n_segments=int(ce_segments)))
        g = self.loss_segments
        if loss_only and labels is not None and self.capture_layer is None:
            ce = self.ce_losses(input_ids, attention_mask, pixel_values, labels, position_ids, n_segments=g)
            return LlavaOutput(loss=ce[0] if g == 1 else ce.mean(), logits=None)
        logits = self.logits(input_ids, attention_mask, pixel_values, position_ids)
        loss = None
        if labels is not None:
            if g == 1:
                loss = causal_lm_loss(logits, labels.to(logits.device))
            else:  # merged micro-batches: `.loss` = mean over the micro-batches of each one's own token mean
                loss = segment_ce_losses(logits, labels.to(logits.device), g).mean()
        hs = None
        if self.capture_layer is not None:
            hs = self._last_capture
        return LlavaOutput(loss=loss, logits=logits, hidden_states=hs)

    def _positions(self, position_ids, attention_mask, B, S):
        """int32 [B*S] RoPE positions.  None -> `self.position_ids_mode`: "arange" (transformers >= 4.47 / 5.x:
        modeling_llama.py:386-389) or "from_mask" = cumsum(mask) - 1 with padded slots at 1, what the reference's pinned
        transformers 4.45.1 LLaVA derives for left-padded rows (`padding_side="left"`, mm/forget.py:54).  The string
        "from_mask" / "arange" may also be passed per call.  RoPE scores depend on position differences only and left
        padding shifts a row by a constant, so both give the same function up to the rounding of the bf16 cos / sin
        tables (tests/golden/trajectory_tiny.npz: HF fp32 differs by 7e-7 between the two)."""
        dev = self.device
        mode = position_ids if isinstance(position_ids, str) else (self.position_ids_mode if position_ids is None else None)
        if mode is None:
            return position_ids.to(dev, dtype=torch.int32).reshape(-1).contiguous()
        if mode == "from_mask" and attention_mask is not None:
            m = attention_mask.to(dev, dtype=torch.int64)
            return (m.cumsum(-1) - 1).masked_fill(m == 0, 1).to(torch.int32).reshape(-1).contiguous()
        if mode not in ("arange", "from_mask"):
            raise ValueError(f"position_ids mode {mode!r}: expected 'arange' or 'from_mask'")
        return torch.arange(S, device=dev, dtype=torch.int32).repeat(B)

    def logits(self, input_ids, attention_mask=None, pixel_values=None, position_ids=None):
        dev = self.device
        kmask = self._kmask(attention_mask)
        ids = input_ids.to(dev, dtype=torch.int64).contiguous()
        B, S = ids.shape
        pos = self._positions(position_ids, attention_mask, B, S)
        pix = None
        if pixel_values is not None and pixel_values.numel() > 0:
            pix = pixel_values.to(dev, dtype=torch.float32).contiguous()
        need_grad = torch.is_grad_enabled() and self.train_flat.requires_grad
        if need_grad:
            return _LlavaFn.apply(self.train_flat, self, ids, kmask, pix, pos)
        with torch.no_grad():
            logits, _ = self._forward_impl(ids, kmask, pix, pos, save=False)
        return logits

    # CLIP tower + projector -> image features [n_img*np, dt]; returns tensors needed by projector backward
    def _vision_forward(self, pix, save: bool = False):
        cfg, v, w = self.cfg, self.cfg.vision, self.w
        # the sweep that keeps what a backward through the tower needs: adapted (LoRA) tower, or unfrozen tower
        vlora = "vision.layers.0.qkv.weight" in self.lora or self.train_vision
        dev = self.device
        n = pix.shape[0]
        npatch, T, dv = v.num_patches, v.num_patches + 1, v.hidden_size
        col = torch.empty(n * npatch, v.patch_k_padded, device=dev, dtype=BF)
        hip.im2col_patch(pix, col, v.patch_size)
        patch = torch.empty(n * npatch, dv, device=dev, dtype=BF)
        hip.gemm(hip.NT, col, w["vision.patch_embed.weight"], patch)
        emb = torch.empty(n * T, dv, device=dev, dtype=BF)
        hip.clip_embed(patch, w["vision.class_embedding"], w["vision.position_embedding"], emb, n, T, dv)
        h = torch.empty_like(emb)
        hip.layernorm_fwd(emb, w["vision.pre_ln.weight"], w["vision.pre_ln.bias"], h, v.layer_norm_eps)
        y = torch.empty_like(h)
        qkv = torch.empty(n * T, 3 * dv, device=dev, dtype=BF)
        a = torch.empty_like(h)
        z = torch.empty(n * T, v.intermediate_size, device=dev, dtype=BF)
        scale = v.head_dim ** -0.5
        clip_saved = [] if (vlora and save) else None
        for i in range(cfg.vision_layers_used if vlora else 0):
            # adapted tower: every Linear goes through _lin_fwd (fused low-rank term); the tensors the backward sweep
            # needs are kept (35 MB each at 30 images: a few GB over 23 layers)
            p = f"vision.layers.{i}."
            y1 = torch.empty_like(h)
            hip.layernorm_fwd(h, w[p + "ln1.weight"], w[p + "ln1.bias"], y1, v.layer_norm_eps)
            qkv = torch.empty(n * T, 3 * dv, device=dev, dtype=BF)
            self._lin_fwd(p + "qkv.weight", y1, qkv, bias=w[p + "qkv.bias"])
            a = torch.empty_like(h)
            lse = torch.empty(n, v.num_heads, T, device=dev, dtype=torch.float32) if save else None
            hip.attn_fwd(qkv[:, :dv], qkv[:, dv:2 * dv], qkv[:, 2 * dv:], a, lse, None, n, T, v.num_heads,
                         v.head_dim, scale, False)
            h2 = torch.empty_like(h)
            self._lin_fwd(p + "out.weight", a, h2, bias=w[p + "out.bias"], resid=h)
            y2 = torch.empty_like(h)
            hip.layernorm_fwd(h2, w[p + "ln2.weight"], w[p + "ln2.bias"], y2, v.layer_norm_eps)
            z = torch.empty(n * T, v.intermediate_size, device=dev, dtype=BF)
            self._lin_fwd(p + "fc1.weight", y2, z, bias=w[p + "fc1.bias"])
            za = torch.empty_like(z)
            hip.quick_gelu_fwd(z, za)
            hn = torch.empty_like(h)
            self._lin_fwd(p + "fc2.weight", za, hn, bias=w[p + "fc2.bias"], resid=h2)
            if clip_saved is not None:
                clip_saved.append((h, y1, qkv, a, lse, h2, y2, z, za))
            h = hn
        for i in range(0 if vlora else cfg.vision_layers_used):
            p = f"vision.layers.{i}."
            hip.layernorm_fwd(h, w[p + "ln1.weight"], w[p + "ln1.bias"], y, v.layer_norm_eps)
            hip.gemm(hip.NT, y, w[p + "qkv.weight"], qkv, bias=w[p + "qkv.bias"])
            hip.attn_fwd(qkv[:, :dv], qkv[:, dv:2 * dv], qkv[:, 2 * dv:], a, None, None, n, T, v.num_heads,
                         v.head_dim, scale, False)
            h2 = torch.empty_like(h)
            hip.gemm(hip.NT, a, w[p + "out.weight"], h2, bias=w[p + "out.bias"], resid=h)
            hip.layernorm_fwd(h2, w[p + "ln2.weight"], w[p + "ln2.bias"], y, v.layer_norm_eps)
            hip.gemm(hip.NT, y, w[p + "fc1.weight"], z, bias=w[p + "fc1.bias"], act=hip.ACT_QUICK_GELU)
            h = torch.empty_like(h2)
            hip.gemm(hip.NT, z, w[p + "fc2.weight"], h, bias=w[p + "fc2.bias"], resid=h2)
        feats = h.view(n, T, dv)[:, 1:].reshape(n * npatch, dv)  # drop CLS ("default" select strategy)
        dt = cfg.text.hidden_size
        p1 = torch.empty(n * npatch, dt, device=dev, dtype=BF)
        self._lin_fwd("projector.linear_1.weight", feats, p1, bias=w["projector.linear_1.bias"])
        p1a = torch.empty_like(p1)
        hip.gelu_fwd(p1, p1a)
        img = torch.empty_like(p1)
        self._lin_fwd("projector.linear_2.weight", p1a, img, bias=w["projector.linear_2.bias"])
        # unfrozen tower: the patch conv's im2col operand and the pre-LayerNorm input are kept for their gradients
        extra = (col, emb) if (self.train_vision and save) else None
        return img, (feats, p1, p1a, clip_saved, n, extra)

    def _vision_backward(self, clip_saved, dfeats, n, extra=None):
        """Reverse sweep through the CLIP layers.  Adapted tower (LoRA): only the adapters receive gradients (base weights,
        LayerNorms and embeddings are frozen), so the chain carries dX only and stops at layer 0.  UNfrozen tower
        (`freeze_vision_module: false`, full fine-tuning): every Linear also gets its weight gradient (the wgrad GEMMs of
        `_lin_bwd`) and its bias gradient (column sums), every LayerNorm its dgamma / dbeta (`mmu_layernorm_bwd_params`),
        and the chain continues through the pre-LayerNorm into the position / class embeddings (sums over the images)
        and the patch conv (dW = d_patch^T im2col).  Attention (16 heads x 64, non-causal) runs on the head_dim-64
        instantiation of the backward kernels."""
        v, w, g = self.cfg.vision, self.w, self.g
        dev = self.device
        T, dv, H, D = v.num_patches + 1, v.hidden_size, v.num_heads, v.head_dim
        M = n * T
        tv = self.train_vision
        dh = torch.zeros(M, dv, device=dev, dtype=BF)
        dh.view(n, T, dv)[:, 1:].copy_(dfeats.view(n, T - 1, dv))  # the CLS row feeds nothing downstream
        scale = D ** -0.5
        delta = torch.empty(2, n, H, T, device=dev, dtype=torch.float32)
        cpart = torch.empty(hip.colsum_nparts(M) * max(3 * dv, v.intermediate_size), device=dev, dtype=torch.float32) if tv else None
        eps = v.layer_norm_eps
        for i in reversed(range(len(clip_saved))):
            p = f"vision.layers.{i}."
            h, y1, qkv, a, lse, h2, y2, z, za = clip_saved[i]
            clip_saved[i] = None
            dza = torch.empty_like(za)
            self._lin_bwd(p + "fc2.weight", dh, za, dza)
            if tv:
                hip.colsum(dh, cpart, g[p + "fc2.bias"], True)
            dz = torch.empty_like(z)
            hip.quick_gelu_bwd(z, dza, dz)
            dy2 = torch.empty_like(h)
            self._lin_bwd(p + "fc1.weight", dz, y2, dy2)
            if tv:
                hip.colsum(dz, cpart, g[p + "fc1.bias"], True)
                hip.layernorm_bwd_params(dy2, h2, g[p + "ln2.weight"], g[p + "ln2.bias"], eps)
            dh2 = torch.empty_like(h)
            hip.layernorm_bwd(dy2, h2, w[p + "ln2.weight"], dh, dh2, eps)
            da = torch.empty_like(h)
            self._lin_bwd(p + "out.weight", dh2, a, da)
            if tv:
                hip.colsum(dh2, cpart, g[p + "out.bias"], True)
            dqkv = torch.empty(M, 3 * dv, device=dev, dtype=BF)
            hip.attn_bwd(qkv[:, :dv], qkv[:, dv:2 * dv], qkv[:, 2 * dv:], a, lse, None, da, delta, dqkv[:, :dv],
                         dqkv[:, dv:2 * dv], dqkv[:, 2 * dv:], n, T, H, D, scale, False)  # native head_dim-64 kernels
            dy1 = torch.empty_like(h)
            self._lin_bwd(p + "qkv.weight", dqkv, y1, dy1)
            if tv:
                hip.colsum(dqkv, cpart, g[p + "qkv.bias"], True)
                hip.layernorm_bwd_params(dy1, h, g[p + "ln1.weight"], g[p + "ln1.bias"], eps)
            dhn = torch.empty_like(h)
            hip.layernorm_bwd(dy1, h, w[p + "ln1.weight"], dh2, dhn, eps)
            dh = dhn
        if not tv:
            return
        # below layer 0: pre-LayerNorm, then emb[b, 0] = class + pos[0], emb[b, 1 + p] = patch[b, p] + pos[1 + p]
        col, emb = extra
        hip.layernorm_bwd_params(dh, emb, g["vision.pre_ln.weight"], g["vision.pre_ln.bias"], eps)
        demb = torch.empty_like(emb)
        hip.layernorm_bwd(dh, emb, w["vision.pre_ln.weight"], None, demb, eps)
        # position embedding: the sum over the images of every token row ([n, T * dv] column sums); class embedding: the
        # same sum restricted to token 0 (it only ever meets pos[0])
        ppart = torch.empty(hip.colsum_nparts(n) * T * dv, device=dev, dtype=torch.float32)
        hip.colsum(demb.view(n, T * dv), ppart, g["vision.position_embedding"].view(-1), True)
        hip.colsum(demb.view(n, T, dv)[:, 0].contiguous(), ppart, g["vision.class_embedding"], True)
        # patch conv (no bias): dW [dv, C P P (padded)] += d_patch^T im2col; the zero-padded tail columns of the im2col operand
        # keep the padded weight columns at zero gradient
        dpatch = demb.view(n, T, dv)[:, 1:].reshape(n * (T - 1), dv)
        self._wgrad(dpatch, col, "vision.patch_embed.weight")

    @property
    def decoder_layers(self):
        """Handles for `trainer_utils.forward_with_cache(model, inputs, module=model.decoder_layers[i])` — the reference
        passes `model.language_model.model.layers[i]` there (mm/trainer.py:302-310)."""
        from .trainer_utils import DecoderLayerHandle

        return [DecoderLayerHandle(i) for i in range(self.cfg.text.num_layers)]

    def hidden_states(self, layer: int, input_ids=None, attention_mask=None, pixel_values=None, position_ids=None,
                      **_unused):
        """Output of decoder layer `layer` ([B,S,d], attached to autograd) — what the reference captures with a
        forward hook on `language_model.model.layers[layer]` (mm/trainer_utils.py:24-45, mm/trainer.py:302-326).
        The sweep stops at that layer: the layers above cannot influence the captured activation or its gradient."""
        dev = self.device
        kmask = self._kmask(attention_mask)
        ids = input_ids.to(dev, dtype=torch.int64).contiguous()
        B, S = ids.shape
        pos = self._positions(position_ids, attention_mask, B, S)
        pix = None
        if pixel_values is not None and pixel_values.numel() > 0:
            pix = pixel_values.to(dev, dtype=torch.float32).contiguous()
        if torch.is_grad_enabled() and self.train_flat.requires_grad:
            return _LlavaFn.apply(self.train_flat, self, ids, kmask, pix, pos, layer)
        with torch.no_grad():
            out, _ = self._forward_impl(ids, kmask, pix, pos, save=False, stop_layer=layer)
        return out

    def _decoder_layer(self, i, h, y, kmask, pos, B, S, save):
        """One LlamaDecoderLayer forward (modeling_llama.py:295-326) on [tokens, d] rows: returns (output, what the backward
        sweep needs besides the input: rstd1, post-RoPE q|k|v, attention out, lse, post-attention residual, rstd2, gate|up,
        act^T or None).  `y` is a [tokens, d] scratch for the two normalised inputs.  Also the recompute body of the backward
        sweep when `recompute_activations` is on."""
        t, w = self.cfg.text, self.w
        dev = self.device
        M = B * S
        dt, I, H, D = t.hidden_size, t.intermediate_size, t.num_heads, t.head_dim
        scale = D ** -0.5
        p = f"text.layers.{i}."
        self._wait_params(i)
        rstd1 = torch.empty(M, device=dev, dtype=torch.float32)
        hip.rmsnorm_fwd(h, w[p + "input_norm.weight"], y, rstd1, t.rms_norm_eps)
        qkv = torch.empty(M, 3 * dt, device=dev, dtype=BF)
        self._qkv_fwd(p + "qkv.weight", y, qkv, pos, H, D, dt)
        a = torch.empty(M, dt, device=dev, dtype=BF)
        lse = torch.empty(B, H, S, device=dev, dtype=torch.float32) if save else None
        # with keep_act_t the attention forward also leaves a^T (the o-projection's wgrad operand, + M*d*2 B per layer: 8.5 GB
        # at 7B / 30 sequences) from the staging image of its row-major store: the backward then transposes dh2 only
        aT = (torch.empty(dt, M, device=dev, dtype=BF)
              if (save and self.keep_act_t and self.attn_out_t and self._use_t(M) and D == 128 and S % 8 == 0) else None)
        hip.attn_fwd(qkv[:, :dt], qkv[:, dt:2 * dt], qkv[:, 2 * dt:], a, lse, kmask, B, S, H, D, scale, True, out_t=aT)
        h2 = torch.empty(M, dt, device=dev, dtype=BF)
        self._lin_fwd(p + "o.weight", a, h2, resid=h)
        rstd2 = torch.empty(M, device=dev, dtype=torch.float32)
        hip.rmsnorm_fwd(h2, w[p + "post_norm.weight"], y, rstd2, t.rms_norm_eps)
        gu = torch.empty(M, 2 * I, device=dev, dtype=BF)
        act = torch.empty(M, I, device=dev, dtype=BF)
        actT = None
        if save and self.keep_act_t and self._use_t(M):
            actT = torch.empty(I, M, device=dev, dtype=BF)  # the down-proj wgrad operand, kept until backward
        # gate|up projection with SwiGLU in its epilogue (round 4): gu, act and act^T leave ONE launch; the elementwise
        # kernel that re-read gu (4 x M*I*2 B of HBM traffic per layer) is gone.  Shapes off the 256^2 kernel and adapted
        # (LoRA) projections take the two-kernel path below.
        if not (self.fuse_swiglu and not self.lora and hip.gemm_swiglu_fwd(y, w[p + "gate_up.weight"], gu, act, actT)):
            self._lin_fwd(p + "gate_up.weight", y, gu)
            if actT is not None:
                hip.swiglu_fwd_both(gu, act, actT)
            else:
                hip.swiglu_fwd(gu, act)
        hn = torch.empty(M, dt, device=dev, dtype=BF)
        self._lin_fwd(p + "down.weight", act, hn, resid=h2)
        if save and actT is None and self.keep_act_t and self.lora:
            actT = act  # LoRA sweep: the row-major act is the down-proj adapter's wgrad operand (no transposed path)
        return hn, (rstd1, qkv, (a, aT) if aT is not None else a, lse, h2, rstd2, gu, actT)

    def _forward_impl(self, ids, kmask, pix, pos, save: bool, stop_layer: Optional[int] = None,
                      rows: Optional[torch.Tensor] = None):
        self._sync_derived()
        cfg, t, w = self.cfg, self.cfg.text, self.w
        dev = self.device
        B, S = ids.shape
        M = B * S
        dt, I, H, D = t.hidden_size, t.intermediate_size, t.num_heads, t.head_dim
        assert S <= t.max_position
        saved = {"B": B, "S": S, "ids": ids, "kmask": kmask, "pos": pos, "layers": [], "stop_layer": stop_layer}
        self._fwd_U = saved.setdefault("U", {}) if (save and self.lora) else None
        self._drop_seeds = (saved.setdefault("drop", {})
                            if (save and self.lora and self.lora_spec.dropout > 0 and self.training) else None)
        self._wait_params("pre")
        img = None
        slot = None
        if pix is not None:
            img, vis_saved = self._vision_forward(pix, save)
            slot = torch.empty(M, device=dev, dtype=torch.int32)
            cnt = torch.empty(1, device=dev, dtype=torch.int32)
            hip.image_slots(ids.view(-1), cfg.image_token_index, slot, cnt)
            saved["vis"] = vis_saved
        saved["slot"] = slot
        h = torch.empty(M, dt, device=dev, dtype=BF)
        hip.embed_fwd(ids.view(-1), slot, w["text.embed_tokens.weight"], img, h)
        scale = D ** -0.5
        y = torch.empty(M, dt, device=dev, dtype=BF)
        self._last_capture = None
        for i in range(t.num_layers):
            hn, kept = self._decoder_layer(i, h, y, kmask, pos, B, S, save)
            if save:
                # gradient checkpointing (model_config.yaml `gradient_checkpointing`, mm/forget.py:123-124): only the layer's
                # input survives the forward; the backward sweep re-runs `_decoder_layer` for everything else
                saved["layers"].append((h,) + ((None,) * 8 if self.recompute_activations else kept))
            h = hn
            if self.capture_layer is not None and i == self.capture_layer:
                self._last_capture = h.view(B, S, dt)
            if stop_layer is not None and i == stop_layer:
                self._fwd_U = self._drop_seeds = None
                return h.view(B, S, dt), saved
        self._wait_params("post")
        if rows is not None:
            # loss-only forward: final norm + lm_head on the supervised rows only (padding slots are zero rows, their
            # targets are -100); returns the compact logits [len(rows), V]
            Mc = rows.numel()
            hc = torch.empty(Mc, dt, device=dev, dtype=BF)
            hip.gather_rows(h, rows, hc)
            del h
            rstdf = torch.empty(Mc, device=dev, dtype=torch.float32)
            yc = y[:Mc] if Mc <= M else torch.empty(Mc, dt, device=dev, dtype=BF)
            hip.rmsnorm_fwd(hc, w["text.norm.weight"], yc, rstdf, t.rms_norm_eps)
            logits = torch.empty(Mc, t.vocab_size, device=dev, dtype=BF)
            self._lin_fwd("lm_head.weight", yc, logits)
            if save:
                saved["final"] = (hc, rstdf)
                saved["rows"] = rows
            self._fwd_U = self._drop_seeds = None
            return logits, saved
        rstdf = torch.empty(M, device=dev, dtype=torch.float32)
        hip.rmsnorm_fwd(h, w["text.norm.weight"], y, rstdf, t.rms_norm_eps)
        logits = torch.empty(M, t.vocab_size, device=dev, dtype=BF)
        self._lin_fwd("lm_head.weight", y, logits)
        if save:
            saved["final"] = (h, rstdf)
        self._fwd_U = self._drop_seeds = None
        return logits.view(B, S, t.vocab_size), saved

    # ---------------------------------------------------------------------------------------- backward
    def _backward_impl(self, saved, dlogits, fire_hooks: bool = True):
        cfg, t, w = self.cfg, self.cfg.text, self.w
        dev = self.device
        self.ensure_grad()
        g = self.g
        self._bwd_U = saved.get("U")
        self._drop_seeds = saved.get("drop")
        B, S = saved["B"], saved["S"]
        M = B * S
        dt, I, H, D, V = t.hidden_size, t.intermediate_size, t.num_heads, t.head_dim, t.vocab_size
        scale = D ** -0.5
        rows = saved.get("rows")
        Mh = rows.numel() if rows is not None else M   # rows that went through the final norm and the lm_head
        dlog = dlogits.reshape(Mh, -1)
        if not dlog.is_contiguous():
            dlog = dlog.contiguous()
        if dlog.dtype != BF:
            dlog = dlog.to(BF)
        npart = hip.rmsnorm_bwd_nparts(M)
        dw_part = torch.empty(npart * dt, device=dev, dtype=torch.float32)
        y = torch.empty(M, dt, device=dev, dtype=BF)
        hook = self.grad_ready_hook if fire_hooks else None

        stop_layer = saved.get("stop_layer")
        dy = torch.empty(M, dt, device=dev, dtype=BF)
        use_t = self._use_t(M)
        yT = torch.empty(dt, M, device=dev, dtype=BF) if use_t else None
        if stop_layer is None and rows is not None:
            # loss-only forward: the head ran on the supervised rows; their hidden-state gradients are scattered back
            # into a zero [tokens, d] gradient (every other position reaches the loss through attention only)
            hc, rstdf = saved["final"]
            dyc = torch.empty(Mh, dt, device=dev, dtype=BF)
            if self._use_t(Mh):
                ycT = torch.empty(dt, Mh, device=dev, dtype=BF)
                hip.rmsnorm_apply_t(hc, w["text.norm.weight"], rstdf, ycT)
                self._wgrad_t(dlog, None, ycT, "lm_head.weight")
                self._dgrad(dlog, "lm_head.weight", dyc)
                del ycT
            else:
                yc = torch.empty(Mh, dt, device=dev, dtype=BF)
                hip.rmsnorm_fwd(hc, w["text.norm.weight"], yc, None, t.rms_norm_eps)
                self._lin_bwd("lm_head.weight", dlog, yc, dyc)
                del yc
            dhc = torch.empty(Mh, dt, device=dev, dtype=BF)
            dwp = dw_part if hip.rmsnorm_bwd_nparts(Mh) <= npart else torch.empty(hip.rmsnorm_bwd_nparts(Mh) * dt, device=dev,
                                                                                  dtype=torch.float32)
            hip.rmsnorm_bwd(dyc, hc, w["text.norm.weight"], rstdf, None, dhc, dwp, g.get("text.norm.weight"), True)
            dh = torch.zeros(M, dt, device=dev, dtype=BF)
            hip.scatter_rows(dhc, rows, dh)
            del dyc, dhc, hc
            top = t.num_layers - 1
        elif stop_layer is None:
            h, rstdf = saved["final"]
            if use_t:
                hip.rmsnorm_apply_t(h, w["text.norm.weight"], rstdf, yT)
                self._wgrad_t(dlog, None, yT, "lm_head.weight")
                self._dgrad(dlog, "lm_head.weight", dy)
            else:
                hip.rmsnorm_fwd(h, w["text.norm.weight"], y, None, t.rms_norm_eps)
                self._lin_bwd("lm_head.weight", dlog, y, dy)
            dh = torch.empty(M, dt, device=dev, dtype=BF)
            hip.rmsnorm_bwd(dy, h, w["text.norm.weight"], rstdf, None, dh, dw_part, g.get("text.norm.weight"), True)
            top = t.num_layers - 1
        else:
            dh = dlog.clone()  # gradient of the captured activation; the layers above get no gradient
            top = stop_layer
        if hook:  # everything laid out after decoder layer `top` (upper layers, final norm, lm_head) is final
            above = [self.range_of(n) for n in self._toff if self._is_above(n, top)]
            if above:
                hook(min(a for a, _ in above), max(b for _, b in above))
        delta = torch.empty(2, B, H, S, device=dev, dtype=torch.float32)
        layers = saved["layers"]
        for i in reversed(range(top + 1)):
            p = f"text.layers.{i}."
            hin, rstd1, qkv, a, lse, h2, rstd2, gu, actT = layers[i]
            layers[i] = None
            if rstd1 is None:  # gradient checkpointing: re-run the layer's forward from its input (same kernels, same bits)
                self._fwd_U = self._bwd_U  # LoRA: the x A^T products of the recomputed Linears feed the adapter wgrads below
                _, (rstd1, qkv, a, lse, h2, rstd2, gu, actT) = self._decoder_layer(i, hin, y, saved["kmask"], saved["pos"], B, S, True)
                self._fwd_U = None
            # MLP
            dact = None
            dgu = torch.empty(M, 2 * I, device=dev, dtype=BF)
            if use_t:  # wgrad operands come out of their producers already transposed
                if actT is None:
                    actT = torch.empty(I, M, device=dev, dtype=BF)
                    hip.swiglu_fwd_t(gu, actT)
                self._wgrad_t(dh, None, actT, p + "down.weight")
                del actT
                dguT = torch.empty(2 * I, M, device=dev, dtype=BF)
                # down-projection dgrad with the SwiGLU backward in its epilogue (round 4): dact never reaches HBM, dgu and
                # dgu^T leave the GEMM (7 x M*I*2 B of elementwise traffic per layer gone, 2 + 4 of them back as epilogue I/O)
                wdn = p + "down.weight"
                if not (self.fuse_swiglu and hip.gemm_swiglu_bwd(dh, self.wT[wdn] if wdn in self.wT else w[wdn], wdn in self.wT,
                                                                 gu, dgu, dguT)):
                    dact = torch.empty(M, I, device=dev, dtype=BF)
                    self._dgrad(dh, wdn, dact)
                    hip.swiglu_bwd_t(gu, dact, dgu, dguT)
                del dact, gu
                hip.rmsnorm_apply_t(h2, w[p + "post_norm.weight"], rstd2, yT)
                self._wgrad_t(dgu, dguT, yT, p + "gate_up.weight")
                self._dgrad(dgu, p + "gate_up.weight", dy)
                del dguT
            else:
                act = actT
                if act is None:
                    act = torch.empty(M, I, device=dev, dtype=BF)
                    hip.swiglu_fwd(gu, act)
                dact = torch.empty(M, I, device=dev, dtype=BF)
                self._lin_bwd(p + "down.weight", dh, act, dact)
                hip.swiglu_bwd(gu, dact, dgu)
                del act, dact, gu
                hip.rmsnorm_fwd(h2, w[p + "post_norm.weight"], y, None, t.rms_norm_eps)
                self._lin_bwd(p + "gate_up.weight", dgu, y, dy)
            del dgu
            dh2 = torch.empty(M, dt, device=dev, dtype=BF)
            hip.rmsnorm_bwd(dy, h2, w[p + "post_norm.weight"], rstd2, dh, dh2, dw_part, g.get(p + "post_norm.weight"),
                            True)
            # attention
            da = dh  # reuse the buffer of the (now consumed) upstream gradient
            aT = None
            if isinstance(a, tuple):
                a, aT = a
            if aT is not None and use_t:   # a^T came out of the attention forward: only dh2 is transposed here
                self._wgrad_t(dh2, None, aT, p + "o.weight")
                self._dgrad(dh2, p + "o.weight", da)
                del aT
            else:
                self._lin_bwd(p + "o.weight", dh2, a, da)
            dqkv = torch.empty(M, 3 * dt, device=dev, dtype=BF)
            # the backward kernels also leave dqkv^T for the q|k|v weight gradient (the persistent dK/dV kernel writes its
            # two thirds from the staging image it already has; only valid while RoPE is fused into the same stores)
            dqkvT = (torch.empty(3 * dt, M, device=dev, dtype=BF)
                     if (use_t and self.fuse_rope and self.fuse_dqkv_t and M % 8 == 0) else None)
            hip.attn_bwd(qkv[:, :dt], qkv[:, dt:2 * dt], qkv[:, 2 * dt:], a, lse, saved["kmask"], da, delta,
                         dqkv[:, :dt], dqkv[:, dt:2 * dt], dqkv[:, 2 * dt:], B, S, H, D, scale, True,
                         rope=(saved["pos"], self._cos, self._sin) if self.fuse_rope else None, dqkv_t=dqkvT)
            if not self.fuse_rope:
                hip.rope(dqkv, saved["pos"], self._cos, self._sin, H, H, D, dt, inverse=True)
            del qkv, a
            if use_t:
                hip.rmsnorm_apply_t(hin, w[p + "input_norm.weight"], rstd1, yT)
                self._wgrad_t(dqkv, dqkvT, yT, p + "qkv.weight")
                del dqkvT
                self._dgrad(dqkv, p + "qkv.weight", dy)
            else:
                hip.rmsnorm_fwd(hin, w[p + "input_norm.weight"], y, None, t.rms_norm_eps)
                self._lin_bwd(p + "qkv.weight", dqkv, y, dy)
            del dqkv
            dh = da  # overwrite: da is dead after attn_bwd
            hip.rmsnorm_bwd(dy, hin, w[p + "input_norm.weight"], rstd1, dh2, dh, dw_part,
                            g.get(p + "input_norm.weight"), True)
            del dh2, hin, h2
            if hook and self.layer_grad_range(i):
                hook(*self.layer_grad_range(i))
        # embeddings / projector
        slot = saved["slot"]
        dimg = None
        clip_saved = None
        if "vis" in saved:
            feats, p1, p1a, clip_saved, n_img, vis_extra = saved["vis"]
            dimg = torch.zeros(p1.shape[0], dt, device=dev, dtype=BF)
        if self._embed_scratch is None:
            self._embed_scratch = torch.zeros(V, dt, device=dev, dtype=torch.float32)
            self._embed_flag = torch.zeros(V, device=dev, dtype=torch.int32)
        hip.embed_bwd(saved["ids"].view(-1), slot, dh, dimg, self._embed_scratch, self._embed_flag,
                      g.get("text.embed_tokens.weight"), V)
        full = not self.lora
        vlora = dimg is not None and clip_saved is not None
        if dimg is not None and (full or vlora or "projector.linear_2.weight" in self.lora
                                 or "projector.linear_1.weight" in self.lora):
            n = dimg.shape[0]
            cpart = torch.empty(hip.colsum_nparts(n) * dt, device=dev, dtype=torch.float32)
            dp1a = torch.empty_like(p1)
            self._lin_bwd("projector.linear_2.weight", dimg, p1a, dp1a)
            if full:
                hip.colsum(dimg, cpart, g["projector.linear_2.bias"], True)
            dp1 = torch.empty_like(p1)
            hip.gelu_bwd(p1, dp1a, dp1)
            dfeats = torch.empty_like(feats) if vlora else None
            self._lin_bwd("projector.linear_1.weight", dp1, feats, dfeats)
            if full:
                hip.colsum(dp1, cpart, g["projector.linear_1.bias"], True)
            if vlora:
                self._vision_backward(clip_saved, dfeats, n_img, vis_extra)
        if hook:  # everything else: projector, token embedding, the CLIP tower's adapters
            below = [self.range_of(n) for n in self._toff
                     if n.startswith("projector.") or n.startswith("text.embed_tokens") or n.startswith("vision.")]
            if below:
                hook(min(a for a, _ in below), max(b for _, b in below))


class _LlavaFn(torch.autograd.Function):
    """Autograd edge: logits = f(train_flat).  backward() runs the explicit reverse sweep, ACCUMULATES straight
    into the model's flat gradient buffer (train_flat.grad) and returns no tensor for it."""

    @staticmethod
    def forward(ctx, train_flat, model: LlavaHIP, ids, kmask, pix, pos, stop_layer=None, rows=None):
        if model.autograd_grads and model.live_sweeps == 0 and model._grad_flat is not None:
            model._grad_flat.zero_()  # first forward of a micro-batch: what the last one accumulated has been handed to autograd
        logits, saved = model._forward_impl(ids, kmask, pix, pos, save=True, stop_layer=stop_layer, rows=rows)
        model.live_sweeps += 1
        ctx.model = model
        ctx.saved = saved
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model, saved = ctx.model, ctx.saved
        ctx.saved = None
        model.live_sweeps = max(0, model.live_sweeps - 1)
        model._backward_impl(saved, dlogits, fire_hooks=model.live_sweeps == 0)
        # autograd-gradient mode: the last sweep of the micro-batch hands the accumulated buffer to autograd (see
        # LlavaHIP.enable_autograd_gradients); otherwise `.grad` IS that buffer and there is nothing to return
        g = model._grad_flat if (model.autograd_grads and model.live_sweeps == 0) else None
        return g, None, None, None, None, None, None, None
