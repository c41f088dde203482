"""The plug-in hosted by a REAL transformers.Trainer (mm/trainer.py:112 `class MMTrainerForgetting(Trainer)`): HF's
`training_step` / `train()` drive `compute_loss` of `hosted_by_hf_trainer()`.  CPU: a tiny torch stand-in with the
HF output shape (`.loss`, `.logits`) takes the place of the model — the GPU twin with LlavaHIP is in
tests/test_model_gpu.py::test_compute_loss_under_hf_trainer."""
import tempfile
import types

import pytest
import torch
import torch.nn as nn

from multimodal_unlearning_amd.llava import LlavaOutput
from multimodal_unlearning_amd.trainer import MMTrainerForgetting, hosted_by_hf_trainer


class TinyLM(nn.Module):
    def __init__(self, V=32, d=16):
        super().__init__()
        self.emb, self.head, self.V = nn.Embedding(V, d), nn.Linear(d, V), V
        self.config = types.SimpleNamespace(image_token_index=V - 1)

    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, labels=None, **kw):
        lg = self.head(self.emb(input_ids))
        loss = None
        if labels is not None:
            loss = nn.functional.cross_entropy(lg[:, :-1].reshape(-1, self.V), labels[:, 1:].reshape(-1), ignore_index=-100)
        return LlavaOutput(loss=loss, logits=lg)


def _mb(seed):
    g = torch.Generator().manual_seed(seed)

    def sp():
        ids = torch.randint(0, 31, (2, 8), generator=g)
        return {"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": ids.clone()}

    return {"forget": sp(), "retain": sp()}


def test_plugin_runs_under_transformers_trainer():
    import transformers

    torch.manual_seed(0)
    m = TinyLM()
    data = [_mb(i) for i in range(4)]
    H = hosted_by_hf_trainer()
    assert issubclass(H, transformers.Trainer) and issubclass(H, MMTrainerForgetting)
    args = transformers.TrainingArguments(output_dir=tempfile.mkdtemp(), per_device_train_batch_size=1,
                                          gradient_accumulation_steps=2, max_steps=2, learning_rate=1e-2, report_to=[],
                                          remove_unused_columns=False, disable_tqdm=True, use_cpu=True, save_strategy="no")
    kw = dict(forget_loss="grad_diff_forget_ce_retain_ce", teacher_model=None, loss_beta=0.5, l1_lambda=0, l0_lambda=0,
              l_norm_from="zero", loss_args={"fuse_pairs": False})
    tr = H(model=m, args=args, train_dataset=data, data_collator=lambda x: x[0], **kw)
    # HF's training_step = compute_loss + loss / GA + backward: same number and gradients as the stand-alone plug-in
    tr.current_gradient_accumulation_steps = 2
    m.zero_grad()
    hf_loss = tr.training_step(m, data[0])
    g_hf = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    ours = MMTrainerForgetting(model=m, **kw).compute_loss(m, data[0])
    (ours / 2).backward()
    assert float(hf_loss) == pytest.approx(float(ours) / 2, rel=1e-6)
    for a, p in zip(g_hf, m.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-5, atol=1e-7)
    m.zero_grad()
    before = [p.detach().clone() for p in m.parameters()]
    tr.train()
    assert tr.state.global_step == 2
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, m.parameters()))
    with pytest.raises(ValueError):  # the plug-in's own validation still runs before the host is built
        H(model=m, args=args, forget_loss="npo", teacher_model=None, loss_beta=1.0, l1_lambda=0, l0_lambda=0, l_norm_from="zero")
