"""Host logic of the merged gradient-accumulation path (no GPU): MMTrainerForgetting.merge_micro_batches (shapes, host
views, left padding of ragged micro-batches, refusals) and the world > 1 guard of the transformers.Trainer-hosted
arrangement (mm/trainer.py:112, mm/forget.py:161)."""
import types

import pytest
import torch

from multimodal_unlearning_amd.config import tiny_test_config
from multimodal_unlearning_amd.synthetic import make_forget_batch
from multimodal_unlearning_amd.trainer import CustomTrainerForgetting, MMTrainerForgetting, hosted_by_hf_trainer


def _stub_model():
    return types.SimpleNamespace(loss_segments=1)


def _tr(loss, cls=MMTrainerForgetting, **kw):
    return cls(model=None, forget_loss=loss, teacher_model=object(), loss_beta=1.0, l1_lambda=0.0, l0_lambda=0.0, **kw)


def test_merge_concatenates_equal_micro_batches_in_order():
    cfg = tiny_test_config()
    mbs = [make_forget_batch(cfg, 2, 40, seed=i, n_question=6) for i in range(3)]
    m = _tr("grad_diff_forget_ce_retain_ce").merge_micro_batches(_stub_model(), mbs)
    assert set(m) == {"forget", "retain"}
    for split in m:
        for k, v in m[split].items():
            assert torch.equal(v, torch.cat([mb[split][k] for mb in mbs], dim=0)), (split, k)
    assert m["forget"]["pixel_values"].shape[0] == 6


def test_merge_left_pads_ragged_micro_batches_for_ce_only_losses():
    cfg = tiny_test_config()
    mbs = [make_forget_batch(cfg, 2, S, seed=i, n_question=6) for i, S in enumerate((40, 33))]
    m = _tr("grad_diff_forget_ce_retain_ce").merge_micro_batches(_stub_model(), mbs, pad_token_id=0)
    f = m["forget"]
    assert f["input_ids"].shape == (4, 40)
    assert torch.equal(f["input_ids"][:2], mbs[0]["forget"]["input_ids"])
    assert torch.equal(f["input_ids"][2:, 7:], mbs[1]["forget"]["input_ids"]) and bool((f["input_ids"][2:, :7] == 0).all())
    assert bool((f["attention_mask"][2:, :7] == 0).all()) and bool((f["labels"][2:, :7] == -100).all())
    assert torch.equal(f["labels"][2:, 7:], mbs[1]["forget"]["labels"])
    # losses normalised over all B*S rows cannot take padding rows
    assert _tr("kl").merge_micro_batches(_stub_model(), mbs) is None
    assert _tr("npo").merge_micro_batches(_stub_model(), mbs) is None


def test_merge_refusals():
    cfg = tiny_test_config()
    mbs = [make_forget_batch(cfg, 2, 40, seed=i, n_question=6) for i in range(2)]
    tr = _tr("grad_diff_forget_ce_retain_ce")
    assert tr.merge_micro_batches(_stub_model(), mbs[:1]) is None                       # nothing to merge
    assert tr.merge_micro_batches(types.SimpleNamespace(), mbs) is None                 # a foreign model (no loss_segments)
    assert _tr("rmu").merge_micro_batches(_stub_model(), mbs) is None                   # control vector of batch shape
    assert _tr("sku", loss_args={"bad_weight": 1, "random_weight": 1, "normal_weight": 1}).merge_micro_batches(_stub_model(), mbs) is None
    other = make_forget_batch(cfg, 3, 40, seed=9, n_question=6)                         # another batch size
    assert tr.merge_micro_batches(_stub_model(), [mbs[0], other]) is None
    assert tr.merge_micro_batches(_stub_model(), [mbs[0], {"forget": mbs[1]["forget"]}]) is None
    with pytest.raises(ValueError):
        _tr("rmu").compute_loss(types.SimpleNamespace(parameters=lambda: iter([torch.zeros(1)]), loss_segments=1), mbs[0], _merge=2)


def test_text_trainer_merges_tuples_and_refuses_idk():
    g = torch.Generator().manual_seed(0)
    tup = lambda: tuple(torch.randint(0, 50, (2, 16), generator=g) for _ in range(3))
    mbs = [[tup(), tup()] for _ in range(3)]
    tr = CustomTrainerForgetting(model=None, forget_loss="grad_ascent", oracle_model=None, loss_beta=1.0, l1_lambda=0.0, l0_lambda=0.0)
    m = tr.merge_micro_batches(_stub_model(), mbs)
    assert len(m) == 2 and all(len(t) == 3 for t in m)
    for j in range(2):
        for i in range(3):
            assert torch.equal(m[j][i], torch.cat([mb[j][i] for mb in mbs], dim=0))
    tr2 = CustomTrainerForgetting(model=None, forget_loss="idk", oracle_model=None, loss_beta=1.0, l1_lambda=0.0, l0_lambda=0.0)
    assert tr2.merge_micro_batches(_stub_model(), mbs) is None


def test_hosted_trainer_refuses_world_size_above_one_for_flat_models(monkeypatch):
    """torch DDP listens on AccumulateGrad of the parameter; LlavaHIP's edge accumulates by hand, so the
    hosted arrangement must fail loudly at world > 1 instead of training unsynchronised replicas."""
    monkeypatch.setenv("WORLD_SIZE", "2")
    flat_model = types.SimpleNamespace(train_flat=torch.zeros(4))
    with pytest.raises(RuntimeError, match="ForgetEngine"):
        hosted_by_hf_trainer()(model=flat_model, forget_loss="grad_ascent")
