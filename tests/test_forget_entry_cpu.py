"""Host logic of the forget entry point (counterpart of mm/forget.py:20-68 + run_exps.py's `k=v` overrides)."""
import pytest

from multimodal_unlearning_amd.forget import ForgetRunConfig, _architecture, main, parse_overrides


def test_overrides_follow_the_reference_flag_surface():
    rc = parse_overrides(["forget_loss=KL", "LoRA.r=8", "LoRA.alpha=16", "lr=2e-5", "batch_size=4", "num_epochs=3",
                          "overwrite_dir=false", "model_family=llava-1.5-13b", "l1_lambda=0.01", "l_norm_from=init"],
                         ForgetRunConfig())
    assert (rc.forget_loss, rc.lora_r, rc.lora_alpha, rc.lr, rc.batch_size, rc.num_epochs) == ("KL", 8, 16, 2e-5, 4, 3)
    assert rc.overwrite_dir is False and rc.l1_lambda == 0.01 and rc.l_norm_from == "init"
    # config/mm/forget.yaml:50: save_dir: ${model_path}/${forget_loss}_${forget_split}_L1_${l1_lambda}_from${l_norm_from}
    assert rc.save_dir == "llava-1.5-13b/KL_forget10_L1_0.01_frominit"
    assert _architecture(rc.model_family).text.hidden_size == 5120
    with pytest.raises(SystemExit):
        parse_overrides(["not_a_key=1"], ForgetRunConfig())
    with pytest.raises(SystemExit):
        parse_overrides(["novalue"], ForgetRunConfig())
    with pytest.raises(SystemExit):
        _architecture("qwen-vl2-2b")  # the reference's other family is outside this path


def test_entry_refuses_to_run_without_a_gpu(monkeypatch):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        main(["model_family=tiny", "save_dir=/tmp/mmu_never_written", "forget_data_path=synthetic:8"])
    assert "MI355X" in str(e.value)


def test_existing_directory_is_refused_before_the_rendezvous(tmp_path, monkeypatch):
    """mm/forget.py:40-44 with overwrite_dir=false: every rank must leave, so the check runs before init_process_group
    (WORLD_SIZE=2 here: a rendezvous attempt would hang the test instead of raising)."""
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    with pytest.raises(SystemExit) as e:
        main(["model_family=tiny", f"save_dir={tmp_path}", "overwrite_dir=false", "forget_data_path=synthetic:8"])
    assert "Directory already exists" in str(e.value)
