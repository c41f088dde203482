"""Host logic of the loss-only forward: which hidden-state rows reach the lm_head (llava.supervised_rows) against a
plain loop over HF's shifted-CE rule (position s is scored against labels[s + 1] unless that label is -100)."""
import torch

from multimodal_unlearning_amd.llava import supervised_rows


def _loop(labels, V):
    B, S = labels.shape
    per = []
    for b in range(B):
        per.append([(b * S + s, int(labels[b, s + 1])) for s in range(S - 1) if 0 <= int(labels[b, s + 1]) < V])
    cap = max(8, (max(len(p) for p in per) + 7) // 8 * 8)
    cap = min(cap, (S - 1 + 7) // 8 * 8)
    rows, tgt = [], []
    for p in per:
        rows += [r for r, _ in p] + [-1] * (cap - len(p))
        tgt += [t for _, t in p] + [-100] * (cap - len(p))
    return rows, tgt, cap


def test_supervised_rows_match_the_shifted_ce_rule():
    g = torch.Generator().manual_seed(0)
    V = 50
    for B, S in ((1, 9), (3, 17), (4, 64), (2, 8)):
        labels = torch.randint(0, V, (B, S), generator=g)
        labels[torch.rand(B, S, generator=g) < 0.6] = -100
        labels[0, :] = -100 if B > 1 else labels[0, :]          # a sequence with nothing supervised
        if B > 2:
            labels[2, 1:] = torch.randint(0, V, (S - 1,), generator=g)  # every position supervised
        rows, tgt, cap = supervised_rows(labels, V, "cpu")
        er, et, ec = _loop(labels, V)
        assert cap == ec and cap % 8 == 0
        assert rows.tolist() == er and tgt.tolist() == et
        assert rows.dtype == torch.int64 and tgt.dtype == torch.int64


def test_out_of_vocabulary_labels_are_ignored_like_minus_100():
    labels = torch.tensor([[5, 7, 99, -100, 3, 2, 1, 0, 4]])
    rows, tgt, cap = supervised_rows(labels, 10, "cpu")
    assert rows.tolist() == [0, 3, 4, 5, 6, 7, -1, -1] and tgt.tolist() == [7, 3, 2, 1, 0, 4, -100, -100]
