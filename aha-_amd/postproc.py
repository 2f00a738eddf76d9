"""Score post-processing of the evaluation path, numpy only.

Restated from the formulas (not the code) of:
  fused score            test/evaluate.py:584-589   alpha*info + beta*rel - eps*max(0, unc - tau)
  TVSum metrics          test/tvsum/tvsum_utils.py:9-91,202-220   (mAP@rho, top-5 mAP, Spearman, Kendall tau-b, F1@rho)
  Mr.HiSum metrics       test/hisum/hisum_eval.py:9-134           (shot mAP@rho, budgeted F1)
  knapsack selection     test/highlight_generator.py:8-37         (unit-cost 0/1 knapsack with its tie rule)
The reference delegates to sklearn/scipy (average_precision_score, spearmanr, kendalltau, f1_score);
here they are written out so the metrics carry no sklearn dependency.  Pinned against the reference's
own functions in tests/test_postproc.py (live when /root/reference is present, and via
tests/golden/postproc.json).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence

import numpy as np


# ---- fused score ---------------------------------------------------------------------------------
def fuse_scores(debug_data: Sequence[dict], alpha: float, beta: float, epsilon: float,
                uncertainty_threshold: float) -> np.ndarray:
    """test/evaluate.py:584-589 over a prediction's `debug_data` rows."""
    out = np.empty(len(debug_data), dtype=np.float64)
    for i, e in enumerate(debug_data):
        s = alpha * e["informative_score"] + beta * e["relevance_score"]
        if e["uncertainty_score"] >= uncertainty_threshold:
            s -= (e["uncertainty_score"] - uncertainty_threshold) * epsilon
        out[i] = s
    return out


# ---- primitives the reference takes from sklearn / scipy -----------------------------------------------
def average_precision(y_true: np.ndarray, y_score: np.ndarray) -> float:
    """AP = sum_n (R_n - R_{n-1}) P_n over the distinct score thresholds, descending
    (sklearn.metrics.average_precision_score semantics, ties share one threshold)."""
    y_true = np.asarray(y_true).astype(np.float64)
    y_score = np.asarray(y_score, dtype=np.float64)
    order = np.argsort(-y_score, kind="mergesort")
    y_true, y_score = y_true[order], y_score[order]
    last = np.r_[np.nonzero(np.diff(y_score))[0], y_true.size - 1]     # last index of every tie group
    tp = np.cumsum(y_true)[last]
    npos = y_true.sum()
    if npos == 0:
        return float("nan")
    precision = tp / (last + 1.0)
    recall = tp / npos
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def rank_average(x: np.ndarray) -> np.ndarray:
    """1-based ranks, ties get the average rank (scipy.stats.rankdata default)."""
    x = np.asarray(x, dtype=np.float64)
    order = np.argsort(x, kind="mergesort")
    xs = x[order]
    starts = np.r_[0, np.nonzero(np.diff(xs))[0] + 1]
    ends = np.r_[starts[1:], xs.size]
    ranks = np.empty(xs.size, dtype=np.float64)
    for s, e in zip(starts, ends):
        ranks[order[s:e]] = 0.5 * (s + e - 1) + 1.0
    return ranks


def spearman_rho(a, b) -> float:
    ra, rb = rank_average(a), rank_average(b)
    ra, rb = ra - ra.mean(), rb - rb.mean()
    den = np.sqrt((ra * ra).sum() * (rb * rb).sum())
    return float((ra * rb).sum() / den) if den > 0 else float("nan")


def kendall_tau_b(a, b) -> float:
    """tau-b = (P - Q) / sqrt((n0 - n1)(n0 - n2)), the scipy.stats.kendalltau default."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    n = a.size
    iu = np.triu_indices(n, k=1)
    sa = np.sign(a[:, None] - a[None, :])[iu]
    sb = np.sign(b[:, None] - b[None, :])[iu]
    n0 = n * (n - 1) / 2.0
    n1, n2 = float((sa == 0).sum()), float((sb == 0).sum())
    den = np.sqrt((n0 - n1) * (n0 - n2))
    return float((sa * sb).sum() / den) if den > 0 else float("nan")


def f1_binary(y_true, y_pred) -> float:
    y_true, y_pred = np.asarray(y_true).astype(bool), np.asarray(y_pred).astype(bool)
    tp = float((y_true & y_pred).sum())
    den = 2 * tp + float((~y_true & y_pred).sum()) + float((y_true & ~y_pred).sum())
    return 2 * tp / den if den > 0 else 0.0


# ---- TVSum (test/tvsum/tvsum_utils.py) ----------------------------------------------------------------
def binarize_gt(gt_scores, rho):
    gt_scores = np.asarray(gt_scores)
    k = max(1, int(rho * len(gt_scores)))
    return (gt_scores >= np.sort(gt_scores)[-k]).astype(int)


def map_at_rho(gt_scores, pred_scores, rho):
    return average_precision(binarize_gt(gt_scores, rho), pred_scores)


def top_k_ap(gt_binary, sorted_indices, k=5):
    """trapezoidal AP over the first k predictions (tvsum_utils.py:202-220)."""
    sel = np.asarray(gt_binary)[sorted_indices][:k]
    num_gt = sel.sum()
    if num_gt == 0:
        return 0.0
    hits, ap, rec_prev, prec_prev = 0, 0.0, 0.0, 1.0
    for j, lab in enumerate(sel):
        hits += lab
        rec, prec = hits / num_gt, hits / (j + 1)
        ap += (rec - rec_prev) * (prec + prec_prev) / 2.0
        rec_prev, prec_prev = rec, prec
    return float(ap)


def evaluate_top5_map_tvsum(gt_dict, pred_dict, rho=0.5, top_k=5):
    aps = []
    for vid in gt_dict:
        gt, pr = np.array(gt_dict[vid]), np.array(pred_dict[vid])
        assert len(gt) == len(pr), f"Length mismatch for video {vid}"
        aps.append(top_k_ap(binarize_gt(gt, rho), np.argsort(pr)[::-1], k=top_k))
    return float(np.mean(aps))


def evaluate_tvsum(gt_dict, pred_dict):
    """-> (mAP50, mAP15, top5_mAP, spearman, kendall)  (tvsum_utils.py:37-69)"""
    m50, m15, ken, spe = [], [], [], []
    for vid, gt in gt_dict.items():
        pr = pred_dict[vid]
        if len(gt) != len(pr):
            continue
        if len(gt) > 1:
            spe.append(spearman_rho(gt, pr))
            ken.append(kendall_tau_b(gt, pr))
        else:
            spe.append(0.0)
            ken.append(0.0)
        m50.append(map_at_rho(gt, pr, 0.50))
        m15.append(map_at_rho(gt, pr, 0.15))
    return (float(np.mean(m50)), float(np.mean(m15)), evaluate_top5_map_tvsum(gt_dict, pred_dict),
            float(np.mean(spe)), float(np.mean(ken)))


def f1_at_rho(gt_scores, pred_scores, rho):
    gt_scores, pred_scores = np.asarray(gt_scores), np.asarray(pred_scores)
    n = len(gt_scores)
    k = max(1, int(rho * n))
    gt_bin = (gt_scores >= np.sort(gt_scores)[-k]).astype(int)
    pred_bin = np.zeros(n, dtype=int)
    pred_bin[np.argsort(pred_scores)[-k:]] = 1
    return f1_binary(gt_bin, pred_bin)


def evaluate_f1(gt_dict, pred_dict, rho=0.15):
    return float(np.mean([f1_at_rho(gt, pred_dict[v], rho) for v, gt in gt_dict.items()]))


# ---- Mr.HiSum (test/hisum/hisum_eval.py) ------------------------------------------------------------
def _shots(scores, shot_length=1, fps=1):
    seg = shot_length * fps
    scores = np.asarray(scores, dtype=np.float64)
    return np.array([scores[i:i + seg].mean() for i in range(0, len(scores), seg)])


def hisum_mean_average_precision(gt_dict, pred_dict, rho=0.5):
    aps = []
    for vid in gt_dict:
        pred_seg, gt_seg = _shots(pred_dict[vid]), _shots(gt_dict[vid])
        k = max(1, int(rho * len(pred_seg)))
        labels = np.zeros(len(pred_seg))
        labels[np.argsort(gt_seg)[-k:]] = 1
        ap = average_precision(labels, pred_seg)
        if not np.isnan(ap):
            aps.append(ap)
    return float(np.mean(aps))


def hisum_f1_score_summarization(gt_dict, pred_dict, budget=0.15, shot_length=1):
    f1s = []
    for vid in gt_dict:
        gt, pr = np.asarray(gt_dict[vid], dtype=np.float64), np.asarray(pred_dict[vid], dtype=np.float64)
        n = len(gt)
        bounds = [(i, min(i + shot_length, n)) for i in range(0, n, shot_length)]
        shot_scores = [pr[s:e].mean() for s, e in bounds]
        total, acc = int(budget * n), 0
        sel = np.zeros(n, dtype=bool)
        for idx in np.argsort(shot_scores)[::-1]:
            s, e = bounds[idx]
            if acc + (e - s) <= total:
                sel[s:e] = True
                acc += e - s
            if acc >= total:
                break
        gt_sel = gt >= np.percentile(gt, 100 * (1 - budget))
        f1s.append(round(f1_binary(gt_sel, sel), 2))
    return float(np.mean(f1s))


def hisum_evaluate_scores(gt_dict, pred_dict, spearman_kendall=False):
    out = {}
    if spearman_kendall:
        ken, spe = [], []
        for vid in gt_dict:
            gt, pr = gt_dict[vid], pred_dict[vid]
            if len(gt) != len(pr):
                continue
            if len(gt) > 1:
                spe.append(spearman_rho(gt, pr))
                ken.append(kendall_tau_b(gt, pr))
            else:
                spe.append(0.0)
                ken.append(0.0)
        out["kendall"], out["spearman"] = float(np.mean(ken)), float(np.mean(spe))
    out["mAP@50"] = hisum_mean_average_precision(gt_dict, pred_dict, rho=.5)
    out["mAP@15"] = hisum_mean_average_precision(gt_dict, pred_dict, rho=.15)
    out["f1"] = hisum_f1_score_summarization(gt_dict, pred_dict)
    return out


# ---- highlight selection (test/highlight_generator.py:8-37) ------------------------------------------
def knapsack_selection(frames_with_index: List[dict], max_duration: int, weight, alpha, beta, epsilon) -> set:
    """0/1 knapsack with unit costs over value = info*alpha + rel*beta + unc*epsilon, including the
    reference's back-tracking tie rule (a frame is taken iff dp[i][c] != dp[i-1][c])."""
    n = len(frames_with_index)
    vals = np.array([f["informative_score"] * alpha + f["relevance_score"] * beta + f["uncertainty_score"] * epsilon
                     for f in frames_with_index], dtype=np.float64)
    dp = np.zeros((n + 1, max_duration + 1), dtype=np.float64)
    for i in range(1, n + 1):
        dp[i] = dp[i - 1]
        if max_duration >= 1:
            dp[i, 1:] = np.maximum(dp[i - 1, 1:], dp[i - 1, :-1] + vals[i - 1])
    chosen, cap = [], max_duration
    for i in range(n, 0, -1):
        if dp[i, cap] != dp[i - 1, cap]:
            chosen.append(frames_with_index[i - 1])
            cap -= 1
    return set(f["idx"] for f in chosen)
