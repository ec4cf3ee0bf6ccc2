"""Corpus BLEU-4 over token sequences -- SURVEY.md section 8(f) row 1.

The reference scores captions with `torchmetrics.BLEUScore()` (ref CLIP-DDPM.py:604-631: n_gram=4, no smoothing, uniform
weights, one score per validation batch, averaged over batches) and `torchtext.data.metrics.bleu_score` in COCO_BLEU.py:263
(same definition).  Neither package is installed here, so this restates the published definition (Papineni et al. 2002,
corpus level): clipped n-gram matches and candidate n-gram totals are summed over the whole corpus per order n = 1..4,
precision_n = matches_n / total_n, BLEU = BP * exp(mean_n log precision_n), BP = 1 if c > r else exp(1 - r/c) with r the
sum of the reference lengths closest to each candidate's length.  Ties in |len(ref) - len(cand)| go to the FIRST such reference
in list order: torchmetrics' `_bleu_score_update` takes `target_len_list[target_len_diff.index(min(target_len_diff))]`, and
`list.index` returns the first minimum (NLTK's corpus_bleu would take the shorter one -- the two differ exactly when a longer
reference precedes an equally distant shorter one, which tests/test_next_rows.py pins).  Any zero precision gives 0 (no
smoothing).  PARITY UNPINNED against torchmetrics itself (absent here, no network); held by hand-computed cases, a brute-force
restatement and the two libraries' documented examples (0.7598 / 0.8408964276313782) in tests/test_next_rows.py.  The string side of the
loop (tokenizer call, `tokenizer.decode`) is in wordpiece.py and IS pinned against transformers' tokenizer.
"""
from __future__ import annotations

import math
from collections import Counter
from typing import Iterable, Sequence


def _tokens(x):
    return x.split() if isinstance(x, str) else [t.item() if hasattr(t, "item") else t for t in x]


def _ngrams(tok: Sequence, n: int) -> Counter:
    return Counter(tuple(tok[i:i + n]) for i in range(len(tok) - n + 1))


def corpus_bleu(candidates: Iterable, references: Iterable[Iterable], n_gram: int = 4) -> float:
    """candidates: list of token sequences (or whitespace-separated strings); references: per candidate, a list of references."""
    matches = [0] * n_gram
    totals = [0] * n_gram
    c_len = r_len = 0
    for cand, refs in zip(candidates, references):
        cand = _tokens(cand)
        refs = [_tokens(r) for r in refs]
        c_len += len(cand)
        diffs = [abs(len(r) - len(cand)) for r in refs]
        r_len += len(refs[diffs.index(min(diffs))])          # first closest reference in list order (torchmetrics' rule)
        for n in range(1, n_gram + 1):
            cg = _ngrams(cand, n)
            if not cg:
                continue
            best = Counter()
            for r in refs:
                for g, c in _ngrams(r, n).items():
                    if c > best[g]:
                        best[g] = c
            matches[n - 1] += sum(min(c, best[g]) for g, c in cg.items())
            totals[n - 1] += sum(cg.values())
    if c_len == 0 or min(totals) == 0 or min(matches) == 0:
        return 0.0
    log_p = sum(math.log(m / t) for m, t in zip(matches, totals)) / n_gram
    bp = 1.0 if c_len > r_len else math.exp(1.0 - r_len / c_len)
    return bp * math.exp(log_p)


def batch_averaged_bleu(batches) -> float:
    """The reference's aggregation (ref :611-631): BLEU of each validation batch, averaged over batches."""
    scores = [corpus_bleu(c, r) for c, r in batches]
    return sum(scores) / max(len(scores), 1)
