"""Corpus word error rate, the metric of inference/ger.py:98,112 (jiwer 3.0.2 via evaluate 0.4.0 in
the reference; neither is installed here, so this is an own implementation of the published
definition: WER = (S + D + I) / N_ref over the whole corpus, words = whitespace-separated tokens
after stripping and collapsing whitespace — jiwer's default transform).  WER parity with jiwer is
UNPINNED; the known-answer cases live in tests/test_host_logic.py."""
from __future__ import annotations

from typing import Dict, Sequence, Tuple


def _words(s: str):
    return s.split()


def edit_counts(ref: Sequence[str], hyp: Sequence[str]) -> Tuple[int, int, int]:
    """(substitutions, deletions, insertions) of a minimum-cost word alignment."""
    n, m = len(ref), len(hyp)
    # dp over (cost, S, D, I); ties resolved S < D < I like the usual Levenshtein back-trace
    prev = [(j, 0, 0, j) for j in range(m + 1)]
    for i in range(1, n + 1):
        cur = [(i, 0, i, 0)] + [None] * m
        for j in range(1, m + 1):
            if ref[i - 1] == hyp[j - 1]:
                cur[j] = prev[j - 1]
                continue
            c_s, c_d, c_i = prev[j - 1], prev[j], cur[j - 1]
            best = min((c_s[0], 0), (c_d[0], 1), (c_i[0], 2))
            if best[1] == 0:
                cur[j] = (c_s[0] + 1, c_s[1] + 1, c_s[2], c_s[3])
            elif best[1] == 1:
                cur[j] = (c_d[0] + 1, c_d[1], c_d[2] + 1, c_d[3])
            else:
                cur[j] = (c_i[0] + 1, c_i[1], c_i[2], c_i[3] + 1)
        prev = cur
    _, s, d, i = prev[m]
    return s, d, i


def wer_counts(predictions: Sequence[str], references: Sequence[str]) -> Dict[str, int]:
    """Additive counters (errors, reference words, exact matches, utterances) — what ranks all-reduce."""
    err = nref = exact = 0
    for p, r in zip(predictions, references):
        rw, pw = _words(r), _words(p)
        s, d, i = edit_counts(rw, pw)
        err += s + d + i
        nref += len(rw)
        exact += int(p == r)
    return {"errors": err, "ref_words": nref, "exact": exact, "n": len(references)}


def wer(predictions: Sequence[str], references: Sequence[str]) -> float:
    c = wer_counts(predictions, references)
    return c["errors"] / c["ref_words"] if c["ref_words"] else 0.0


_PUNCT = str.maketrans("", "", ".,-?'")


def post_normalize(s: str) -> str:
    """The 'post' normalisation of inference/ger.py:108-109: lower-case, drop . , - ? '"""
    return s.lower().translate(_PUNCT)
