"""The seeded cases of tests/golden/hf_rules_golden.npz (shared by the generator, tests/golden/make_golden.py, and the test)."""
import numpy as np


def rule_cases(beg, eot):
    """(history, trial) pairs: empty history, lone / paired / repeated timestamps, text after timestamps, long text; trial shapes the raw logits"""
    hists = [[], [beg + 10], [beg + 10, 500], [500, beg + 20], [beg + 5, beg + 9], [100, 200, 300], [beg + 40, 7, 8, beg + 80, beg + 80],
             [beg], [beg, 17], [beg + 3, 21, 22, beg + 30], [beg + 3, 21, 22, beg + 30, beg + 30, 9], [eot - 1, eot - 2]]
    return [(h, t) for h in hists for t in range(4)]


def rule_logits(rng, n_vocab, beg, eot, trial):
    raw = (9.0 * rng.standard_normal(n_vocab)).astype(np.float32)
    if trial == 1:
        raw[beg:] += 6.0        # probability mass onto timestamps -> the "timestamp mass beats every text token" branch
    if trial == 2:
        raw[eot] += 60.0
    if trial == 3:
        raw[beg:] -= 200.0      # timestamp probabilities underflow
    return raw
