"""Pins the CTC oracle: recalled TensorFlow known-answer vectors, brute-force path enumeration,
torch.nn.functional.ctc_loss, and the C restatement against the numpy one."""

import json
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import cref
from oracle import ctc as octc

KAT = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'ctc_kat.json')))


def test_tensorflow_known_answers_loss_and_gradient():
    for case in KAT['loss_cases']:
        logits = np.log(np.array(case['probs']))
        loss, grad = octc.ctc_loss_single(logits, case['targets'])
        assert loss == pytest.approx(case['loss'], abs=2e-5)
        assert np.abs(grad - np.array(case['grad'])).max() < 2e-6
        c_loss, c_grad, status = cref.ctc_loss(logits[:, None, :], [case['targets']],
                                               [logits.shape[0]])
        assert status[0] == 0 and c_loss[0] == pytest.approx(case['loss'], abs=2e-5)
        assert np.abs(c_grad[:, 0] - np.array(case['grad'])).max() < 2e-6


def test_decode_known_answers():
    case = KAT['decode_case']
    logits = np.log(np.array(case['probs']))
    assert octc.greedy_decode(logits[:, None, :], [5])[0] == case['greedy']
    assert octc.beam_search_decode_single(logits, 64)[0] == case['beam_wide']
    assert octc.beam_search_decode_single(logits, 2)[0] == case['beam_width_2']   # TF's own test
    post = octc.brute_force_posteriors(logits)
    ranked = sorted(post.items(), key=lambda kv: -kv[1])[:5]
    for (labels, prob), (ref_labels, ref_prob) in zip(ranked, case['top5']):
        assert list(labels) == ref_labels and prob == pytest.approx(ref_prob, rel=2e-3)


@settings(max_examples=25, deadline=None)
@given(st.integers(2, 4), st.integers(1, 5), st.integers(0, 2 ** 31 - 1))
def test_loss_equals_brute_force(classes, steps, seed):
    rng = np.random.default_rng(seed)
    logits = rng.normal(size=(steps, classes)) * 2
    post = octc.brute_force_posteriors(logits)
    assert sum(post.values()) == pytest.approx(1.0)
    for label, prob in post.items():
        loss, _ = octc.ctc_loss_single(logits, list(label))
        assert loss == pytest.approx(-np.log(prob), abs=1e-9)
    # an infeasible label raises like TensorFlow
    with pytest.raises(octc.InfeasibleAlignment):
        octc.ctc_loss_single(logits, [0] * (steps + 1))


@settings(max_examples=25, deadline=None)
@given(st.integers(2, 4), st.integers(1, 5), st.integers(0, 2 ** 31 - 1))
def test_unpruned_beam_search_finds_the_most_probable_labelling(classes, steps, seed):
    rng = np.random.default_rng(seed)
    logits = rng.normal(size=(steps, classes)) * 2
    post = octc.brute_force_posteriors(logits)
    best_label, best_prob = max(post.items(), key=lambda kv: kv[1])
    for norm in ('max', 'log_softmax'):
        path, logp = octc.beam_search_decode_single(logits, 10 ** 6, normalization=norm)
        assert post[tuple(path)] == pytest.approx(best_prob, rel=1e-5)
        if norm == 'log_softmax':
            assert np.exp(logp) == pytest.approx(best_prob, rel=1e-4)
    assert tuple(octc.greedy_decode(logits[:, None, :], [steps])[0]) in post


def test_gradient_matches_torch_and_is_zero_beyond_seq_len():
    rng = np.random.default_rng(7)
    steps, batch, classes = 30, 4, 29
    logits = rng.normal(size=(steps, batch, classes))
    labels = [[1, 1, 2], [3], [], list(rng.integers(0, 28, size=10))]
    seq_len = [30, 12, 7, 25]
    loss, grad = octc.ctc_loss(logits, labels, seq_len)
    x = torch.tensor(logits, requires_grad=True)
    ref = torch.nn.functional.ctc_loss(
        torch.log_softmax(x, -1), torch.tensor([v for r in labels for v in r]),
        torch.tensor(seq_len), torch.tensor([len(r) for r in labels]), blank=classes - 1,
        reduction='none')
    ref.sum().backward()
    assert np.abs(loss - ref.detach().numpy()).max() < 1e-9
    assert np.abs(grad - x.grad.numpy()).max() < 1e-9
    for b, length in enumerate(seq_len):
        assert np.abs(grad[length:, b]).max(initial=0.0) == 0.0
    # C restatement agrees, including status codes
    c_loss, c_grad, status = cref.ctc_loss(logits, labels, seq_len)
    assert (status == 0).all() and np.abs(c_loss - loss).max() < 1e-4   # float32 input
    assert np.abs(c_grad - grad).max() < 1e-5
    _, _, status = cref.ctc_loss(logits[:4], [[0, 0, 0], [1], [28], [2]], [4, 4, 4, 4])
    assert status.tolist() == [1, 0, 2, 0]


def test_c_beam_search_equals_numpy_beam_search():
    rng = np.random.default_rng(11)
    logits = (rng.normal(size=(25, 5, 7)) * 2).astype(np.float32)
    seq_len = [25, 9, 17, 1, 25]
    for width in (1, 2, 3, 8, 50):
        for norm in ('max', 'log_softmax'):
            p1, s1 = octc.beam_search_decode(logits, seq_len, width, normalization=norm)
            p2, s2 = cref.beam_search_decode(logits, seq_len, width, normalization=norm)
            assert p1 == p2 and np.allclose(s1, s2, atol=1e-5)


def test_dense_to_label_lists_drops_padding_zeros():
    assert octc.dense_to_label_lists(np.array([[3, 4, 0, 0], [0, 0, 0, 0], [1, 0, 2, 0]])) == \
        [[3, 4], [], [1, 2]]
