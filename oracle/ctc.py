"""ORACLE (test infrastructure, not product code): CTC loss / gradient / decoding on the CPU.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  PARITY UNPINNED by the reference (no tests, TensorFlow not runnable here); pinned
instead by the recalled TensorFlow known-answer vectors in ``tests/golden/ctc_kat.json``, by
brute-force path enumeration and by ``torch.nn.functional.ctc_loss`` on the CPU.

Restates, in float64 numpy:

* ``tf.nn.ctc_loss(time_major=True, ctc_merge_repeated=True,
  preprocess_collapse_repeated=False)`` as called by ``CTCModel.loss_fn``
  (``asr/model.py:259-264``) — TensorFlow 1.12 ``ctc_loss_calculator``: blank = C-1, log-space
  alpha/beta, gradient w.r.t. the *logits*, zero gradient beyond ``seq_len``, error for
  infeasible alignments (``ignore_longer_outputs_than_inputs=False``).
* ``tf.nn.ctc_beam_search_decoder(top_paths=1, merge_repeated=False)`` as called by
  ``CTCModel.decode_fn`` (``asr/model.py:292-296``) — TensorFlow 1.12 ``ctc_beam_search.h``.
* ``ctc_greedy_decoder`` semantics (argmax, collapse repeats, drop blank) — only mentioned in
  comments of the reference (``asr/model.py:290,298``) but required by BASELINE.json.
"""

import itertools

import numpy as np

NEG_INF = -np.inf


class InfeasibleAlignment(ValueError):
    """TensorFlow's InvalidArgumentError 'Not enough time for target transition sequence'."""


def logsumexp2(a, b):
    if a == NEG_INF:
        return b
    if b == NEG_INF:
        return a
    hi, lo = (a, b) if a > b else (b, a)
    return hi + np.log1p(np.exp(lo - hi))


def log_softmax(x):
    """Row-wise log-softmax over the last axis (max-subtracted), float64."""
    x = np.asarray(x, dtype=np.float64)
    shifted = x - np.max(x, axis=-1, keepdims=True)
    return shifted - np.log(np.sum(np.exp(shifted), axis=-1, keepdims=True))


def required_time(label):
    """Minimum number of frames for ``label``: L + number of adjacent repeats."""
    label = list(label)
    return len(label) + sum(1 for a, b in zip(label, label[1:]) if a == b)


def extended_labels(label, blank):
    ext = [blank]
    for sym in label:
        ext.extend([int(sym), blank])
    return ext


def ctc_loss_single(logits, label, blank=None):
    """One utterance.  ``logits`` [T, C] raw activations, ``label`` list of ints in [0, C-2].

    Returns (loss = -ln p(label | x), grad [T, C] w.r.t. the logits).
    """
    logits = np.asarray(logits, dtype=np.float64)
    num_steps, num_classes = logits.shape
    blank = num_classes - 1 if blank is None else blank
    label = [int(v) for v in label]
    if any(v < 0 or v >= num_classes or v == blank for v in label):
        raise ValueError('label id out of range')
    if num_steps < required_time(label):
        raise InfeasibleAlignment(
            'Not enough time for target transition sequence (required: {}, available: {})'
            .format(required_time(label), num_steps))
    logp = log_softmax(logits)
    ext = extended_labels(label, blank)
    size = len(ext)

    alpha = np.full((num_steps, size), NEG_INF)
    alpha[0, 0] = logp[0, blank]
    if size > 1:
        alpha[0, 1] = logp[0, ext[1]]
    for t in range(1, num_steps):
        for u in range(size):
            acc = alpha[t - 1, u]
            if u >= 1:
                acc = logsumexp2(acc, alpha[t - 1, u - 1])
            if u >= 2 and ext[u] != blank and ext[u] != ext[u - 2]:
                acc = logsumexp2(acc, alpha[t - 1, u - 2])
            alpha[t, u] = acc + logp[t, ext[u]] if acc != NEG_INF else NEG_INF

    # beta(t, u) excludes the emission at t (TensorFlow convention): alpha + beta = joint.
    beta = np.full((num_steps, size), NEG_INF)
    beta[num_steps - 1, size - 1] = 0.0
    if size > 1:
        beta[num_steps - 1, size - 2] = 0.0
    for t in range(num_steps - 2, -1, -1):
        for u in range(size):
            acc = beta[t + 1, u] + logp[t + 1, ext[u]]
            if u + 1 < size:
                acc = logsumexp2(acc, beta[t + 1, u + 1] + logp[t + 1, ext[u + 1]])
            if u + 2 < size and ext[u + 2] != blank and ext[u + 2] != ext[u]:
                acc = logsumexp2(acc, beta[t + 1, u + 2] + logp[t + 1, ext[u + 2]])
            beta[t, u] = acc

    log_pzx = alpha[num_steps - 1, size - 1]
    if size > 1:
        log_pzx = logsumexp2(log_pzx, alpha[num_steps - 1, size - 2])

    grad = np.exp(logp)
    if log_pzx != NEG_INF:
        for t in range(num_steps):
            occupancy = np.full(num_classes, NEG_INF)
            for u in range(size):
                occupancy[ext[u]] = logsumexp2(occupancy[ext[u]], alpha[t, u] + beta[t, u])
            grad[t] -= np.exp(occupancy - log_pzx)
    return -log_pzx, grad


def ctc_loss(logits, labels, seq_len, blank=None):
    """Batch form matching ``tf.nn.ctc_loss``: ``logits`` [T, B, C] time-major, ``labels`` list of
    B label lists, ``seq_len`` [B].  Returns (loss f64[B], grad f64[T, B, C]); gradient rows at
    ``t >= seq_len[b]`` are zero.  Raises `InfeasibleAlignment` like TensorFlow."""
    logits = np.asarray(logits, dtype=np.float64)
    num_steps, batch, _ = logits.shape
    losses = np.zeros(batch)
    grads = np.zeros_like(logits)
    for b in range(batch):
        length = int(seq_len[b])
        if length > num_steps:
            raise ValueError('sequence_length(b) <= max_time violated')
        losses[b], grads[:length, b] = ctc_loss_single(logits[:length, b], labels[b], blank)
    return losses, grads


def dense_to_label_lists(dense):
    """``tfc.layers.dense_to_sparse`` (``asr/model.py:71``): drop the 0 (pad/eos) entries."""
    return [[int(v) for v in row if int(v) != 0] for row in np.asarray(dense)]


# ------------------------------------------------------------------------------------------
# Brute force (tiny cases only): exact p(label | x) by enumerating all C^T paths.
# ------------------------------------------------------------------------------------------
def collapse_path(path, blank):
    out, prev = [], None
    for sym in path:
        if sym != prev and sym != blank:
            out.append(int(sym))
        prev = sym
    return tuple(out)


def brute_force_posteriors(logits, blank=None):
    """dict: label tuple -> probability, from all C^T paths of one utterance ([T, C] logits)."""
    logp = log_softmax(logits)
    num_steps, num_classes = logp.shape
    blank = num_classes - 1 if blank is None else blank
    table = {}
    for path in itertools.product(range(num_classes), repeat=num_steps):
        prob = float(np.exp(sum(logp[t, s] for t, s in enumerate(path))))
        key = collapse_path(path, blank)
        table[key] = table.get(key, 0.0) + prob
    return table


# ------------------------------------------------------------------------------------------
# Decoding
# ------------------------------------------------------------------------------------------
def greedy_decode(logits, seq_len, blank=None):
    """Argmax per frame (first maximum on ties), merge repeats, drop blanks.  Returns a list of
    B integer lists."""
    logits = np.asarray(logits)
    _, batch, num_classes = logits.shape
    blank = num_classes - 1 if blank is None else blank
    out = []
    for b in range(batch):
        best = np.argmax(logits[:int(seq_len[b]), b], axis=-1)
        out.append(list(collapse_path(best.tolist(), blank)))
    return out


class _Beam:
    """One prefix of the TensorFlow beam search tree (``BeamEntry``)."""
    __slots__ = ('parent', 'label', 'children', 'old', 'new', 'serial')

    def __init__(self, parent, label, serial):
        self.parent, self.label, self.serial = parent, label, serial
        self.children = {}
        self.old = [NEG_INF, NEG_INF, NEG_INF]  # total, blank, label
        self.new = [NEG_INF, NEG_INF, NEG_INF]

    def active(self):
        return self.new[0] != NEG_INF

    def path(self):
        labels, node = [], self
        while node.parent is not None:
            labels.append(node.label)
            node = node.parent
        return labels[::-1]


def _f32(x):
    return np.float32(x)


def _lse32(a, b):
    """float32 LogSumExp as in TensorFlow's ``ctc_loss_util.h`` (log1p form)."""
    if a == NEG_INF:
        return b
    if b == NEG_INF:
        return a
    hi, lo = (a, b) if a > b else (b, a)
    return _f32(hi + _f32(np.log1p(_f32(np.exp(_f32(lo - hi))))))


def beam_search_decode_single(logits, beam_width, blank=None, normalization='max'):
    """TensorFlow ``CTCBeamSearchDecoder`` (top path, ``merge_repeated=False``) for one utterance.

    float32 arithmetic like TensorFlow.  ``normalization='max'`` subtracts the per-frame maximum
    (TensorFlow 1.12, the version the reference was tested with); ``'log_softmax'`` subtracts
    max + log-sum-exp (TensorFlow >= 1.14).  Ties in ``total`` are broken towards the prefix
    that entered the beam first (TensorFlow leaves ties to ``gtl::TopN`` / ``std::sort``).
    Returns (labels list, log-probability of the top path = its ``new.total``).
    """
    logits = np.asarray(logits, dtype=np.float32)
    num_steps, num_classes = logits.shape
    blank = num_classes - 1 if blank is None else blank
    serial = itertools.count()
    root = _Beam(None, -1, next(serial))
    root.new = [_f32(0.0), _f32(0.0), NEG_INF]
    leaves = [root]

    def order(nodes):
        return sorted(nodes, key=lambda n: (-n.new[0], n.serial))

    for t in range(num_steps):
        frame = logits[t]
        peak = np.max(frame)
        if normalization == 'log_softmax':
            offset = _f32(peak + _f32(np.log(np.sum(np.exp(frame - peak), dtype=np.float32))))
        else:
            offset = peak
        x = (frame - offset).astype(np.float32)

        branches = order(leaves)
        leaves = []
        for b in branches:
            b.old = list(b.new)
        for b in branches:
            if b.parent is not None:
                if b.parent.active():
                    prev = b.parent.old[1] if b.label == b.parent.label else b.parent.old[0]
                    b.new[2] = _lse32(b.new[2], prev)
                b.new[2] = _f32(b.new[2] + x[b.label])
            b.new[1] = _f32(b.old[0] + x[blank])
            b.new[0] = _lse32(b.new[1], b.new[2])
            leaves.append(b)

        def bottom():
            # lowest total; among equal totals the youngest tree node is evicted first
            return min(leaves, key=lambda n: (n.new[0], -n.serial))

        def is_candidate(total):
            if total == NEG_INF:
                return False
            return len(leaves) < beam_width or total > bottom().new[0]

        for b in branches:
            if not is_candidate(b.old[0]):
                continue
            for c_label in range(num_classes):
                if c_label == blank:
                    continue
                child = b.children.get(c_label)
                if child is None:
                    child = _Beam(b, c_label, next(serial))
                    b.children[c_label] = child
                if child.active():
                    continue
                child.new[1] = NEG_INF
                prev = b.old[1] if c_label == b.label else b.old[0]
                child.new[2] = _f32(x[c_label] + prev)
                child.new[0] = child.new[2]
                if is_candidate(child.new[0]):
                    if len(leaves) == beam_width:
                        worst = bottom()
                        worst.new = [NEG_INF, NEG_INF, NEG_INF]
                        leaves.remove(worst)
                    leaves.append(child)
                else:
                    child.old = [NEG_INF, NEG_INF, NEG_INF]
                    child.new = [NEG_INF, NEG_INF, NEG_INF]

    best = order(leaves)[0]
    return best.path(), float(best.new[0])


def beam_search_decode(logits, seq_len, beam_width, blank=None, normalization='max'):
    """Batch form: ``logits`` [T, B, C]; returns (list of B label lists, logp f32[B])."""
    logits = np.asarray(logits, dtype=np.float32)
    batch = logits.shape[1]
    paths, scores = [], np.zeros(batch, dtype=np.float32)
    for b in range(batch):
        path, score = beam_search_decode_single(logits[:int(seq_len[b]), b], beam_width, blank,
                                                normalization)
        paths.append(path)
        scores[b] = score
    return paths, scores
