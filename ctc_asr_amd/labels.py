"""Alphabet of the CTC acoustic model.

Mirrors the contract of the reference's ``asr/labels.py:11-59``:

* 27 printable symbols ``' a..z'`` map to ids 1..27,
* id 0 is the padding / "unused" slot and renders as the empty string,
* id 28 (= ``num_classes() - 1``) is the CTC blank and has no character.
"""

ALPHABET = ' abcdefghijklmnopqrstuvwxyz'
PAD_ID = 0
BLANK_ID = len(ALPHABET) + 1  # 28

_CHAR_TO_ID = {ch: idx + 1 for idx, ch in enumerate(ALPHABET)}
_ID_TO_CHAR = {idx + 1: ch for idx, ch in enumerate(ALPHABET)}
_ID_TO_CHAR[PAD_ID] = ''


def num_classes():
    """29 = 27 characters + the unused id 0 + the CTC blank (``asr/labels.py:52-59``)."""
    return len(ALPHABET) + 2


def ctoi(char):
    """Character -> integer label; ``ValueError`` for anything outside ``' a-z'``
    (``asr/labels.py:21-36``)."""
    if char not in ALPHABET:
        raise ValueError('Invalid input character \'{}\'.'.format(char))
    if len(char) != 1:
        raise ValueError('"{}" is not a valid character.'.format(char))
    return _CHAR_TO_ID[char]


def itoc(integer):
    """Integer label -> character. ``0 -> ''``; out-of-range raises ``ValueError`` and the
    blank id raises ``KeyError`` exactly like the reference (``asr/labels.py:39-49``)."""
    if not 0 <= integer < num_classes():
        raise ValueError('Integer label ({}) out of range.'.format(integer))
    return _ID_TO_CHAR[int(integer)]


def encode(text):
    """``[ctoi(c) for c in text]`` as used by the input generator
    (``asr/input_functions.py:150``)."""
    return [ctoi(c) for c in text]


def decode(ids):
    """Join ``itoc`` over a row of integer labels (``asr/util/metrics.py:31-32``)."""
    return ''.join(itoc(int(i)) for i in ids)
