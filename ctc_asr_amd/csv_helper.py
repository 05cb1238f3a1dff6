"""Bucket boundaries from the corpus CSV (mirror of ``asr/util/csv_helper.py:9-38``)."""

import csv
import os

from ctc_asr_amd.params import CSV_DELIMITER, CSV_FIELDNAMES, CSV_HEADER_LENGTH, WIN_STEP


def read_csv_rows(csv_path):
    """All rows of a ``path;label;length`` CSV as dicts, header row included (the reference opens
    the file with explicit ``fieldnames`` so the header is an ordinary first row)."""
    if not (os.path.exists(csv_path) and os.path.isfile(csv_path)):
        raise AssertionError('CSV file "{}" does not exist.'.format(csv_path))
    with open(csv_path, 'r', encoding='utf-8') as handle:
        return list(csv.DictReader(handle, delimiter=CSV_DELIMITER, fieldnames=CSV_FIELDNAMES))


def get_bucket_boundaries(csv_path, num_buckets):
    """Bucket boundaries (in feature frames) taken at every ``len // num_buckets``-th example.

    Follows the reference literally: only the header row is dropped (``[1:]``, unlike the input
    generator's ``[1:-1]``), a length of ``s`` seconds counts as ``int(s / WIN_STEP)`` frames, and
    the picks are de-duplicated and sorted — true quantiles only for a length-sorted CSV.
    Like the reference, a CSV with fewer examples than buckets raises ``ValueError``
    (``range()`` step of zero).
    """
    rows = read_csv_rows(csv_path)[1:]
    lengths = [int(float(row[CSV_HEADER_LENGTH]) / WIN_STEP) for row in rows]
    step = len(lengths) // num_buckets
    return sorted({lengths[i] for i in range(step, len(lengths), step)})
