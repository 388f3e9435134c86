"""Grouping half of /root/reference/src/scanner.rs (lines 1588-1832) on top of the C ABI.

    PDQ_MIN_QUALITY, is_low_pdq_quality            scanner.rs:1588-1594
    group_with_pdqhash / group_files_generic       scanner.rs:1640-1832 up to the union-find
File-name logic after the union-find (merge_groups_by_stem, process_raw_groups) stays with the caller.
"""
import numpy as np

from . import _lib
from .engine import default_engine

PDQ_MIN_QUALITY = 50


def is_low_pdq_quality(quality):
    return bool(_lib.load().rph_is_low_pdq_quality(-1 if quality is None else int(quality)))


def stored_quality(q):
    """scanner.rs:1416-1417: (q * 100).round().clamp(0, 100) as u16 (round half away from zero)."""
    v = np.float32(q) * np.float32(100.0)
    return int(min(100, max(0, np.floor(v + np.float32(0.5)))))


def group_with_pdqhash(hashes, similarity, coefficients=None, has_features=None, quality=None, engine=None):
    """Returns (groups, comparison_count): connected components (> 1 member, members ascending,
    groups by first member) of the edge set of group_files_generic::<[u8;32], PdqStrategy>."""
    q = None if quality is None else np.array([-1 if x is None else int(x) for x in quality], np.int32)
    return (engine or default_engine()).group_files_pdq(hashes, similarity, coefficients, has_features, q)
