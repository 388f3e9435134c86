"""Grouping half of /root/reference/src/scanner.rs (lines 1588-1832) on top of the C ABI.

    PDQ_MIN_QUALITY, is_low_pdq_quality            scanner.rs:1588-1594
    group_with_pdqhash / group_files_generic       scanner.rs:1640-1832 up to the union-find
    group_max_dist                                 scanner.rs:2214-2241 (the per-group max_dist of process_raw_groups)
    load_image_fast ("jpg" | "jpeg" arm)           scanner.rs:461-508
File-name logic after the union-find (merge_groups_by_stem, the sorting inside process_raw_groups) stays with the caller.
"""
import numpy as np

from . import _lib
from .engine import default_engine

PDQ_MIN_QUALITY = 50


def load_image_fast(path, data, engine=None, flavour=_lib.RPH_JPEG_ZUNE):
    """The "jpg" | "jpeg" arm of load_image_fast (scanner.rs:461-508): decoded on the device, returned as (h, w) uint8 [Luma8] or
    (h, w, 3) [Rgb8] -- the DynamicImage the reference builds from zune-jpeg's buffer.  A stream the device path does not take
    (CMYK, arithmetic coding, corrupt) raises RphError: the caller goes on to its next decoder, as the reference goes from tier 1 to
    tier 2 (scanner.rs:510-551).  Every other extension is the host's business (ValueError)."""
    import os

    ext = os.path.splitext(str(path))[1].lstrip(".").lower()
    if ext not in ("jpg", "jpeg"):
        raise ValueError(f"load_image_fast: '{ext}' files are decoded by the host's decoders, not by this library")
    return (engine or default_engine()).jpeg_decode(data, flavour)


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


def group_max_dist(groups, hashes, pivots, coefficients=None, has_features=None, engine=None):
    """Per-group `max_dist` of process_raw_groups (scanner.rs:2214-2241).

    `groups`: lists of file indices (in the caller's display order: the reference sorts by file name first, which stays with
    the caller); `pivots[g]`: index of the group's pivot file, i.e. the first file of the sorted group that has features, else
    the first that has a hash (the caller's find_map), or None.  With features the distance of a member is the minimum over
    the pivot's 8 dihedral hashes (one batched rph_pdq_hashes_from_coeffs for all pivots), otherwise the plain distance to the
    pivot's hash; the group's value is the maximum over its members that have a hash (here: all listed members)."""
    eng = engine or default_engine()
    hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
    bits = np.unpackbits(hashes, axis=1)
    out = [0] * len(groups)
    with_feats = [g for g, p in enumerate(pivots)
                  if p is not None and coefficients is not None and (has_features is None or has_features[p])]
    variants = {}
    if with_feats:
        c = np.ascontiguousarray(np.asarray(coefficients, np.float32).reshape(-1, 256)[[pivots[g] for g in with_feats]])
        _, dih = eng.pdq_hashes_from_coeffs(c, want_hash=False, want_dihedral=True)
        variants = {g: np.unpackbits(dih[k], axis=1) for k, g in enumerate(with_feats)}
    for g, members in enumerate(groups):
        p = pivots[g]
        if p is None or not len(members):
            continue
        mb = bits[np.asarray(members, np.int64)]
        if g in variants:
            d = (mb[:, None, :] != variants[g][None, :, :]).sum(axis=2).min(axis=1)
        else:
            d = (mb != bits[p][None, :]).sum(axis=1)
        out[g] = int(d.max())
    return out
