"""Hierarchy-map helpers (oracle).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates reference ``train.py:52-99``:

* ``build_fine_to_coarse_map``  (``train.py:52-66``): each YAML entry is ``[lbl]`` or an
  *inclusive* ``[start, end]`` range of fine ids; entry ``i`` maps those ids to coarse ``i``.
* ``build_hiera_index``         (``train.py:69-83``): the same entries as half-open
  ``[start, end+1]`` pairs.
* ``build_fine_to_super_map``   (``train.py:86-99``): identical arithmetic applied to
  ``super_coarse_to_coarse_map`` -- i.e. the entries are interpreted as FINE id ranges.

Deviation (documented): the reference starts from ``torch.empty`` so ids
not covered by any entry are uninitialised garbage; here an uncovered id raises ``ValueError``.
"""
import torch


def _ranges(cfg):
    for idx, sub in enumerate(cfg):
        if len(sub) == 1:
            yield idx, int(sub[0]), int(sub[0])
        else:
            yield idx, int(sub[0]), int(sub[1])


def _fine_to_level(cfg, n_fine, what):
    out = torch.full((n_fine,), -1, dtype=torch.long)
    for idx, lo, hi in _ranges(cfg):
        out[lo:hi + 1] = idx
    if bool((out < 0).any()):
        missing = torch.nonzero(out < 0).flatten().tolist()
        raise ValueError(f"{what}: fine ids {missing} are not covered by any entry")
    return out


def build_fine_to_coarse_map(coarse_to_fine_cfg, n_fine):
    return _fine_to_level(coarse_to_fine_cfg, n_fine, "coarse_to_fine_map")


def build_hiera_index(coarse_to_fine_cfg):
    return [[lo, hi + 1] for _, lo, hi in _ranges(coarse_to_fine_cfg)]


def build_fine_to_super_map(super_to_coarse_cfg, n_fine):
    return _fine_to_level(super_to_coarse_cfg, n_fine, "super_coarse_to_coarse_map")
