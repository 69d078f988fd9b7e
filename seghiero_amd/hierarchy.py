"""Hierarchy-map helpers: YAML range lists -> index tensors (host side, integer, bit-exact).

Mirrors reference ``train.py:52-99`` (``build_fine_to_coarse_map``, ``build_hiera_index``,
``build_fine_to_super_map``): same names, argument meaning and results.  Entries are ``[lbl]`` or an inclusive
``[start, end]`` range of FINE ids; entry ``i`` maps them to level id ``i``.

Documented deviation: the reference starts from ``torch.empty`` and leaves ids that no entry
covers as uninitialised memory; here that raises ``ValueError`` instead of training on garbage.
"""
import torch


def _entries(cfg):
    out = []
    for sub in cfg:
        if len(sub) == 1:
            out.append((int(sub[0]), int(sub[0])))
        else:
            out.append((int(sub[0]), int(sub[1])))
    return out


def _level_map(cfg, n_fine, name):
    table = [-1] * n_fine
    for idx, (lo, hi) in enumerate(_entries(cfg)):
        for f in range(lo, hi + 1):
            table[f] = idx          # IndexError for ids >= n_fine, as in the reference
    holes = [f for f, v in enumerate(table) if v < 0]
    if holes:
        raise ValueError(f"{name}: fine ids {holes} are not covered by any entry")
    return torch.tensor(table, dtype=torch.long)


def build_fine_to_coarse_map(coarse_to_fine_cfg: list, n_fine: int) -> torch.Tensor:
    return _level_map(coarse_to_fine_cfg, n_fine, "coarse_to_fine_map")


def build_hiera_index(coarse_to_fine_cfg: list) -> list:
    return [[lo, hi + 1] for lo, hi in _entries(coarse_to_fine_cfg)]


def build_fine_to_super_map(super_to_coarse_cfg: list, n_fine: int) -> torch.Tensor:
    return _level_map(super_to_coarse_cfg, n_fine, "super_coarse_to_coarse_map")
