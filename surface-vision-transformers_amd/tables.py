"""Icosahedral patch-index tables: ico-6 vertex ids of every triangular patch.

data/ico6_sub_ico_{1,2}.npy are the reference's utils/triangle_indices_ico_6_sub_ico_{1,2}.csv packed
patch-major as uint16 (P, V) by data/make_tables.py; data/ico6_sub_ico_3_synth.npy is a documented
SYNTHETIC 1280 x 45 table (the reference ships none for 1280 patches)."""
import os

import numpy as np
import torch

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
ICO6_VERTICES = 40962
_FILES = {(80, 561): "ico6_sub_ico_1.npy", (320, 153): "ico6_sub_ico_2.npy", (1280, 45): "ico6_sub_ico_3_synth.npy"}
_SYNTHETIC = {(1280, 45)}
_cache = {}


def load_table(num_patches, num_vertices, allow_synthetic=False):
    """(P, V) uint16 numpy array, patch-major.  The (1280, 45) table is SYNTHETIC (45-vertex slices of the sub_ico_2
    patches, not the geometric sub_ico_3 grid the reference would use): fine for throughput runs on synthetic
    surfaces, wrong for real 1280-patch data -- loading it warns unless allow_synthetic=True; real data needs
    SiT.set_patch_table() with the user's own table."""
    key = (int(num_patches), int(num_vertices))
    if key in _SYNTHETIC and not allow_synthetic:
        import warnings
        warnings.warn(f"sitk.tables: the built-in {key[0]} x {key[1]} patch table is SYNTHETIC (the reference ships no 1280-patch "
                      f"table); gathers from real surfaces need your own table (SiT.set_patch_table). Pass "
                      f"allow_synthetic=True / set model.allow_synthetic_table = True to silence this.", stacklevel=2)
    if key not in _FILES:
        raise KeyError(f"no ico-6 patch table for num_patches={key[0]}, num_vertices={key[1]}; "
                       f"available: {sorted(_FILES)} (pass your own (P, V) table instead)")
    if key not in _cache:
        t = np.load(os.path.join(_DATA, _FILES[key]), allow_pickle=False)
        assert t.shape == key and t.dtype == np.uint16
        _cache[key] = t
    return _cache[key]


def table_tensor(table, device):
    """Device tensor holding the uint16 ids (stored with dtype int16: same bits, universally supported)."""
    t = np.ascontiguousarray(np.asarray(table))
    if t.min() < 0 or t.max() >= 65536:
        raise ValueError("vertex ids must fit in uint16")
    return torch.from_numpy(t.astype(np.uint16).view(np.int16)).to(device)
