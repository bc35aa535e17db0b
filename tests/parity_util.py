"""Whole-frame comparison of a product frame with the checker's (oracle) frame, to north_star's bars: primary hit ids
identical except t-ties within 1e-6 (fraction <= 1e-4, reported), shadow bits identical on the agreeing pixels, radiance
within 1e-3 relative.  Used by tests/test_gpu_parity.py and by bench.py's cpu_baseline leg (outside its timed region)."""
import numpy as np


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def shadow_bits(rgb, ids, ref_rgb_lit_floor=None):
    """a pixel of a one-light scene is in shadow iff its colour has no diffuse / specular part, i.e. it equals the
    smallest colour any pixel of the same object takes.  Returns a bool array (True = in shadow), False for misses."""
    out = np.zeros(len(ids), dtype=bool)
    for i in np.unique(ids[ids >= 0]) if len(np.unique(ids)) < 64 else []:
        m = ids == i
        floor = rgb[m].min(axis=0)
        out[m] = (rgb[m] <= floor + 1e-9).all(axis=1)
    return out


def frame_parity(rgb, ids, ref_rgb, ref_ids, t_of_mismatch=None, two_class_ids=None):
    """rgb (n,3), ids (n,) of the product; ref_* of the oracle on the same pixels.
    t_of_mismatch(idx) -> (t_product, t_oracle) for the mismatching pixels (optional: gives max_dt / the tie check).
    two_class_ids: optional function mapping prim ids to a few classes (e.g. plane / mesh) for the shadow-bit census."""
    n = len(ids)
    mism = np.nonzero(ids != ref_ids)[0]
    same = ids == ref_ids
    res = {"rays": int(n), "id_mismatches": int(len(mism)), "tie_fraction": float(len(mism)) / max(1, n), "max_dt": None}
    if len(mism) and t_of_mismatch is not None:
        tp, to = t_of_mismatch(mism)
        res["max_dt"] = float(np.abs(np.asarray(tp) - np.asarray(to)).max())
    err = rel_err(rgb[same], ref_rgb[same])
    res["max_rel_colour_err"] = float(err.max()) if err.size else 0.0
    # shadow bits: with one light a lit pixel is strictly brighter than the ambient-only colour of its object
    cls = two_class_ids(ids) if two_class_ids is not None else np.where(ids >= 0, 0, -1)
    sb, sb_ref = shadow_bits(rgb, cls), shadow_bits(ref_rgb, two_class_ids(ref_ids) if two_class_ids is not None else np.where(ref_ids >= 0, 0, -1))
    hit = same & (ids >= 0)
    res["shadow_rays"] = int(hit.sum())
    res["shadow_bit_mismatches"] = int((sb[hit] != sb_ref[hit]).sum())
    res["pixels_in_shadow"] = int(sb_ref[hit].sum())
    return res
