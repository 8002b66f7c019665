"""Host-side mirror of the keyframe descriptor path over the C ABI (first half):
  describeBRIEF  <- FeatureExtractor::describeBRIEF  src/feature_extractor.cpp:224-285
  matchToMap     <- Mapper::matchToMap               src/mapper.cpp:576-774 (flat inputs: MatchInput)
Plumbing only (ctypes + numpy); the box sums, projections and Hamming distances run in csrc/match.hip."""
import ctypes as C

import numpy as np

from .frontend import _check

f32p, f64p, i32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


class MatchInputC(C.Structure):
    _fields_ = [("Twc", C.c_double * 7), ("K", C.c_double * 4), ("img_w", C.c_int32), ("img_h", C.c_int32), ("cell", C.c_int32),
                ("nb3dkps", C.c_int32), ("n_kp", C.c_int32), ("kp_px", f32p), ("kp_desc_ptr", i32p), ("kp_descs", u8p),
                ("kp_kf_ptr", i32p), ("kp_kfids", i32p), ("kp_kf_px", f32p), ("grid_ptr", i32p), ("grid_kp", i32p),
                ("n_cand", C.c_int32), ("cand_wpt", f64p), ("cand_desc_ptr", i32p), ("cand_descs", u8p), ("cand_kf_ptr", i32p),
                ("cand_kfids", i32p), ("n_kf", C.c_int32), ("kf_Twc", f64p), ("cam", C.c_void_p)]


def _csr(lists, dtype, width=1):
    ptr = np.zeros(len(lists) + 1, np.int32)
    ptr[1:] = np.cumsum([len(x) for x in lists])
    flat = np.concatenate([np.asarray(x, dtype).reshape(-1, width) for x in lists]) if len(lists) and ptr[-1] else np.zeros((0, width), dtype)
    return ptr, np.ascontiguousarray(flat.reshape(-1, width) if width > 1 else flat.ravel(), dtype)


class MatchInput:
    """owns the flat arrays of one Mapper::matchToMap call.
    kps: list of dict(px (2,), descs (d,32) u8, kfids [ascending], kf_px (len(kfids),2)); cands: list of dict(wpt (3,), descs,
    kfids); kf_Twc: (n_kf, 7) indexed by kfid; the grid (Frame::vgridkps_) is rebuilt from kp order = insertion order."""

    def __init__(self, Twc, K, img_w, img_h, cell, nb3dkps, kps, cands, kf_Twc):
        c = np.ascontiguousarray
        self.kp_px = c([k["px"] for k in kps], np.float32).reshape(-1, 2)
        self.kp_desc_ptr, self.kp_descs = _csr([k["descs"] for k in kps], np.uint8, 32)
        self.kp_kf_ptr, self.kp_kfids = _csr([k["kfids"] for k in kps], np.int32)
        _, self.kp_kf_px = _csr([k["kf_px"] for k in kps], np.float32, 2)
        nbw, nbh = int(np.ceil(np.float32(img_w) / np.float32(cell))), int(np.ceil(np.float32(img_h) / np.float32(cell)))
        cells = [[] for _ in range(nbw * nbh)]
        for i, k in enumerate(kps):
            r, cc = int(np.floor(np.float32(k["px"][1]) / np.float32(cell))), int(np.floor(np.float32(k["px"][0]) / np.float32(cell)))
            cells[r * nbw + cc].append(i)
        self.grid_ptr, self.grid_kp = _csr(cells, np.int32)
        self.cand_wpt = c([q["wpt"] for q in cands], np.float64).reshape(-1, 3)
        self.cand_desc_ptr, self.cand_descs = _csr([q["descs"] for q in cands], np.uint8, 32)
        self.cand_kf_ptr, self.cand_kfids = _csr([q["kfids"] for q in cands], np.int32)
        self.kf_Twc = c(kf_Twc, np.float64).reshape(-1, 7)
        m = MatchInputC()
        m.Twc[:] = np.asarray(Twc, np.float64).tolist()
        m.K[:] = np.asarray(K, np.float64).tolist()
        m.img_w, m.img_h, m.cell, m.nb3dkps = int(img_w), int(img_h), int(cell), int(nb3dkps)
        m.n_kp, m.n_cand, m.n_kf = len(kps), len(cands), len(self.kf_Twc)
        p = lambda a, t: a.ctypes.data_as(t)
        m.kp_px, m.kp_desc_ptr, m.kp_descs = p(self.kp_px, f32p), p(self.kp_desc_ptr, i32p), p(self.kp_descs, u8p)
        m.kp_kf_ptr, m.kp_kfids, m.kp_kf_px = p(self.kp_kf_ptr, i32p), p(self.kp_kfids, i32p), p(self.kp_kf_px, f32p)
        m.grid_ptr, m.grid_kp = p(self.grid_ptr, i32p), p(self.grid_kp, i32p)
        m.cand_wpt, m.cand_desc_ptr, m.cand_descs = p(self.cand_wpt, f64p), p(self.cand_desc_ptr, i32p), p(self.cand_descs, u8p)
        m.cand_kf_ptr, m.cand_kfids, m.kf_Twc = p(self.cand_kf_ptr, i32p), p(self.cand_kfids, i32p), p(self.kf_Twc, f64p)
        self.c = m


def describeBRIEF(ctx, pyr, pts, pattern, b=0):
    """returns (desc (n,32) u8, valid (n,) bool)"""
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    pat = np.ascontiguousarray(pattern, np.int8).reshape(256, 4)
    n = len(pts)
    desc, valid = np.zeros((n, 32), np.uint8), np.zeros(n, np.uint8)
    _check(ctx.h, ctx.lib.ov2_describe_brief(ctx.h, pyr.h, int(b), n, pts.ctypes.data, pat.ctypes.data, desc.ctypes.data, valid.ctypes.data))
    return desc, valid.astype(bool)


def matchToMap(ctx, inp, fmaxprojerr=2.0, fdistratio=0.2):
    """returns (match_cand (n_kp,) int32: candidate index or -1, match_dist (n_kp,) f32)"""
    n = inp.c.n_kp
    mc, md = np.full(max(n, 1), -1, np.int32), np.zeros(max(n, 1), np.float32)
    _check(ctx.h, ctx.lib.ov2_match_to_map(ctx.h, C.addressof(inp.c), float(fmaxprojerr), float(fdistratio), mc.ctypes.data, md.ctypes.data))
    return mc[:n], md[:n]


def random_brief_pattern(seed=0):
    """a 256 x 4 test table with the statistics of BRIEF's (isotropic Gaussian offsets, clipped to the patch): the real
    table of opencv_contrib is not in the reference tree"""
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.normal(0, 48 / 5.0, (256, 4))), -24, 24).astype(np.int8)
