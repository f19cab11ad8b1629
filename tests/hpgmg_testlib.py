"""Shared helpers for the test-suite.

`Backend` wraps one build of the host layer behind the same ctypes API:
    Backend.hip()     -> hpgmg_amd/libhpgmg_fv.so  (HIP operator plugin; needs a GPU to compute)
    Backend.oracle()  -> oracle/liboracle_fv.so    (CPU restatement; the CHECKER, never the product)
so a parity test is literally "run the same calls on both and compare bytes".
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hpgmg_amd as H  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
REFERENCE_SRC = "/root/reference/finite-volume/source"


def have_reference():
    return os.path.isdir(REFERENCE_SRC)


def build_oracle():
    """(Re)build oracle/liboracle_fv.so with gcc; cheap when up to date."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
    return os.path.join(ROOT, "oracle", "liboracle_fv.so")


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)


class Level:
    """A level_type* plus numpy views of its geometry."""

    def __init__(self, backend, ptr, owned=False):
        self.b, self.ptr, self.owned = backend, ptr, owned
        info = (ctypes.c_int * H.INFO_COUNT)()
        backend.lib.hpgmg_level_info(ptr, info)
        self.info = list(info)
        self.dim, self.box_dim, self.ghosts = info[H.INFO_DIM], info[H.INFO_BOX_DIM], info[H.INFO_GHOSTS]
        self.jStride, self.kStride, self.volume = info[H.INFO_JSTRIDE], info[H.INFO_KSTRIDE], info[H.INFO_VOLUME]
        self.num_boxes, self.num_vectors = info[H.INFO_NUM_MY_BOXES], info[H.INFO_NUM_VECTORS]

    @property
    def h(self):
        return self.b.lib.hpgmg_level_h(self.ptr)

    @property
    def eigenvalue(self):
        return self.b.lib.hpgmg_level_eigenvalue(self.ptr)

    def box_low(self, box):
        out = (ctypes.c_int * 3)()
        self.b.lib.hpgmg_level_box_low(self.ptr, box, out)
        return tuple(out)

    def padded_shape(self):
        w = self.box_dim + 2 * self.ghosts
        return (w, w, self.jStride)

    def read(self, box, vid):
        """Whole padded box of vector `vid` as a (k, j, i) array (includes ghosts and row padding)."""
        buf = np.empty(self.volume, dtype=np.float64)
        self.b.lib.hpgmg_level_read_vector(self.ptr, box, vid, buf.ctypes.data)
        w = self.box_dim + 2 * self.ghosts
        return buf[: w * self.kStride].reshape(w, self.kStride)[:, : w * self.jStride].reshape(w, w, self.jStride).copy()

    def read_raw(self, box, vid):
        buf = np.empty(self.volume, dtype=np.float64)
        self.b.lib.hpgmg_level_read_vector(self.ptr, box, vid, buf.ctypes.data)
        return buf

    def write_raw(self, box, vid, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float64)
        assert flat.size == self.volume
        self.b.lib.hpgmg_level_write_vector(self.ptr, box, vid, flat.ctypes.data)

    def read_all(self, vid):
        return np.stack([self.read_raw(b, vid) for b in range(self.num_boxes)])

    def write_all(self, vid, arr):
        for b in range(self.num_boxes):
            self.write_raw(b, vid, arr[b])

    def interior(self, vid):
        """Global (k, j, i) array of the interior cells of this rank's boxes (single-rank levels only)."""
        n, d, g = self.dim, self.box_dim, self.ghosts
        out = np.full((n, n, n), np.nan)
        for b in range(self.num_boxes):
            li, lj, lk = self.box_low(b)
            a = self.read(b, vid)
            out[lk:lk + d, lj:lj + d, li:li + d] = a[g:g + d, g:g + d, g:g + d]
        return out

    def destroy(self):
        if self.owned and self.ptr:
            self.b.lib.hpgmg_level_destroy(self.ptr)
            self.ptr = None


class Backend:
    def __init__(self, lib, name):
        self.lib, self.name = lib, name
        lib.hpgmg_set_verbose(0)

    @staticmethod
    def hip():
        return Backend(H.load_driver(), "hip")

    @staticmethod
    def oracle():
        return Backend(H.bind_driver_library(build_oracle()), "oracle")

    def configure(self, op=H.OP_7PT, smoother=H.SMOOTH_CHEBY, helmholtz=0, variable_coeff=1):
        cfg = H.Config(op, smoother, helmholtz, variable_coeff)
        rc = self.lib.hpgmg_configure(ctypes.byref(cfg))
        assert rc == 0, "configuration rejected"
        return cfg

    def vectors_reserved(self):
        return self.lib.hpgmg_vectors_reserved()

    def level(self, boxes_in_i, box_dim, ghosts=None, num_vectors=None, bc=H.BC_DIRICHLET, rank=0, ranks=1, h=None):
        ghosts = self.lib.stencil_get_radius() if ghosts is None else ghosts
        num_vectors = self.vectors_reserved() if num_vectors is None else num_vectors
        h = 1.0 / (boxes_in_i * box_dim) if h is None else h
        ptr = self.lib.hpgmg_level_create(boxes_in_i, box_dim, ghosts, num_vectors, bc, rank, ranks, h)
        return Level(self, ptr, owned=True)

    def solver(self, boxes_in_i, box_dim, bc=H.BC_DIRICHLET, rank=0, ranks=1):
        return Solver(self, self.lib.hpgmg_solver_create_explicit(boxes_in_i, box_dim, bc, rank, ranks))

    def solver_cli(self, log2_box_dim, boxes_per_rank, rank=0, ranks=1, bc=H.BC_DIRICHLET):
        ptr = self.lib.hpgmg_solver_create(log2_box_dim, boxes_per_rank, bc, rank, ranks)
        assert ptr, "no acceptable problem size"
        return Solver(self, ptr)


class Solver:
    def __init__(self, backend, ptr):
        self.b, self.ptr = backend, ptr

    def num_levels(self):
        return self.b.lib.hpgmg_solver_num_levels(self.ptr)

    def level(self, l):
        return Level(self.b, self.b.lib.hpgmg_solver_level(self.ptr, l))

    def fmg(self, l=0):
        return self.b.lib.hpgmg_solver_fmg(self.ptr, l)

    def restrict_rhs(self, l):
        self.b.lib.hpgmg_solver_restrict_rhs(self.ptr, l)

    def three_sizes(self):
        """F-cycle residual norms at h, 2h, 4h exactly as the benchmark computes them (hpgmg-fv.c:320-329)."""
        out = []
        for l in range(3):
            self.restrict_rhs(l)
            out.append(self.fmg(l))
        return out

    def richardson(self):
        out = (ctypes.c_double * 2)()
        self.b.lib.hpgmg_solver_richardson(self.ptr, out)
        return out[0], out[1]

    def bench(self, l, warmup, solves):
        return self.b.lib.hpgmg_solver_bench(self.ptr, l, warmup, solves)

    def destroy(self):
        if self.ptr:
            self.b.lib.hpgmg_solver_destroy(self.ptr)
            self.ptr = None


def seeded_field(level, seed, scale=1.0):
    """Deterministic pseudo-random padded boxes (ghosts included), one array per box."""
    rng = np.random.default_rng(seed)
    return scale * (rng.random((level.num_boxes, level.volume)) * 2.0 - 1.0)


def split_variant(name):
    """'7pt-cheby-periodic' -> (VARIANTS key, boundary condition)"""
    if name.endswith("-periodic"):
        return name[:-len("-periodic")], H.BC_PERIODIC
    return name, H.BC_DIRICHLET


VARIANTS = {
    "7pt-cheby": dict(op=H.OP_7PT, smoother=H.SMOOTH_CHEBY, helmholtz=0, variable_coeff=1),
    "7pt-gsrb": dict(op=H.OP_7PT, smoother=H.SMOOTH_GSRB, helmholtz=0, variable_coeff=1),
    "7pt-cheby-helm": dict(op=H.OP_7PT, smoother=H.SMOOTH_CHEBY, helmholtz=1, variable_coeff=1),
    "7ptcc-cheby": dict(op=H.OP_7PT, smoother=H.SMOOTH_CHEBY, helmholtz=0, variable_coeff=0),
    "7pt-jacobi": dict(op=H.OP_7PT, smoother=H.SMOOTH_JACOBI, helmholtz=0, variable_coeff=1),
    "27pt-cheby": dict(op=H.OP_27PT, smoother=H.SMOOTH_CHEBY, helmholtz=0, variable_coeff=0),
    "27pt-gsrb": dict(op=H.OP_27PT, smoother=H.SMOOTH_GSRB, helmholtz=0, variable_coeff=0),
    "fv4-gsrb": dict(op=H.OP_FV4, smoother=H.SMOOTH_GSRB, helmholtz=0, variable_coeff=1),
    "fv4-gsrb-helm": dict(op=H.OP_FV4, smoother=H.SMOOTH_GSRB, helmholtz=1, variable_coeff=1),
    "fv4-cheby": dict(op=H.OP_FV4, smoother=H.SMOOTH_CHEBY, helmholtz=0, variable_coeff=1),
    "fv2-cheby": dict(op=H.OP_FV2, smoother=H.SMOOTH_CHEBY, helmholtz=0, variable_coeff=1),
}
