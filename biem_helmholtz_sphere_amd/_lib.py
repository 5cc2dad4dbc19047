"""ctypes binding of libbiem_mi355.so (the C ABI declared in include/biem_mi355.h).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbiem_mi355.so")

BIEM_OK = 0
TREE_IDS = {"a": 0, "ba": 1, "bba": 2, "caa": 3}
FILL_REFERENCE, FILL_EQUILIBRATED, FILL_SYMMETRIC = 0, 1, 2
USCAT_FAR_FIELD, USCAT_PER_BALL, USCAT_KIND_INNER, USCAT_POINTS_BATCHED = 1, 2, 4, 8

_vp, _i, _ll, _sz, _dp, _ip = C.c_void_p, C.c_int, C.c_longlong, C.c_size_t, C.c_void_p, C.c_void_p

# name -> (restype, argtypes): must list every symbol of include/biem_mi355.h (tests/test_host_logic.py::test_library_exports_every_declared_symbol checks it)
SIGNATURES = {
    "biem_version": (_i, []),
    "biem_build_id": (C.c_char_p, []),
    "biem_last_error": (C.c_char_p, []),
    "biem_device_count": (_i, [C.POINTER(_i)]),
    "biem_plan_create_host": (_i, [_i, _i, C.POINTER(_vp)]),
    "biem_plan_upload": (_i, [_vp]),
    "biem_plan_create": (_i, [_i, _i, C.POINTER(_vp)]),
    "biem_plan_destroy": (_i, [_vp]),
    "biem_plan_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_ll)]),
    "biem_plan_labels": (_i, [_vp, _vp, _vp]),
    "biem_plan_symmetric_order": (_i, [_vp, _vp, _vp]),
    "biem_plan_fill_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "biem_plan_quadrature": (_i, [_vp, _vp, _vp]),
    "biem_plan_projection": (_i, [_vp, _vp]),
    "biem_plan_terms": (_i, [_vp, _vp, _vp, _vp]),
    "biem_radial": (_i, [_i, _i, _i, _dp, _dp, _vp]),
    "biem_radial_complex": (_i, [_i, _i, _i, _dp, _dp, _vp]),
    "biem_harmonics": (_i, [_vp, _i, _dp, _dp, _vp]),
    "biem_ball_tables": (_i, [_vp, _i, _i, _dp, _dp, _dp, _i, _dp, _dp, _i, _dp, _vp]),
    "biem_rhs_project": (_i, [_vp, _i, _i, _i, _dp, _dp, _ll, _ll, _ll, _vp]),
    "biem_fill_workspace_bytes": (_sz, [_vp, _i, _i]),
    "biem_fill": (_i, [_vp, _i, _i, _dp, _dp, _i, _dp, _i, _dp, _ll, _ll, _i, _vp, _sz, _vp]),
    "biem_lu_npad": (_i, [_i]),
    "biem_lu_workspace_bytes": (_sz, [_i, _i, _i]),
    "biem_lu_factor_solve": (_i, [_i, _i, _i, _dp, _ll, _ll, _ip, _ip, _vp, _sz, _vp]),
    "biem_lu_factor": (_i, [_i, _i, _dp, _ll, _ll, _ip, _ip, _vp, _sz, _vp]),
    "biem_ldlt_factor": (_i, [_i, _i, _dp, _ll, _ll, _ip, _ip, _vp, _sz, _vp]),
    "biem_lu_solve": (_i, [_i, _i, _i, _dp, _ll, _ll, _ip, _dp, _ll, _ll, _vp]),
    "biem_density": (_i, [_vp, _i, _i, _i, _dp, _ll, _ll, _ll, _dp, _dp, _vp]),
    "biem_uscat_workspace_bytes": (_sz, [_vp, _i, _i]),
    "biem_uscat": (_i, [_vp, _i, _i, _i, _dp, _dp, _dp, _dp, _i, _dp, _dp, _i, _dp, _vp, _sz, _vp]),
    "biem_solve_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "biem_solve": (_i, [_vp, _i, _i, _i, _dp, _dp, _dp, _dp, _i, _dp, _dp, _i, _dp, _dp, _ip, _i, _vp, _sz, _vp]),
    "biem_solve_ldlt": (_i, [_vp, _i, _i, _i, _dp, _dp, _dp, _dp, _i, _dp, _dp, _i, _dp, _dp, _ip, _i, _vp, _sz, _vp]),
    "biem_ldlt_factor_solve": (_i, [_i, _i, _i, _dp, _ll, _ll, _ip, _ip, _vp, _sz, _vp]),
    "biem_sym_factor_solve": (_i, [_i, _i, _i, _dp, _ll, _ll, _ip, _vp, _sz, _vp]),
    "biem_profile_begin": (_i, []),
    "biem_profile_end": (_i, [_vp, _vp, _vp]),
    "biem_bench_mfma_f64": (_i, [_i, C.POINTER(C.c_double), _vp]),
    "biem_bench_mfma_f64_ex": (_i, [_i, _i, C.POINTER(C.c_double), _vp]),
}

_lock = threading.Lock()
_lib = None


class BiemLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library; raises loudly if it is missing and cannot be built (no fallback path exists).

    The existence / staleness test and a rebuild happen under a file lock (one process per GPU: several ranks get here at
    once), the build writes a temporary file and renames it, so no rank ever maps a half-written library.  A library whose
    stored source hash differs from the checkout's (`_build.is_stale`) is rebuilt when hipcc is present; without hipcc a
    missing library is an error and a stale one a warning.
    """
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        import fcntl

        from . import _build

        import contextlib
        import warnings

        # A read-only install (or a package shipped without csrc/) must still load a library that exists: the lock file and
        # the staleness test are best effort there.  A different BIEM_HIPCC_FLAGS in the environment is part of the hash; it is
        # an experiment switch of the build script, never set by the product.
        try:
            lk = open(LIB_PATH + ".lock", "a")
        except OSError:
            lk = None
        with (lk if lk is not None else contextlib.nullcontext()):
            if lk is not None:
                fcntl.flock(lk, fcntl.LOCK_EX)
            missing = not os.path.exists(LIB_PATH)
            try:
                stale = missing or _build.is_stale()
            except OSError as e:              # no sources to compare with: take the library as it is
                if missing:
                    raise BiemLibraryError(f"{LIB_PATH} is missing and the sources to build it are not readable ({e}). "
                                           "This package has no CPU fallback.") from e
                stale = False
            if stale:
                try:
                    _build.build(force=True)       # not a fallback: the same HIP library, compiled now
                except Exception as e:  # noqa: BLE001
                    if missing:
                        raise BiemLibraryError(
                            f"{LIB_PATH} is missing and could not be built ({e}). Build it with "
                            "`python -m biem_helmholtz_sphere_amd._build` (hipcc --offload-arch=gfx950). "
                            "This package has no CPU fallback."
                        ) from e
                    warnings.warn(f"{LIB_PATH} was built from other sources than this checkout's csrc/ (hash "
                                  f"{_build.built_hash()} vs {_build.source_hash()}) and could not be rebuilt: {e}", RuntimeWarning)
            lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc: int, what: str = "") -> None:
    if rc != BIEM_OK:
        msg = load().biem_last_error()
        raise BiemLibraryError(f"{what or 'libbiem_mi355'} failed (status {rc}): {msg.decode() if msg else ''}")
