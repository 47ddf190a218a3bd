"""Build / bind / time the C++ CPU restatement (oracle/scn_cpu_baseline.cpp).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.build() and bench.py's cpu_baseline leg -- never by the product (sparse_rcnn_amd).

Two builds of the same source:
  portable  -march=x86-64-v3 (AVX2 + FMA), made by __graft_entry__.build(); runs on any box this image runs on; used by
            the CPU tests that check the restatement against the Python oracle.
  native    -march=native, made on the box that TIMES it (bench.py), so the timed baseline uses what the host CPU has
            (AVX-512 where present); falls back to the portable build if no compiler is there.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import platform
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "scn_cpu_baseline.cpp")
OUT = os.path.join(HERE, "build")


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def _cpu_flags_digest():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return hashlib.sha1(line.encode()).hexdigest()[:10]
    except OSError:
        pass
    return "noflags"


def build(native=False, force=False):
    """-> path of the shared library (compiles when missing or older than the source)."""
    os.makedirs(OUT, exist_ok=True)
    tag = f"native_{_cpu_flags_digest()}" if native else "portable"
    lib = os.path.join(OUT, f"libscn_cpu_baseline_{tag}.so")
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= os.path.getmtime(SRC):
        return lib
    march = "-march=native" if native else "-march=x86-64-v3"
    cmd = ["g++", "-std=c++17", "-O3", march, "-fopenmp", "-fPIC", "-shared", "-o", lib, SRC]
    subprocess.run(cmd, check=True)
    return lib


_libs = {}


def load(native=False):
    key = bool(native)
    if key not in _libs:
        try:
            path = build(native=native)
        except (OSError, subprocess.CalledProcessError):
            if not native:
                raise
            path = build(native=False)
        lib = C.CDLL(path)
        lib.scn_cpu_unet_step.restype = C.c_int
        lib.scn_cpu_unet_step.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64), C.c_int]
        lib.scn_cpu_max_threads.restype = C.c_int
        lib.scn_cpu_isa.restype = C.c_char_p
        _libs[key] = lib
    return _libs[key]


def flat_params(params: dict, cin, channels):
    """oracle.scn_oracle parameter dict -> one fp32 vector in unet_param_shapes order (the C entry point's layout)."""
    from . import scn_oracle as O
    parts = [np.ascontiguousarray(params[name].detach().cpu().numpy(), dtype=np.float32).reshape(-1)
             for name, _ in O.unet_param_shapes(cin, list(channels))]
    return np.concatenate(parts)


def unflatten(vec, cin, channels):
    from . import scn_oracle as O
    out, off = {}, 0
    for name, shape in O.unet_param_shapes(cin, list(channels)):
        n = int(np.prod(shape))
        out[name] = vec[off:off + n].reshape(shape)
        off += n
    return out


def unet_step(coords, feats, channels, params_flat, dY=None, threads=0, native=False, want=("out", "grads", "dfeats")):
    """One full step (rulebooks + fwd + bwd).  -> dict(out [N0,C0], grads flat, dfeats [Npts,cin], n_active, n_rules0)."""
    lib = load(native)
    coords = np.ascontiguousarray(coords, dtype=np.int64)
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    n_pts, cin = feats.shape
    ch = np.ascontiguousarray(channels, dtype=np.int32)
    pf = np.ascontiguousarray(params_flat, dtype=np.float32)
    out = np.empty((n_pts, int(ch[0])), np.float32) if "out" in want else None
    grads = np.empty_like(pf) if "grads" in want else None
    dfeats = np.empty_like(feats) if "dfeats" in want else None
    dy = None if dY is None else np.ascontiguousarray(dY, dtype=np.float32)
    n_active, n_rules = C.c_int64(0), C.c_int64(0)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = lib.scn_cpu_unet_step(p(coords), n_pts, p(feats), cin, p(ch), len(ch), p(pf), p(dy), p(out), p(grads), p(dfeats),
                               C.byref(n_active), C.byref(n_rules), int(threads))
    if rc:
        raise RuntimeError(f"scn_cpu_unet_step failed with code {rc}")
    n0 = int(n_active.value)
    return dict(out=None if out is None else out[:n0], grads=grads, dfeats=dfeats, n_active=n0, n_rules0=int(n_rules.value))


def timed_baseline(coords, feats, channels, threads, budget_s=12.0):
    """bench.py's `cpu_baseline` object: the restatement timed single-threaded and on `threads` host threads on whole
    steps of the same scene; `value` is the multi-thread throughput (the stronger baseline)."""
    from . import scn_oracle as O
    cin = feats.shape[1]
    pf = flat_params(O.init_unet_params(cin, list(channels), seed=0), cin, channels)
    lib = load(native=True)

    def run(th, reps):
        t0 = time.perf_counter()
        n = 0
        for _ in range(reps):
            n = unet_step(coords, feats, channels, pf, threads=th, native=True, want=())["n_active"]
        return n, (time.perf_counter() - t0) / reps

    n, t_warm = run(threads, 1)                                   # also pages the code and the scene in
    reps = max(1, min(5, int(budget_s * 0.5 / max(t_warm, 1e-3))))
    n, t_all = run(threads, reps)
    reps1 = max(1, min(2, int(budget_s * 0.5 / max(t_all * threads * 0.6, 1e-3))))
    _, t_one = run(1, reps1)
    return dict(value=n / t_all, unit="active-voxels/s", cores=int(threads), kind="port", language="c++",
                single_thread_value=n / t_one, cpu_model=_cpu_model(), isa=lib.scn_cpu_isa().decode(),
                sample=f"{reps} full steps (rulebooks + fwd + bwd) of the same {n}-voxel scene on {threads} threads "
                       f"({t_all:.2f} s/step) and {reps1} on 1 thread ({t_one:.2f} s/step); C++17/OpenMP restatement of "
                       "the SparseConvNet CPU algorithm (oracle/scn_cpu_baseline.cpp, -O3 -march=native), NOT the "
                       "SparseConvNet binary (unavailable, SURVEY.md §8c)")
