"""ctypes binding of include/barbay_hip.h and the `Engine` wrapper the host layer drives.

The product path loads ``lib/libbarbay_hip.so`` (built by ``__graft_entry__.build()`` with hipcc
for gfx950) and raises if it is missing or fails to load -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libbarbay_hip.so")

BB_MODEL = {"fitness": 0, "multienv": 1, "genotype": 2, "replicate": 3, "multienv_replicate": 4}
BB_OPT_TRUNCATED_ADAGRAD = 0
BB_OPT_DECAYED_ADAGRAD = 1
BB_COMM_ID_BYTES = 128
BB_P2P_HANDLE_BYTES = 64
BB_ERR_UNSUPPORTED = -4
BB_ERR_NONFINITE = -5

EXPORTS = [
    "bb_version", "bb_last_error", "bb_default_opts", "bb_create", "bb_destroy", "bb_num_latents",
    "bb_get_layout", "bb_init_meanfield", "bb_set_params", "bb_get_params", "bb_get_permutation", "bb_get_owned", "bb_run", "bb_run_profiled",
    "bb_get_posterior", "bb_elbo_grad", "bb_logdensity_grad", "bb_get_elbo_trace", "bb_debug_normals", "bb_debug_stamps", "bb_get_stats", "bb_kernel_name",
    "bb_comm_make_id", "bb_comm_init", "bb_step_moments", "bb_step_apply", "bb_hier_units", "bb_hier_fitness", "bb_p2p_export", "bb_p2p_import", "bb_p2p_selftest", "bb_p2p_enable",
]

_dp = C.POINTER(C.c_double)


class bb_prior(C.Structure):
    _fields_ = [("mean", _dp), ("std", _dp), ("n", C.c_int64)]


class bb_model_desc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n_rep", C.c_int32), ("n_neutral", C.c_int64), ("n_bc", C.c_int64),
        ("n_time", C.POINTER(C.c_int32)), ("counts", C.POINTER(C.c_int64)), ("totals", C.POINTER(C.c_int64)),
        ("n_env", C.c_int32), ("env_idx", C.POINTER(C.c_int32)),
        ("n_geno", C.c_int32), ("geno_idx", C.POINTER(C.c_int32)),
        ("s_pop_prior", bb_prior), ("logsigma_pop_prior", bb_prior), ("s_bc_prior", bb_prior),
        ("logsigma_bc_prior", bb_prior), ("loglambda_prior", bb_prior), ("logtau_prior", bb_prior),
        ("flags", C.c_int32),
    ]


class bb_advi_opts(C.Structure):
    _fields_ = [
        ("samples_per_step", C.c_int32), ("optimizer", C.c_int32), ("eta", C.c_double), ("tau", C.c_double),
        ("window", C.c_int32), ("resum_every", C.c_int32), ("pre", C.c_double), ("post", C.c_double),
        ("seed", C.c_uint64), ("device", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
        ("steps_per_graph", C.c_int32), ("elbo_every", C.c_int32), ("launch_mode", C.c_int32),
        ("n_devices", C.c_int32), ("device_ids", C.POINTER(C.c_int32)),
    ]


class bb_block_range(C.Structure):
    _fields_ = [("name", C.c_char * 24), ("lo", C.c_int64), ("hi", C.c_int64)]


class bb_stats(C.Structure):
    _fields_ = [
        ("n_latents", C.c_int64), ("n_moments", C.c_int64), ("steps_done", C.c_int64),
        ("shard_lo", C.c_int64), ("shard_hi", C.c_int64), ("bytes_per_step", C.c_int64),
        ("bytes_sample", C.c_int64), ("bytes_update", C.c_int64), ("last_run_ms", C.c_double),
        ("avg_sample_ms", C.c_double), ("avg_update_ms", C.c_double),
        ("n_blocks", C.c_int32), ("block_threads", C.c_int32), ("lds_bytes", C.c_int32),
        ("persistent_pairs", C.c_int32), ("launches_last_run", C.c_int32), ("resident_kernel", C.c_int32),
        ("geno_lo", C.c_int32), ("geno_hi", C.c_int32),
        ("device_bytes", C.c_int64), ("window_row", C.c_int64),
        ("rows_same_xcd", C.c_int32), ("reserved0", C.c_int32),
    ]


class BarBayHipError(RuntimeError):
    """Raised for any non-zero status of the C ABI (the reference throws ErrorException)."""


class BarBayNonFinite(BarBayHipError):
    """bb_run took its steps but the variational parameters went NaN / Inf (BB_ERR_NONFINITE); the state can still be read."""


def _declare(lib: C.CDLL) -> C.CDLL:
    vp = C.c_void_p
    lib.bb_version.restype = C.c_char_p
    lib.bb_last_error.restype = C.c_char_p
    lib.bb_default_opts.argtypes = [C.POINTER(bb_advi_opts)]
    lib.bb_default_opts.restype = None
    lib.bb_create.argtypes = [C.POINTER(bb_model_desc), C.POINTER(bb_advi_opts), C.POINTER(vp)]
    lib.bb_destroy.argtypes = [vp]
    lib.bb_destroy.restype = None
    lib.bb_num_latents.argtypes = [vp]
    lib.bb_num_latents.restype = C.c_int64
    lib.bb_get_layout.argtypes = [vp, C.POINTER(bb_block_range), C.POINTER(C.c_int32)]
    lib.bb_init_meanfield.argtypes = [vp]
    lib.bb_set_params.argtypes = [vp, _dp, _dp]
    lib.bb_get_params.argtypes = [vp, _dp, _dp]
    if hasattr(lib, "bb_get_permutation"):      # (A/B builds of older sources, tools/xp.py)
        lib.bb_get_permutation.argtypes = [vp, C.POINTER(C.c_int64)]
    if hasattr(lib, "bb_get_owned"):
        lib.bb_get_owned.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.bb_run.argtypes = [vp, C.c_int64]
    lib.bb_run_profiled.argtypes = [vp, C.c_int64]
    lib.bb_get_posterior.argtypes = [vp, _dp, _dp]
    lib.bb_elbo_grad.argtypes = [vp, _dp, _dp, _dp, C.c_int32, _dp, _dp, _dp]
    lib.bb_logdensity_grad.argtypes = [vp, _dp, _dp, _dp]
    lib.bb_get_elbo_trace.argtypes = [vp, C.c_int64, C.c_int64, _dp]
    lib.bb_debug_normals.argtypes = [vp, C.c_int64, C.c_uint32, C.c_int64, C.c_int64, _dp]
    lib.bb_get_stats.argtypes = [vp, C.POINTER(bb_stats)]
    if hasattr(lib, "bb_kernel_name"):          # (A/B builds of older sources, tools/xp.py)
        lib.bb_kernel_name.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.bb_debug_stamps.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int64]
    lib.bb_comm_make_id.argtypes = [C.c_void_p]
    lib.bb_comm_init.argtypes = [vp, C.c_void_p]
    lib.bb_step_moments.argtypes = [vp, _dp]
    lib.bb_step_apply.argtypes = [vp, _dp]
    lib.bb_p2p_export.argtypes = [vp, C.c_void_p]
    lib.bb_p2p_import.argtypes = [vp, C.c_void_p]
    lib.bb_p2p_selftest.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.bb_p2p_enable.argtypes = [vp, C.c_int32]
    lib.bb_hier_units.argtypes = [vp]
    lib.bb_hier_units.restype = C.c_int64
    lib.bb_hier_fitness.argtypes = [vp, C.c_int32, C.c_uint64, _dp, _dp]
    return lib


_LIB: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load the HIP engine.  No fallback: a missing/unloadable library is an error."""
    global _LIB
    if path is None and _LIB is not None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise BarBayHipError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:
        import torch  # noqa: F401  (loads its libamdhip64.so.7 first so both share one HIP runtime)
    except Exception:
        pass
    lib = _declare(C.CDLL(p))
    if path is None:
        _LIB = lib
    return lib


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a: np.ndarray, ty=_dp):
    return a.ctypes.data_as(ty)


class Engine:
    """One model instance on one GPU (`bb_handle`).

    counts: list (one per replicate) of T_r x B int64 arrays, neutrals first.
    priors: name -> (mean, std), floats (Vector form) or 1-D arrays (Matrix form); names
    s_pop_prior, logsigma_pop_prior, s_bc_prior, logsigma_bc_prior, loglambda_prior, logtau_prior.
    """

    def __init__(self, kind: str, counts: Sequence[np.ndarray], n_neutral: int, n_bc: int, *,
                 totals: Optional[Sequence[np.ndarray]] = None, env_idx=None, geno_idx=None,
                 priors: Optional[Dict[str, Tuple[object, object]]] = None,
                 samples_per_step: int = 1, optimizer: str = "TruncatedADAGrad", eta: float = 0.1,
                 tau: float = 40.0, window: int = 100, resum_every: int = 0, pre: float = 1.0,
                 post: float = 0.9, seed: int = 0, device: int = 0, rank: int = 0, world_size: int = 1,
                 steps_per_graph: int = 0, elbo_every: int = 0, launch_mode: int = 0, ragged_method: bool = False,
                 n_devices: int = 1, device_ids: Optional[Sequence[int]] = None, _lib: Optional[C.CDLL] = None):
        self._lib = _lib if _lib is not None else load_library()
        self._h = C.c_void_p()
        self.kind = kind
        counts = [np.asarray(c, dtype=np.int64) for c in counts]
        if totals is None:
            totals = [c.sum(axis=1) for c in counts]
        totals = [np.asarray(t, dtype=np.int64) for t in totals]
        keep: List[np.ndarray] = []   # arrays the descriptor points into, alive for the bb_create call

        def hold(a):
            keep.append(a)
            return a

        md = bb_model_desc()
        md.kind = BB_MODEL[kind]
        md.n_rep = len(counts)
        md.n_neutral = int(n_neutral)
        md.flags = 1 if ragged_method else 0      # BB_FLAG_RAGGED_METHOD
        md.n_bc = int(n_bc)
        nt = hold(np.asarray([c.shape[0] for c in counts], dtype=np.int32))
        md.n_time = _ptr(nt, C.POINTER(C.c_int32))
        # Julia column-major T x B (t fastest) == C-order of the transpose
        cflat = hold(np.concatenate([np.ascontiguousarray(c.T).reshape(-1) for c in counts]).astype(np.int64))
        tflat = hold(np.concatenate(totals).astype(np.int64))
        md.counts = _ptr(cflat, C.POINTER(C.c_int64))
        md.totals = _ptr(tflat, C.POINTER(C.c_int64))
        if env_idx is not None:
            if isinstance(env_idx, (list, tuple)) and len(env_idx) and np.ndim(env_idx[0]) == 1:
                env_idx = np.concatenate([np.asarray(x) for x in env_idx])     # one list per replicate -> replicate-major
            e = hold(np.ascontiguousarray(env_idx, dtype=np.int32))
            md.n_env = int(e.max()) + 1
            md.env_idx = _ptr(e, C.POINTER(C.c_int32))
        if geno_idx is not None:
            g = hold(np.ascontiguousarray(geno_idx, dtype=np.int32))
            md.n_geno = int(g.max()) + 1
            md.geno_idx = _ptr(g, C.POINTER(C.c_int32))
        for name, (mean, std) in (priors or {}).items():
            m = hold(np.atleast_1d(_f64(mean)))
            s = hold(np.atleast_1d(_f64(std)))
            if m.shape != s.shape or m.ndim != 1:
                raise BarBayHipError(f"{name}: mean/std must be scalars or equal-length vectors")
            p = getattr(md, name)
            p.mean, p.std, p.n = _ptr(m), _ptr(s), m.shape[0]
        o = bb_advi_opts()
        self._lib.bb_default_opts(C.byref(o))
        o.samples_per_step = samples_per_step
        o.optimizer = {"TruncatedADAGrad": 0, "DecayedADAGrad": 1}[optimizer]
        o.eta, o.tau, o.window, o.resum_every, o.pre, o.post = eta, tau, window, resum_every, pre, post
        o.seed, o.device, o.rank, o.world_size = seed, device, rank, world_size
        o.steps_per_graph, o.elbo_every, o.launch_mode = steps_per_graph, elbo_every, launch_mode
        if device_ids is not None:
            n_devices = len(device_ids)
            ids = hold(np.ascontiguousarray(device_ids, dtype=np.int32))
            o.device_ids = _ptr(ids, C.POINTER(C.c_int32))
        o.n_devices = int(n_devices)
        self._check(self._lib.bb_create(C.byref(md), C.byref(o), C.byref(self._h)))
        self.D = int(self._lib.bb_num_latents(self._h))
        self.samples_per_step = samples_per_step
        self.world_size, self.rank = world_size, rank

    # -- plumbing -------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc == BB_ERR_NONFINITE:
            raise BarBayNonFinite(f"barbay_hip error {rc}: {self._lib.bb_last_error().decode()}")
        if rc != 0:
            raise BarBayHipError(f"barbay_hip error {rc}: {self._lib.bb_last_error().decode()}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.bb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- API ------------------------------------------------------------------------------------
    def layout(self) -> List[Tuple[str, int, int]]:
        blocks = (bb_block_range * 8)()
        n = C.c_int32()
        self._check(self._lib.bb_get_layout(self._h, blocks, C.byref(n)))
        return [(blocks[i].name.decode(), int(blocks[i].lo), int(blocks[i].hi)) for i in range(n.value)]

    def init_meanfield(self):
        self._check(self._lib.bb_init_meanfield(self._h))

    def set_params(self, mu, omega):
        mu, omega = _f64(mu), _f64(omega)
        assert mu.shape == (self.D,) and omega.shape == (self.D,)
        self._check(self._lib.bb_set_params(self._h, _ptr(mu), _ptr(omega)))

    def get_params(self) -> Tuple[np.ndarray, np.ndarray]:
        mu, om = np.empty(self.D), np.empty(self.D)
        self._check(self._lib.bb_get_params(self._h, _ptr(mu), _ptr(om)))
        return mu, om

    def permutation(self) -> np.ndarray:
        """caller_index[i] of the handle's internal latent i (identity unless the genotype model's mutants were regrouped)."""
        out = np.empty(self.D, dtype=np.int64)
        self._check(self._lib.bb_get_permutation(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def owned(self) -> np.ndarray:
        """The caller's flat indices of the latents this handle owns on a sharded run (`bb_get_owned`): its barcodes' latents and,
        genotype model, theta of its own genotypes -- not the replicated global blocks."""
        out = np.empty(self.D, dtype=np.int64)
        n = C.c_int64(0)
        self._check(self._lib.bb_get_owned(self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(n)))
        return out[:n.value].copy()

    def run(self, n_steps: int):
        self._check(self._lib.bb_run(self._h, int(n_steps)))

    def run_profiled(self, n_steps: int):
        self._check(self._lib.bb_run_profiled(self._h, int(n_steps)))

    def posterior(self) -> Tuple[np.ndarray, np.ndarray]:
        m, s = np.empty(self.D), np.empty(self.D)
        self._check(self._lib.bb_get_posterior(self._h, _ptr(m), _ptr(s)))
        return m, s

    def elbo_grad(self, mu, omega, eps=None, n_samples: Optional[int] = None):
        mu, omega = _f64(mu), _f64(omega)
        if eps is not None:
            eps = np.atleast_2d(_f64(eps))
            assert eps.shape[1] == self.D
            n_samples = eps.shape[0]
        n_samples = n_samples or 1
        gm, go = np.empty(self.D), np.empty(self.D)
        elbo = C.c_double()
        self._check(self._lib.bb_elbo_grad(self._h, _ptr(mu), _ptr(omega), _ptr(eps) if eps is not None else None,
                                           n_samples, C.byref(elbo), _ptr(gm), _ptr(go)))
        return elbo.value, gm, go

    def elbo_trace(self, first_step: int, n: int) -> np.ndarray:
        out = np.empty(n)
        self._check(self._lib.bb_get_elbo_trace(self._h, first_step, n, _ptr(out)))
        return out

    def normals(self, step: int, stream: int, lo: int, hi: int) -> np.ndarray:
        out = np.empty(hi - lo)
        self._check(self._lib.bb_debug_normals(self._h, step, stream, lo, hi, _ptr(out)))
        return out

    def stamps(self, per_wave: bool = False) -> np.ndarray:
        """Diagnostic build only: [tiles][32] block stamps, or with per_wave [tiles][4 events][16 waves]."""
        nb = int(self.stats()["n_blocks"])
        rows = np.zeros(1, dtype=np.uint64)
        self._check(self._lib.bb_debug_stamps(self._h, rows.ctypes.data_as(C.POINTER(C.c_uint64)), -1))
        rows = int(rows[0])
        raw = np.zeros(rows * 96, dtype=np.uint64)
        self._check(self._lib.bb_debug_stamps(self._h, raw.ctypes.data_as(C.POINTER(C.c_uint64)), raw.size))
        if not per_wave:
            return raw[:nb * 32].reshape(nb, 32).copy()
        return raw[rows * 32:rows * 32 + nb * 64].reshape(nb, 4, 16).copy()

    def stats(self) -> Dict[str, float]:
        s = bb_stats()
        self._check(self._lib.bb_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in bb_stats._fields_}

    def kernel_name(self) -> str:
        """The kernel `run` launches, as text: the template instance, e.g. 'k_res<0,1,1024,false,8,false,false>' (`bb_kernel_name`)."""
        buf = C.create_string_buffer(128)
        self._check(self._lib.bb_kernel_name(self._h, buf, 128))
        return buf.value.decode()

    def logdensity_grad(self, z) -> Tuple[float, np.ndarray]:
        """log p(data, z) and its gradient at a point of the flat latent vector (`bb_logdensity_grad`)."""
        z = np.ascontiguousarray(z, dtype=np.float64)
        assert z.shape == (self.D,)
        lp = C.c_double(0.0)
        g = np.empty(self.D)
        self._check(self._lib.bb_logdensity_grad(self._h, _ptr(z), C.byref(lp), _ptr(g)))
        return lp.value, g

    def hier_units(self) -> int:
        """Length of the theta_tilde block (0 for the non-hierarchical models)."""
        return int(self._lib.bb_hier_units(self._h))

    def hier_fitness(self, n_samples: int = 10_000, seed: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        """Device-side `process_hierarchical_samples!`: (median, std) of theta + exp(logtau) * theta_tilde per unit."""
        n = int(self._lib.bb_hier_units(self._h))
        med, sd = np.empty(n), np.empty(n)
        self._check(self._lib.bb_hier_fitness(self._h, n_samples, seed, _ptr(med), _ptr(sd)))
        return med, sd

    # ---- cross-GPU leg of the resident launch (include/barbay_hip.h, bb_p2p_*) ----
    def p2p_export(self) -> bytes:
        buf = C.create_string_buffer(BB_P2P_HANDLE_BYTES)
        self._check(self._lib.bb_p2p_export(self._h, buf))
        return buf.raw

    def p2p_import(self, handles: Sequence[bytes]):
        blob = b"".join(handles)
        assert len(blob) == BB_P2P_HANDLE_BYTES * len(handles)
        buf = C.create_string_buffer(blob, len(blob))
        self._check(self._lib.bb_p2p_import(self._h, buf))

    def p2p_selftest(self) -> bool:
        ok = C.c_int32(0)
        self._check(self._lib.bb_p2p_selftest(self._h, C.byref(ok)))
        return bool(ok.value)

    def p2p_enable(self, on: bool) -> bool:
        """True if the resident launch is now (on) / no longer (off) in use; False if this shard cannot use it."""
        rc = self._lib.bb_p2p_enable(self._h, 1 if on else 0)
        if rc == BB_ERR_UNSUPPORTED:
            return False
        self._check(rc)
        return True

    def make_comm_id(self) -> bytes:
        buf = C.create_string_buffer(BB_COMM_ID_BYTES)
        self._check(self._lib.bb_comm_make_id(buf))
        return buf.raw

    def comm_init(self, comm_id: bytes):
        assert len(comm_id) == BB_COMM_ID_BYTES
        buf = C.create_string_buffer(comm_id, BB_COMM_ID_BYTES)
        self._check(self._lib.bb_comm_init(self._h, buf))

    def step_moments(self) -> np.ndarray:
        k = int(self.stats()["n_moments"])
        out = np.empty(k)
        self._check(self._lib.bb_step_moments(self._h, _ptr(out)))
        return out

    def step_apply(self, total: np.ndarray):
        total = _f64(total)
        self._check(self._lib.bb_step_apply(self._h, _ptr(total)))
