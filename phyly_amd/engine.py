"""
phyly_amd.engine -- thin ctypes binding of the engine C-ABI (include/plk.h).

This is plumbing over libarbplf_amd.so: no numerics happen in Python.  Device
buffers may be passed as integers (raw device pointers, e.g. torch
`tensor.data_ptr()`), host buffers as numpy arrays.  The library is required:
importing this module raises if it has not been built, and creating an Engine
raises if no GPU is usable -- there is no CPU fallback.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PHYLY_AMD_LIB") or os.path.join(_HERE, "csrc", "libarbplf_amd.so")   # the override is for kernel timing experiments

HOST, DEVICE = 0, 1
ROOT_NONE, ROOT_CUSTOM, ROOT_UNIFORM, ROOT_EQUILIBRIUM = 1, 2, 3, 4
INFO_LL_KERNEL, INFO_STACK_SLOTS, INFO_PROGRAM_OPS, INFO_LL_KERNEL_NS, INFO_LL_TOTAL_NS, INFO_LL_KERNEL_NS_SUM, INFO_LL_KERNEL_COUNT, INFO_LL_VARIANT, INFO_PAIR_TABLES, INFO_LL_EXEC_FLOPS = range(10)
OPT_FORCE_GENERIC, OPT_SITE_CHUNK, OPT_FUSED_NS, OPT_FUSED_ASM, OPT_MFMA, OPT_UP_NODES, OPT_PAIR_TABLES, OPT_VEC_REG_STACK, OPT_MFMA_NS2 = 0, 1, 2, 3, 4, 5, 6, 7, 8
COEF_PRIOR, COEF_PRIOR_RATE_EDGE, COEF_PRIOR_RATE = 0, 1, 2
FIT_EM, FIT_LBFGS = 0, 1

_lib = None


class EngineError(RuntimeError):
    pass


def load_library():
    """dlopen libarbplf_amd.so (built by `make -C phyly_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "phyly_amd: %s is missing; build it with `make -C phyly_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    lib.plk_create.argtypes = [ctypes.POINTER(vp), ci]
    lib.plk_destroy.argtypes = [vp]
    lib.plk_destroy.restype = None
    lib.plk_last_error.argtypes = [vp]
    lib.plk_last_error.restype = ctypes.c_char_p
    lib.plk_create_error.restype = ctypes.c_char_p
    lib.plk_set_tree.argtypes = [vp, ci, vp, vp, vp]
    lib.plk_set_model.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, ci, vp]
    lib.plk_update_edge_rates.argtypes = [vp, vp]
    lib.plk_set_patterns_codes.argtypes = [vp, cl, vp, ci, ci, vp]
    lib.plk_set_patterns_dense.argtypes = [vp, cl, vp, ci]
    lib.plk_set_site_weights.argtypes = [vp, vp, ci]
    lib.plk_ll.argtypes = [vp, vp, ci, vp]
    lib.plk_ll_async.argtypes = [vp, vp, vp]
    lib.plk_sync.argtypes = [vp]
    lib.plk_set_stream.argtypes = [vp, vp]
    lib.plk_deriv.argtypes = [vp, vp, vp, vp]
    lib.plk_marginal.argtypes = [vp, vp, vp, vp]
    lib.plk_edge_expect.argtypes = [vp, vp, vp, ci, vp, vp, vp]
    lib.plk_edge_expect_multi.argtypes = [vp, ci, vp, vp, ci, vp, vp, vp]
    lib.plk_get_frechet_matrices.argtypes = [vp, vp, vp, ci, vp]
    lib.plk_fit_edge_rates.argtypes = [vp, ci, ci, ctypes.c_double, vp, vp, vp, ctypes.POINTER(ci), ctypes.POINTER(cl)]
    lib.plk_hess.argtypes = [vp, vp]
    lib.plk_get_transition_matrices.argtypes = [vp, vp]
    lib.plk_get_info.argtypes = [vp, ci, ctypes.POINTER(cl)]
    lib.plk_set_option.argtypes = [vp, ci, cl]
    lib.plk_comm_unique_id.argtypes = [vp]
    lib.plk_comm_available.argtypes = []
    lib.plk_comm_init.argtypes = [vp, ci, ci, vp]
    lib.plk_allreduce_sum_async.argtypes = [vp, vp, cl]
    lib.plk_comm_destroy.argtypes = [vp]
    _lib = lib
    return lib


def _ptr(x):
    """numpy array -> host pointer; int -> raw (device) pointer; None -> NULL."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return ctypes.c_void_p(x.ctypes.data)
    return ctypes.c_void_p(int(x))


def _f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def _i32(x):
    return np.ascontiguousarray(x, dtype=np.int32)


class Engine:
    """One engine = one GPU.  Mirrors the call order of the reference's query
    drivers: tree -> model (cross-site workspace) -> patterns -> ll/deriv/marginal."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        rc = self._lib.plk_create(ctypes.byref(self._h), int(device))
        if rc:
            self._h = None
            raise EngineError(self._lib.plk_create_error().decode())
        self.N = self.E = self.k = self.C = 0
        self.S = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.plk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise EngineError(self._lib.plk_last_error(self._h).decode())

    def set_tree(self, indptr, indices, preorder):
        indptr, indices, preorder = _i32(indptr), _i32(indices), _i32(preorder)
        self.N = len(preorder)
        self.E = self.N - 1
        self._check(self._lib.plk_set_tree(self._h, self.N, _ptr(indptr), _ptr(indices), _ptr(preorder)))

    def set_model(self, Qn, edge_rates_csr, cat_rates, cat_prior, root_mode, root_w=None, Qn_lo=None):
        Qn = _f64(Qn)
        Qn_lo = _f64(Qn_lo) if Qn_lo is not None else None
        self.k = Qn.shape[0]
        cat_rates, cat_prior = _f64(cat_rates), _f64(cat_prior)
        self.C = len(cat_rates)
        er = _f64(edge_rates_csr)
        rw = _f64(root_w) if root_w is not None else None
        self._check(self._lib.plk_set_model(self._h, self.k, self.C, _ptr(Qn), _ptr(Qn_lo), _ptr(er), _ptr(cat_rates),
                                            _ptr(cat_prior), int(root_mode), _ptr(rw)))

    def update_edge_rates(self, edge_rates_csr):
        er = _f64(edge_rates_csr)
        self._check(self._lib.plk_update_edge_rates(self._h, _ptr(er)))

    def set_patterns_codes(self, codes, defs, S=None, where=HOST):
        """codes: [N][S] uint8 numpy array, or a raw device pointer (then give S)."""
        defs = _f64(defs)
        if isinstance(codes, np.ndarray):
            codes = np.ascontiguousarray(codes, dtype=np.uint8)
            S = codes.shape[1]
        self.S = int(S)
        self._check(self._lib.plk_set_patterns_codes(self._h, self.S, _ptr(codes), int(where),
                                                     defs.shape[0], _ptr(defs)))

    def set_patterns_dense(self, B, S=None, where=HOST):
        """B: [N][k][S] float64 numpy array or raw device pointer."""
        if isinstance(B, np.ndarray):
            B = _f64(B)
            S = B.shape[2]
        self.S = int(S)
        self._check(self._lib.plk_set_patterns_dense(self._h, self.S, _ptr(B), int(where)))

    def set_site_weights(self, w, where=HOST):
        if isinstance(w, np.ndarray):
            w = _f64(w)
        self._check(self._lib.plk_set_site_weights(self._h, _ptr(w), int(where)))

    def ll(self, per_site=True, want_sum=True, out_device_ptr=None):
        """-> (site_ll ndarray or None, (hi, lo) or None)"""
        out = None
        where = HOST
        p = None
        if out_device_ptr is not None:
            p, where = ctypes.c_void_p(int(out_device_ptr)), DEVICE
        elif per_site:
            out = np.empty(self.S, dtype=np.float64)
            p = _ptr(out)
        s = np.zeros(2) if want_sum else None
        self._check(self._lib.plk_ll(self._h, p, where, _ptr(s)))
        return out, (tuple(s) if want_sum else None)

    def ll_async(self, sum_device_ptr=None, site_ll_device_ptr=None):
        """queue one ll evaluation on the engine's stream; outputs stay in device memory of the caller
        (raw device pointers: 2 doubles {hi, lo} for the sum, S doubles for the per-site values)"""
        self._check(self._lib.plk_ll_async(self._h, _ptr(site_ll_device_ptr), _ptr(sum_device_ptr)))

    def sync(self):
        self._check(self._lib.plk_sync(self._h))

    def set_stream(self, hip_stream):
        """hip_stream: raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or None for the engine's own"""
        self._check(self._lib.plk_set_stream(self._h, ctypes.c_void_p(int(hip_stream)) if hip_stream else None))

    def deriv(self, edge_mask=None, per_site=True, want_sums=True):
        mask = _i32(edge_mask) if edge_mask is not None else None
        out = np.zeros((self.S, self.E)) if per_site else None
        sums = np.zeros((self.E, 2)) if want_sums else None
        self._check(self._lib.plk_deriv(self._h, _ptr(mask), _ptr(out), _ptr(sums)))
        return out, sums

    def marginal(self, node_mask=None, per_site=True, want_sums=True):
        mask = _i32(node_mask) if node_mask is not None else None
        out = np.zeros((self.S, self.N, self.k)) if per_site else None
        sums = np.zeros((self.N, self.k, 2)) if want_sums else None
        self._check(self._lib.plk_marginal(self._h, _ptr(mask), _ptr(out), _ptr(sums)))
        return out, sums

    def edge_expect(self, L, coef_mode, L_lo=None, edge_mask=None, per_site=True, want_sums=True):
        """conditional edge expectations for the direction matrix L (see include/plk.h:plk_edge_expect)"""
        mask = _i32(edge_mask) if edge_mask is not None else None
        L = _f64(L)
        L_lo = _f64(L_lo) if L_lo is not None else None
        out = np.zeros((self.S, self.E)) if per_site else None
        sums = np.zeros((self.E, 2)) if want_sums else None
        self._check(self._lib.plk_edge_expect(self._h, _ptr(L), _ptr(L_lo), int(coef_mode), _ptr(mask), _ptr(out), _ptr(sums)))
        return out, sums

    def edge_expect_multi(self, Ls, coef_mode, Ls_lo=None, edge_mask=None, per_site=True, want_sums=True):
        """several direction matrices in one call: Ls [nL][k][k] -> ([S][nL][E], [nL][E][2])"""
        mask = _i32(edge_mask) if edge_mask is not None else None
        Ls = _f64(Ls)
        nL = Ls.shape[0]
        Ls_lo = _f64(Ls_lo) if Ls_lo is not None else None
        out = np.zeros((self.S, nL, self.E)) if per_site else None
        sums = np.zeros((nL, self.E, 2)) if want_sums else None
        self._check(self._lib.plk_edge_expect_multi(self._h, int(nL), _ptr(Ls), _ptr(Ls_lo), int(coef_mode), _ptr(mask),
                                                    _ptr(out), _ptr(sums)))
        return out, sums

    def frechet_matrices(self, L, coef_mode, L_lo=None):
        F = np.zeros((self.C, self.E, self.k, self.k))
        L = _f64(L)
        L_lo = _f64(L_lo) if L_lo is not None else None
        self._check(self._lib.plk_get_frechet_matrices(self._h, _ptr(L), _ptr(L_lo), int(coef_mode), _ptr(F)))
        return F

    def fit_edge_rates(self, rates, method=FIT_LBFGS, max_iter=100, ftol=1e-10, edge_mask=None):
        """maximum-likelihood edge rates with everything resident on the device
        -> (rates, ll_trace[:iters + 1], objective evaluations); see include/plk.h:plk_fit_edge_rates"""
        r = _f64(rates).copy()
        mask = _i32(edge_mask) if edge_mask is not None else None
        trace = np.zeros(int(max_iter) + 1)
        iters, evals = ctypes.c_int(0), ctypes.c_long(0)
        self._check(self._lib.plk_fit_edge_rates(self._h, int(method), int(max_iter), float(ftol), _ptr(mask), _ptr(r),
                                                 _ptr(trace), ctypes.byref(iters), ctypes.byref(evals)))
        return r, trace[:iters.value + 1].copy(), evals.value

    def hess(self):
        """[E][E] Hessian of the weighted log likelihood in the edge rates (CSR order), as hi + lo"""
        out = np.zeros((self.E, self.E, 2))
        self._check(self._lib.plk_hess(self._h, _ptr(out)))
        return out[..., 0] + out[..., 1]

    def transition_matrices(self):
        P = np.empty((self.C, self.E, self.k, self.k))
        self._check(self._lib.plk_get_transition_matrices(self._h, _ptr(P)))
        return P

    # -- one process per GPU: the reduction step on RCCL, queued by the engine on its own stream (include/plk.h)
    @staticmethod
    def comm_available():
        """True when the engine can load RCCL in this process (a local check, no communication)"""
        return bool(load_library().plk_comm_available())

    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id (bytes); made on rank 0 and handed to the other ranks by the caller"""
        lib = load_library()
        buf = (ctypes.c_ubyte * 128)()
        if lib.plk_comm_unique_id(buf):
            raise EngineError(lib.plk_create_error().decode())
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id):
        buf = (ctypes.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        self._check(self._lib.plk_comm_init(self._h, int(nranks), int(rank), buf))

    def allreduce_sum_async(self, device_ptr, count):
        self._check(self._lib.plk_allreduce_sum_async(self._h, ctypes.c_void_p(int(device_ptr)), int(count)))

    def comm_destroy(self):
        self._check(self._lib.plk_comm_destroy(self._h))

    def info(self, what):
        v = ctypes.c_long()
        self._check(self._lib.plk_get_info(self._h, int(what), ctypes.byref(v)))
        return v.value

    def set_option(self, option, value):
        self._check(self._lib.plk_set_option(self._h, int(option), int(value)))
