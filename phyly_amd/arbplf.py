"""
phyly_amd.arbplf -- Python 3 equivalent of the reference's `arbplf` extension
module (src/arbplf.c:209-250, :521-546): each function maps one JSON string to
one JSON string and raises RuntimeError("arbplf likelihood error") on any
failure (diagnostics go to stderr, as in the reference).

The work is done by libarbplf_amd.so through its string C-ABI
(include/arbplf.h); nothing is computed in Python and there is no CPU path.
"""
import ctypes

from . import engine as _engine

_fns = {}


def _fn(name):
    f = _fns.get(name)
    if f is None:
        lib = _engine.load_library()
        f = getattr(lib, name)
        f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        f.restype = ctypes.c_void_p          # malloc'd char*; freed below
        _fns[name] = f
    return f


_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]
_libc.free.restype = None


def _call(name, s):
    if isinstance(s, str):
        s = s.encode("utf-8")
    elif not isinstance(s, (bytes, bytearray)):
        raise TypeError("%s() argument must be str, not %s" % (name, type(s).__name__))
    retcode = ctypes.c_int(0)
    p = _fn(name + "_string")(None, bytes(s), ctypes.byref(retcode))
    try:
        if retcode.value != 0 or not p:
            raise RuntimeError("arbplf likelihood error")
        return ctypes.string_at(p).decode("utf-8")
    finally:
        if p:
            _libc.free(p)


def arbplf_ll(s):
    """log likelihood (reference: arbplf_ll_run, src/arbplfll.c:291-323)"""
    return _call("arbplf_ll", s)


def arbplf_deriv(s):
    """d log likelihood / d edge rate coefficient (src/arbplfderiv.c:496-531)"""
    return _call("arbplf_deriv", s)


def arbplf_marginal(s):
    """marginal state distributions at nodes (src/arbplfmarginal.c:408-446)"""
    return _call("arbplf_marginal", s)


def arbplf_dwell(s):
    """conditional expected dwell proportions per edge and state (src/arbplfdwell.c:569-610)"""
    return _call("arbplf_dwell", s)


def arbplf_trans(s):
    """conditional expected labelled transition counts per edge (src/arbplftrans.c:617-660)"""
    return _call("arbplf_trans", s)


def arbplf_em_update(s):
    """one EM update of the edge rate coefficients (src/arbplfem.c:548-588)"""
    return _call("arbplf_em_update", s)


def arbplf_hess(s):
    """Hessian of the log likelihood in the edge rate coefficients (src/arbplfhess.c:1279-1343), fp64"""
    return _call("arbplf_hess", s)


def _out_of_scope(name):
    def f(s):
        raise RuntimeError("arbplf likelihood error: %s is outside the MI355X hot path of this build" % name)
    f.__name__ = name
    return f


# the reference module's other entry points (src/arbplf.c:521-534) are out of scope
for _name in ("arbplf_inv_hess",
              "arbplf_newton_delta", "arbplf_newton_update", "arbplf_newton_refine"):
    globals()[_name] = _out_of_scope(_name)
