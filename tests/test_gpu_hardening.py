"""GPU: the engine C-ABI refuses what its kernels could not survive, instead of launching:
pattern codes outside the definition table, handles that were never created or are already destroyed."""
import ctypes

import numpy as np
import pytest

from phyly_amd import synth
from phyly_amd.engine import Engine, EngineError, load_library, HOST

pytestmark = pytest.mark.gpu


def test_out_of_range_pattern_code_is_refused():
    wl = synth.Workload(3)
    eng = Engine(0)
    wl.setup_engine(eng)
    codes = wl.simulate(1000)
    eng.set_patterns_codes(codes, wl.defs)
    eng.ll()
    bad = codes.copy()
    bad[7, 901] = len(wl.defs)              # one past the last definition, on the last tile
    with pytest.raises(EngineError, match="pattern code"):
        eng.set_patterns_codes(bad, wl.defs)
    with pytest.raises(EngineError):        # no patterns are set after the refusal
        eng.ll()
    eng.set_patterns_codes(codes, wl.defs)
    eng.ll()
    eng.close()


def test_dead_and_foreign_handles_are_refused():
    lib = load_library()
    eng = Engine(0)
    h = ctypes.c_void_p(eng._h.value)
    eng.close()
    out = (ctypes.c_double * 2)()
    assert lib.plk_ll(h, None, HOST, out) == 2             # PLK_E_ARG, not a use after free
    assert lib.plk_last_error(h) == b"invalid engine handle"
    lib.plk_destroy(h)                                       # second destroy: no-op
    junk = ctypes.create_string_buffer(4096)
    assert lib.plk_ll(ctypes.cast(junk, ctypes.c_void_p), None, HOST, out) == 2
    eng2 = Engine(0)                                         # the library is still usable
    wl = synth.Workload(2)
    wl.setup_engine(eng2)
    eng2.set_patterns_codes(wl.simulate(300), wl.defs)
    ll, _ = eng2.ll()
    assert np.all(np.isfinite(ll))
    eng2.close()
