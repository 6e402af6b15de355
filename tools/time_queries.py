#!/usr/bin/env python3
"""Whole-step wall time of plk_deriv and plk_marginal (site-summed outputs, patterns resident) on a BASELINE config:
  python tools/time_queries.py --config 4 --sites 1000000
prints one JSON line.  bench.py times ll and deriv; this adds the marginal query (and, with --hess-sites, the Hessian) for
DESIGN.md's table."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phyly_amd import synth, engine as E   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--sites", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--hess-sites", type=int, default=0, help="also time plk_hess (E + 1 second-order passes) on this many sites")
    a = ap.parse_args()
    wl = synth.Workload(a.config)
    S = a.sites or min(wl.default_S, 2_000_000)
    eng = E.Engine(0)
    wl.setup_engine(eng)
    eng.set_patterns_codes(wl.simulate(S), wl.defs)
    out = {"config": a.config, "sites": S}
    for name, fn in (("deriv", lambda: eng.deriv(per_site=False)), ("marginal", lambda: eng.marginal(per_site=False))):
        fn()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        dt = (time.perf_counter() - t0) / a.reps
        out[name] = {"ms_per_step": dt * 1e3, "sites_per_s": S / dt}
    if a.hess_sites:
        eng.set_patterns_codes(wl.simulate(a.hess_sites), wl.defs)
        eng.hess()
        t0 = time.perf_counter()
        eng.hess()
        dt = time.perf_counter() - t0
        out["hess"] = {"sites": a.hess_sites, "edges": wl.E, "ms": dt * 1e3, "site_passes_per_s": a.hess_sites * (wl.E + 1) / dt}
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
