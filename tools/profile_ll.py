#!/usr/bin/env python3
"""Minimal driver for rocprofv3 counter passes: runs the ll hot path on a BASELINE
config with host-generated pattern codes and no torch kernels, so that PMC
collection only sees the engine's own kernels.

  rocprofv3 --kernel-trace --pmc ... --kernel-include-regex 'k_ll' -- python3 tools/profile_ll.py --config 3 --sites 2000000
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--sites", type=int, default=2_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--kernel", choices=["auto", "generic"], default="auto")
    ap.add_argument("--what", choices=["ll", "deriv", "marginal", "dwell", "em", "fit"], default="ll")
    ap.add_argument("--fused-ns", type=int, default=0)
    args = ap.parse_args()
    from phyly_amd import synth, engine as E
    wl = synth.Workload(args.config)
    eng = E.Engine(0)
    wl.setup_engine(eng)
    if args.kernel == "generic":
        eng.set_option(E.OPT_FORCE_GENERIC, 1)
    if args.fused_ns:
        eng.set_option(E.OPT_FUSED_NS, args.fused_ns)
    # realistic codes for a block of sites, tiled to the requested size
    base = wl.simulate(min(args.sites, 4096))
    reps = -(-args.sites // base.shape[1])
    codes = np.ascontiguousarray(np.tile(base, (1, reps))[:, :args.sites])
    eng.set_patterns_codes(codes, wl.defs)
    for i in range(args.steps):
        eng.update_edge_rates(wl.edge_rates_csr)
        t0 = time.perf_counter()
        if args.what == "ll":
            _, s = eng.ll(per_site=False)
            extra = "kernel %.3f ms" % (eng.info(E.INFO_LL_KERNEL_NS) * 1e-6)
        elif args.what == "deriv":
            _, s = eng.deriv(per_site=False)
            extra = ""
        elif args.what == "dwell":
            # one state-aggregated dwell query (site-summed): Frechet build + down/up pass
            _, s = eng.edge_expect(np.diag(np.arange(1.0, wl.k + 1)), E.COEF_PRIOR, per_site=False)
            extra = ""
        elif args.what == "fit":
            # device-resident L-BFGS / EM fits from perturbed rates; reports iterations per second
            start = wl.edge_rates_csr * np.exp(np.random.default_rng(i).uniform(-0.7, 0.7, wl.E))
            for meth, name in ((E.FIT_LBFGS, "lbfgs"), (E.FIT_EM, "em")):
                t1 = time.perf_counter()
                _, tr, ev = eng.fit_edge_rates(start, method=meth, max_iter=25, ftol=1e-12)
                d1 = time.perf_counter() - t1
                print("  %s: %d iterations, %d ll evaluations, %.1f ms/iteration, ll %.6f -> %.6f"
                      % (name, len(tr) - 1, ev, d1 * 1e3 / max(1, len(tr) - 1), tr[0], tr[-1]), flush=True)
            extra = ""
        elif args.what == "em":
            # one em-update: two edge-expectation passes (transitions, exit-rate dwell)
            Qn = wl.prepare()["Qn"]
            _, t = eng.edge_expect(Qn * (1 - np.eye(wl.k)), E.COEF_PRIOR_RATE, per_site=False)
            _, d = eng.edge_expect(-np.diag(np.diag(Qn)), E.COEF_PRIOR_RATE, per_site=False)
            extra = ""
        else:
            _, s = eng.marginal(per_site=False)
            extra = ""
        dt = time.perf_counter() - t0
        print("step %d: %.3f ms wall, %.1f Msites/s %s" % (i, dt * 1e3, args.sites / dt / 1e6, extra), flush=True)


if __name__ == "__main__":
    main()
