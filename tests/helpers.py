"""Shared helpers for the parity tests (test infrastructure; may use oracle/)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def oracle_model(oracle, workload, codes_host):
    """Oracle Model + prepared workspace for a synth.Workload and codes[N][S]."""
    md = workload.json_model(codes_host[:, :1])
    m = oracle.parse_model(md)
    w = oracle.prepare(m)
    return m, w


def oracle_site_ll(oracle, workload, codes_host, precise=1, nthreads=0):
    m, w = oracle_model(oracle, workload, codes_host)
    codes_sn = np.ascontiguousarray(codes_host.T)
    ll, used = oracle.site_ll(m, w, codes=codes_sn, defs=workload.defs, precise=precise, nthreads=nthreads)
    return ll


def rel_err(got, want, floor=1.0):
    got, want = np.asarray(got, float), np.asarray(want, float)
    return np.max(np.abs(got - want) / np.maximum(np.abs(want), floor))


def load_json(path):
    with open(path) as f:
        return json.load(f)
