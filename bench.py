#!/usr/bin/env python3
"""
bench.py -- sites/sec of the arbplf-ll hot path on MI355X (BASELINE.json metric).

One "step" = one full pass of the ll path over the site patterns: exp(Q r t)
for every (category, edge) (K1, which for k = 4 also writes the matrix stream
and the tip tables), the pruning kernel over all sites with category mixing and
log (K2+K3), the weighted site reduction (K7) and, for N > 1, one RCCL
all-reduce of the double-double log-likelihood sum (X1).  Pattern codes are
resident in HBM before the timed region.  Steps are queued on the framework's
stream (plk_ll_async): the {hi, lo} sum stays in device memory, the all-reduce
works on it there, and the host reads the last sum after the timed region; the
traversal kernel's time is taken from HIP events on that same stream around
every launch of the timed region.

Site sharding (the reference's only cross-site step is the axis
reduction of src/ndaccum.c:198-254 after the site loop of src/arbplfll.c:139-170):
  --scaling strong (default)  the metric's own split: the config's S sites (10M at
      config 3; --sites overrides the TOTAL) are cut into shard.shard_range blocks,
      one per rank; `value` = total sites x steps / time.
  --scaling weak              every rank holds --sites patterns (its block of an
      N x --sites alignment).

After the ll region a short edge-gradient leg (arbplf-deriv: down + up pass,
site-summed, 2E doubles all-reduced for N > 1) is timed and reported under
"deriv"; it is not part of `value`.

Launch: `python bench.py` (1 GPU) or
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`.
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md
FP64_PEAK = 78.6e12        # flop/s, fp64 vector = fp64 matrix dense peak

# sources a kernel's measured HBM traffic depends on (profiles/traffic_*.json record their git blob hashes)
KERNEL_SOURCES = {
    "k_ll_fused4": ["plk_fused4_asm.h", "plk_fused4_v4.h", "plk_fused4_v4_asm.h", "plk_fused4.h", "plk_program.h"],
    "k_ll_vec": ["plk_vec.h", "plk_vec_matvec_asm.h", "plk_program.h"],
    "k_ll_mfma": ["plk_mfma.h"],
    "k_ll_generic": ["plk_engine.hip"],
    "deriv4": ["plk_updown4.h", "plk_down4_asm.h", "plk_program.h"],
    "deriv_vec": ["plk_updown_vec.h", "plk_vec_matvec_asm.h"],
    "deriv_mfma": ["plk_mfma_updown.h"],
}


def git_blob_hash(path):
    """the hash `git hash-object` prints for the file"""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_hashes(key):
    out = {}
    for f in KERNEL_SOURCES.get(key, []):
        p = os.path.join(ROOT, "phyly_amd", "csrc", f)
        if os.path.exists(p):
            out[f] = git_blob_hash(p)
    return out


def load_traffic(path, key, sites):
    """HBM bytes per launch from a PMC file (tools/pmc_traffic.py), or None unless the file was measured at this site
    count on exactly the kernel sources of this tree (git blob hashes recorded in the file)."""
    try:
        tj = json.load(open(path))
    except Exception:
        return None
    if int(tj.get("sites", -1)) != int(sites):
        return None
    want = source_hashes(key)
    if not want or tj.get("source_blobs") != want:
        return None
    return tj


def host_cpu_share():
    """threads this process may really use: min(affinity mask, cgroup CPU quota); plus what the box has"""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    threads = aff if quota is None else max(1, min(aff, int(math.ceil(quota))))
    phys = None
    try:
        cores = set()
        phys_id = core_id = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys_id = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core_id = line.split(":")[1].strip()
            elif not line.strip():
                if phys_id is not None and core_id is not None:
                    cores.add((phys_id, core_id))
                phys_id = core_id = None
        phys = len(cores) or None
    except Exception:
        pass
    return dict(threads=threads, affinity=aff, cgroup_quota=quota, logical=os.cpu_count(), physical=phys)


def cpu_baseline(workload, sample_sites, seconds=15.0):
    """Oracle 'port' (oracle/plf_core.c, double build, OpenMP) on the host cores this process is entitled to.
    Only the C call is timed (inputs are contiguous arrays made before the clock starts)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import arbplf_oracle as O
    O.build()
    share = host_cpu_share()
    nt = share["threads"]
    codes0 = workload.simulate(256)
    md = workload.json_model(codes0[:, :1])
    m = O.parse_model(md)
    w = O.prepare(m)
    # calibrate on a probe (second call: threads already started), then size the sample for ~`seconds`
    probe = min(sample_sites, 50000)
    codes = np.ascontiguousarray(workload.simulate(probe).T)
    O.site_ll(m, w, codes=codes, defs=workload.defs, precise=0, nthreads=nt)
    t0 = time.perf_counter()
    O.site_ll(m, w, codes=codes, defs=workload.defs, precise=0, nthreads=nt)
    dt = max(time.perf_counter() - t0, 1e-4)
    n = int(min(sample_sites, max(probe, probe * seconds / dt)))
    n = max(256, n // 256 * 256)
    reps = -(-n // probe)
    codes = np.ascontiguousarray(np.tile(codes, (reps, 1))[:n])   # the probe block repeated: same per-site work
    ll, used, dt = O.site_ll_timed(m, w, codes, workload.defs, precise=0, nthreads=nt)
    # the same port on ONE host thread, on a sample sized for a few seconds
    n1 = max(256, min(n, int(n * 4.0 / max(dt, 1e-3) / max(int(used), 1))) // 256 * 256)
    _, _, dt1 = O.site_ll_timed(m, w, codes[:n1], workload.defs, precise=0, nthreads=1)
    # checker values for the bench's own cross-check: long double build on the first sites of the alignment
    ncheck = min(probe, 4096)
    ll_check, _ = O.site_ll(m, w, codes=codes[:ncheck], defs=workload.defs, precise=1, nthreads=nt)
    return dict(value=n / dt, unit="sites/s", cores=int(used), kind="port",
                sample="%d sites (the first %d sites of the same synthetic alignment, repeated), double-precision "
                       "oracle port, %d OpenMP threads, %.1f s in the C call" % (n, probe, used, dt),
                host=share,
                one_thread=dict(value=n1 / dt1, unit="sites/s", cores=1, sample="%d sites, %.1f s" % (n1, dt1)),
                parallel_efficiency=(n / dt) / (n1 / dt1) / max(int(used), 1)), ll_check


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config index (2..5)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--sites", type=int, default=0, help="strong: total sites of the job (default: the config's S); weak: sites per GPU")
    ap.add_argument("--kernel", choices=["auto", "generic"], default="auto")
    ap.add_argument("--fused-ns", type=int, default=0, help="sites per lane of the fused kernel (0 = auto)")
    ap.add_argument("--engine-option", action="append", default=[], metavar="ID=VALUE",
                    help="plk_set_option(ID, VALUE) before the run (kernel experiments; include/plk.h lists the options)")
    ap.add_argument("--categories", type=int, default=0, help="override the number of Gamma categories (experiments; not the metric's workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist", action="store_true", help="initialise torch.distributed (nccl) and all-reduce also when launched as one process")
    ap.add_argument("--torch-allreduce", action="store_true", help="N > 1: torch.distributed.all_reduce instead of the engine's own RCCL call")
    ap.add_argument("--sync-allreduce", action="store_true", help="N > 1: block the compute stream on every all-reduce (no overlap with the next step)")
    ap.add_argument("--deriv-steps", type=int, default=3, help="steps of the edge-gradient leg (0 = skip)")
    ap.add_argument("--deriv-sites", type=int, default=0, help="sites per GPU of the edge-gradient leg (default: min(block, 2M))")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from phyly_amd import synth, shard
    from phyly_amd import engine as E

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.dist
    if use_dist:
        # stdout carries the one JSON line of rank 0 and nothing else: RCCL prints its version banner with printf when
        # NCCL_DEBUG is set (the GPU boxes export NCCL_DEBUG=VERSION), so file descriptor 1 points at stderr while the
        # communicators are set up and the steps run, and is put back just before the line is printed
        sys.stdout.flush()
        saved_stdout_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)

    def restore_stdout():
        if use_dist:
            sys.stdout.flush()
            os.dup2(saved_stdout_fd, 1)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    wl = synth.Workload(args.config)
    if args.categories and wl.mixture is not None:
        wl.mixture = dict(wl.mixture, gamma_categories=args.categories)
        wl.name += " [categories overridden: %d]" % args.categories
    if args.scaling == "strong":
        S_total = args.sites or wl.default_S
        if S_total < world:
            raise SystemExit("bench.py: fewer sites than ranks")
        s0, s1 = shard.shard_range(S_total, rank, world)
    else:
        per = args.sites or wl.default_S
        S_total = per * world
        s0, s1 = rank * per, (rank + 1) * per
    S = s1 - s0                                            # this rank's block [s0, s1) of the alignment
    eng = E.Engine(local_rank)
    wl.setup_engine(eng)
    if args.kernel == "generic":
        eng.set_option(E.OPT_FORCE_GENERIC, 1)
    if args.fused_ns:
        eng.set_option(E.OPT_FUSED_NS, args.fused_ns)
    for kv in args.engine_option:
        oid, val = kv.split("=")
        eng.set_option(int(oid), int(val))

    # resident patterns: this rank's block of the alignment, generated on the GPU
    chunk = 1 << 20
    codes = torch.empty((wl.N, S), dtype=torch.uint8, device=dev)
    for off in range(0, S, chunk):
        n = min(chunk, S - off)
        codes[:, off:off + n] = wl.simulate(n, site0=s0 + off, device=dev)
    torch.cuda.synchronize()
    eng.set_patterns_codes(codes.data_ptr(), wl.defs, S=S, where=E.DEVICE)
    del codes
    torch.cuda.empty_cache()

    # the engine issues its kernels on torch's current stream: the all-reduce and the fences below are ordered with them
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    # One {hi, lo} result per step in a small ring.  With N > 1 the all-reduce of step i is issued asynchronously
    # (RCCL's own stream, ordered after the step's kernels by c10d) so that it overlaps the kernels of step i + 1: the
    # evaluations are independent, nothing in step i + 1 reads the reduced sum of step i.  Every step still does its
    # one all-reduce; all of them are waited for before the clock stops.  --sync-allreduce blocks the compute stream on
    # each reduction instead (the latency-exposed figure; DESIGN.md section 5 quotes both).
    # N > 1: the reduction step.  Default: RCCL called by the engine itself on the stream its kernels run on
    # (plk_allreduce_sum_async, include/plk.h: one C call per step, no framework work object, no second stream); the RCCL
    # id travels over torch.distributed once, and a test reduction is compared with torch's before the path is trusted.
    # If anything in that fails on any rank, every rank falls back to torch.distributed.all_reduce (--torch-allreduce
    # forces it).
    native = False
    if use_dist and not args.torch_allreduce:
        def agree(ok):
            """every rank calls this: True only if all ranks say so (so that no rank is left alone in a collective)"""
            f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            return bool(f.item())

        ok = agree(E.Engine.comm_available())          # local: can this process load RCCL at all?
        if ok:
            idb = bytes(128)
            if rank == 0:
                try:
                    idb = E.Engine.comm_unique_id()
                except Exception as exc:      # noqa: BLE001
                    sys.stderr.write("bench.py: ncclGetUniqueId failed (%s)\n" % exc)
                    idb = None
            idt = torch.tensor(list(idb or bytes(128)), dtype=torch.uint8, device=dev)
            dist.broadcast(idt, 0)
            ok = agree(idb is not None)
        if ok:
            try:
                eng.comm_init(world, rank, bytes(idt.cpu().tolist()))     # collective inside RCCL: every rank is here
            except Exception as exc:          # noqa: BLE001
                sys.stderr.write("bench.py: ncclCommInitRank failed on rank %d (%s)\n" % (rank, exc))
                ok = False
            ok = agree(ok)
        if ok:
            t1 = torch.tensor([rank + 1.0, 0.5], dtype=torch.float64, device=dev)
            t2 = t1.clone()
            same = True
            try:
                eng.allreduce_sum_async(t1.data_ptr(), 2)
            except Exception as exc:          # noqa: BLE001
                sys.stderr.write("bench.py: ncclAllReduce failed on rank %d (%s)\n" % (rank, exc))
                same = False
            dist.all_reduce(t2, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            ok = agree(same and torch.equal(t1, t2))
        native = ok
        if not native and rank == 0:
            sys.stderr.write("bench.py: the engine's RCCL reduction is not usable on every rank; using torch.distributed.all_reduce\n")
    RING = 4
    red = torch.zeros((RING, 2), dtype=torch.float64, device=dev)
    pending = []
    state = {"i": 0}

    def step():
        i = state["i"] % RING
        state["i"] += 1
        if len(pending) >= RING - 1:
            pending.pop(0).wait()                      # the slot about to be rewritten has been reduced
        eng.update_edge_rates(wl.edge_rates_csr)       # new rates: K1 (+ stream + tables) runs again
        eng.ll_async(sum_device_ptr=red[i].data_ptr()) # {hi, lo} of this rank's block, left on the device
        if use_dist and native:
            eng.allreduce_sum_async(red[i].data_ptr(), 2)
        elif use_dist:
            if args.sync_allreduce:
                dist.all_reduce(red[i], op=dist.ReduceOp.SUM)
            else:
                pending.append(dist.all_reduce(red[i], op=dist.ReduceOp.SUM, async_op=True))

    def fence():
        while pending:
            pending.pop(0).wait()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.info(E.INFO_LL_KERNEL_NS_SUM)                  # reset the event sums
    eng.info(E.INFO_LL_KERNEL_COUNT)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kern_ns_sum, kern_count = eng.info(E.INFO_LL_KERNEL_NS_SUM), eng.info(E.INFO_LL_KERNEL_COUNT)
    assert kern_count == args.steps, (kern_count, args.steps)
    total = float(red[(state["i"] - 1) % RING].sum().item())
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # per-site values of the first sites of rank 0's block, for the cross-check against the CPU checker below
    ncheck = min(S, 4096)
    gpu_ll_head = None
    if rank == 0:
        site_ll = torch.empty(S, dtype=torch.float64, device=dev)
        eng.ll(out_device_ptr=site_ll.data_ptr(), want_sum=False)
        gpu_ll_head = site_ll[:ncheck].cpu().numpy()
        del site_ll

    # edge-gradient leg: arbplf-deriv on the head of this rank's block (site-summed gradient, reduced across ranks)
    deriv_out = None
    if args.deriv_steps > 0:
        Sd = min(S, args.deriv_sites or 2_000_000)
        if Sd != S:
            cd = torch.empty((wl.N, Sd), dtype=torch.uint8, device=dev)
            for off in range(0, Sd, chunk):
                n = min(chunk, Sd - off)
                cd[:, off:off + n] = wl.simulate(n, site0=s0 + off, device=dev)
            torch.cuda.synchronize()
            eng.set_patterns_codes(cd.data_ptr(), wl.defs, S=Sd, where=E.DEVICE)
            del cd
        grad = None

        def dstep():
            eng.update_edge_rates(wl.edge_rates_csr)
            _, sums = eng.deriv(per_site=False)
            return shard.allreduce_dd(sums, dev)          # one all-reduce of the 2E {hi, lo} words when distributed

        dstep()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.deriv_steps):
            grad = dstep()
        fence()
        dd_t = time.perf_counter() - t1
        if use_dist:
            tmax = torch.tensor([dd_t], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dd_t = float(tmax.item())
        alg_d = wl.algorithmic()
        dkey = "deriv4" if wl.k == 4 else ("deriv_vec" if wl.k <= 20 else "deriv_mfma")
        dtr = load_traffic(os.path.join(ROOT, "profiles", "traffic_cfg%d_deriv.json" % args.config), dkey, Sd)
        deriv_out = dict(value=world * Sd * args.deriv_steps / dd_t, unit="sites/s", sites_per_gpu=Sd, steps=args.deriv_steps,
                         ms_per_step=dd_t / args.deriv_steps * 1e3, grad_max=float(np.abs(np.asarray(grad)).max()),
                         hbm_model_bytes_per_site=alg_d["A_deriv"],
                         hbm_model_equiv_frac=alg_d["A_deriv"] * Sd * args.deriv_steps / dd_t / HBM_PEAK,
                         hbm_measured_bytes_per_step=dtr["hbm_bytes_per_launch"] if dtr else None,
                         hbm_measured_frac=(dtr["hbm_bytes_per_launch"] / (dd_t / args.deriv_steps) / HBM_PEAK) if dtr else None,
                         note="whole arbplf-deriv step (K1, tables, down pass, up pass, weighted site sums%s). hbm_measured_frac = PMC "
                              "bytes of the down + up kernels (profiles/traffic_cfg%d_deriv.json, used only when it was measured on "
                              "this tree's kernel sources) / step time / 8 TB/s: the utilisation figure.  hbm_model_equiv_frac "
                              "prices SURVEY.md 8d's A_deriv = 6(I-1)Ck*8 + N (stored edge vectors) at the step time; the kernels "
                              "recompute edge vectors instead of storing them, so it is an equivalent rate, not a utilisation"
                              % (", all-reduce of 2E doubles" if use_dist else "", args.config))

    if rank == 0:
        alg = wl.algorithmic()
        kernel_kind = eng.info(E.INFO_LL_KERNEL)
        kern_s = kern_ns_sum / kern_count * 1e-9
        value = S_total * args.steps / dt
        hbm_equiv = alg["A_ll"] * S / kern_s
        flops = alg["W_ll"] * S / kern_s
        kname = {1: "k_ll_fused4_asm", 2: "k_ll_generic", 3: "k_ll_mfma", 4: "k_ll_vec"}.get(kernel_kind, "?")
        if kernel_kind == 1:
            kname = {1: "k_ll_fused4_asm", 3: "k_ll_fused4", 5: "k_ll_fused4_asm_pt", 6: "k_ll_fused4_v4"}.get(eng.info(E.INFO_LL_VARIANT), kname)
        # HBM bytes per launch from the PMC passes of the guide's recipe (profiles/traffic_cfgN.json, written by
        # tools/pmc_traffic.py from rocprofv3 --pmc runs of this kernel).  Only used when the file was measured at this
        # site count on the kernel sources of this very tree (git blob hashes); anything else is null, not a stale number.
        tkey = {1: "k_ll_fused4", 2: "k_ll_generic", 3: "k_ll_mfma", 4: "k_ll_vec"}.get(kernel_kind, "?")
        tj = load_traffic(os.path.join(ROOT, "profiles", "traffic_cfg%d.json" % args.config), tkey, S)
        traffic = tj.get("hbm_bytes_per_launch") if tj and str(tj.get("kernel", "")).startswith(kname) else None
        hbm_model = dict(achieved=hbm_equiv / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", equiv_ratio=hbm_equiv / HBM_PEAK,
                         note="A_ll=%d B/site of the HBM-resident-partials design (SURVEY.md 8d) priced at the kernel time; "
                              "an equivalent rate (the partial vectors never move through HBM), not a utilisation" % alg["A_ll"])
        # which roof binds the kernel that ran: the fused, vector and matrix-core kernels keep the
        # partial vectors out of HBM (compulsory traffic N+8 B/site), so fp64 arithmetic binds them;
        # the generic vector kernel streams its stack slots through HBM.
        fp64_bound = kernel_kind in (1, 3, 4) or alg["W_ll"] / FP64_PEAK > alg["A_ll"] / HBM_PEAK
        if fp64_bound:
            # executed work, counted by the engine from the program the kernel ran (PLK_INFO_LL_EXEC_FLOPS): 2k^2 - k per
            # matrix-vector product (k padded to 16 rows on the matrix cores), k per elementwise multiply; table look-ups
            # (leaf rows, two-leaf subtrees), moves and rescaling count nothing
            exec_flops = eng.info(E.INFO_LL_EXEC_FLOPS)
            npairs = eng.info(E.INFO_PAIR_TABLES)
            r02_flops = wl.prepare()["C"] * ((wl.E - wl.T) * (2 * wl.k * wl.k - wl.k) + wl.T * wl.k)
            roofline = dict(bound="fp64", achieved=exec_flops * S / kern_s / 1e12, peak=FP64_PEAK / 1e12, unit="TFLOP/s",
                            frac=exec_flops * S / kern_s / FP64_PEAK, traffic=traffic, kernel=kname, kernel_ms=kern_s * 1e3,
                            flops_per_site=exec_flops,
                            pair_tables=npairs,
                            note="EXECUTED fp64 work / kernel time / fp64 peak.  Executed flops per site are counted by the engine from "
                                 "the traversal program that ran: 2k^2 - k per matrix-vector product (k padded to 16 rows on the matrix "
                                 "cores) + k per elementwise multiply; leaf-edge products are table rows and, in the k = 4 pair-table "
                                 "kernels, every two-leaf subtree with the edge above it is one table row: those products are not "
                                 "executed and not counted, nor are stack and rescaling moves.  Peak 78.6 "
                                 "TFLOP/s = 256 CUs x 4 SIMDs x 16 fp64 FMA lanes x 2 flop x 2.4 GHz, AMD's public MI355X figure for fp64 "
                                 "vector and matrix alike (MI355X_MICROARCH.md lists no fp64 peak).  Kernel time from HIP events "
                                 "over the %d timed launches on the kernel's stream" % kern_count,
                            round2_work=dict(flops_per_site=r02_flops, ratio=r02_flops * S / kern_s / FP64_PEAK,
                                             note="the work the round-2 kernels executed for this workload (every internal edge a "
                                                  "product, every leaf a multiply) priced at this kernel's time: like-for-like speed "
                                                  "against round 2's executed fraction, not a utilisation of this kernel"),
                            algorithmic=dict(flops_per_site=alg["W_ll"], achieved=flops / 1e12, ratio=flops / FP64_PEAK,
                                             note="SURVEY.md 8d's W_ll (a full product on every edge, leaf edges included) priced at "
                                                  "the kernel time: an equivalent rate, may exceed the peak, not a utilisation"),
                            hbm_model_equiv=hbm_model)
        else:
            roofline = dict(bound="hbm", achieved=hbm_equiv / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                            frac=hbm_equiv / HBM_PEAK, traffic=traffic, kernel=kname, kernel_ms=kern_s * 1e3,
                            note="A_ll=%d B/site" % alg["A_ll"])
        out = {
            "metric": "sites/sec (arbplf-ll)", "value": value, "unit": "sites/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config %d: %s, %d sites in all, site-sharded in contiguous blocks" % (args.config, wl.name, S_total),
                       "total_sites": S_total, "sites_per_gpu": S, "states": wl.k, "categories": wl.prepare()["C"], "taxa": wl.T,
                       "parallelism": "site-shard x%d" % world},
            "ll_sum": total,
            "allreduce": ("rccl, queued by the engine on its stream" if native else "torch.distributed (nccl)") if use_dist else None,
            "step_minus_kernel_ms": dt / args.steps * 1e3 - kern_s * 1e3,
            "roofline": roofline,
        }
        if deriv_out is not None:
            out["deriv"] = deriv_out
        if not args.no_cpu_baseline:
            if world == 1:
                cb, cpu_ll = cpu_baseline(wl, min(2 * S, 20_000_000), args.cpu_seconds)
                out["cpu_baseline"] = cb
                out["speedup_vs_cpu_baseline"] = value / cb["value"]
            else:
                # N > 1: no timed CPU leg, only the checker values for the cross-check (long double build, 4096 sites)
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from oracle import arbplf_oracle as O
                from helpers import oracle_site_ll
                cpu_ll = oracle_site_ll(O, wl, wl.simulate(ncheck))
            # the bench checks its own result: GPU per-site ll of the first sites against the CPU checker
            n = min(len(cpu_ll), len(gpu_ll_head))
            err = float(np.max(np.abs(gpu_ll_head[:n] - cpu_ll[:n]) / np.maximum(1.0, np.abs(cpu_ll[:n]))))
            out["check"] = dict(sites=n, max_rel_err=err, tol=1e-12,
                                against="oracle (long double build) on the first %d sites of the alignment" % n)
            if not err <= 1e-12:
                restore_stdout()
                print(json.dumps(out), flush=True)
                raise SystemExit("bench.py: GPU per-site ll differs from the CPU checker (max rel err %g)" % err)
        restore_stdout()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
