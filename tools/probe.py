"""Measurement probes, one sub-command each (GPU box, repo root):  python tools/probe.py <probe> [args]
    h             [log_m] [bit_fraction]   R1CS evaluation + H pipeline alone, both H forms
    h-sha         [depth]   the same on the SHA-256 Merkle circuit
    witness-jobs  each witness multiexp of a 2^22 config-4 proof alone, all four together, the h multiexp
    g2-ratio      [log_n]   plain Pippenger over uniform scalars in G1 and G2: the cost ratio
    ntt           [log_m]   the transform passes alone (for rocprofv3 --kernel-trace --stats)
    prove-timing  [log_m]   zg_prove (pinned host witness) against zg_prove_dev, with the event-timed phases
    interference  [none|witness]   who slows the transforms inside a proof
    numa          alloc_node run_node   the host packing pass against NUMA placement
    small-chains  [eddsa|<log_m>]   each multiexp of a small proof alone: the latency chains a 1 ms proof consists of
"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def probe_h(argv):
    import za_amd
    from za_amd import synth
    lg = int(argv[0]) if len(argv) > 0 else 22
    bf = float(argv[1]) if len(argv) > 1 else 0.97
    ctx = za_amd.Context(0)
    sr = synth.SynthR1CS(lg, 2, bf, 0x5A)
    R = sr.load(ctx)
    d_w = ctx.upload(sr.witness)
    d_h = ctx.alloc(32 << lg)
    for name, fn in (("coefficient form, 7 transforms", ctx.prove_h_dev), ("evaluation form, 6 transforms", ctx.prove_h_eval_dev)) * 2:
        for _ in range(3):
            fn(R, d_w, d_h)
        t = time.perf_counter()
        for _ in range(10):
            fn(R, d_w, d_h)
        print("R1CS evaluation + H pipeline (%s): %.3f ms" % (name, 1e3 * (time.perf_counter() - t) / 10))
    ctx.profile_enable(2)
    ctx.profile_reset()
    for _ in range(4):
        ctx.prove_h_eval_dev(R, d_w, d_h)
    p = ctx.profile_get()
    print("per call: r1cs_eval %.3f ms, transforms %.3f ms in %d launches" % (p["r1cs_eval"][0] / 4, p["ntt_pass"][0] / 4, p["ntt_pass"][1] // 4))

def probe_h_sha(argv):
    import numpy as np
    import za_amd
    from za_amd import workloads
    depth = int(argv[0]) if len(argv) > 0 else 90
    ctx = za_amd.Context(0)
    ni, na, nc, rp, col, cf, wit, info = workloads.sha256_merkle(depth)
    R = ctx.r1cs_load(ni, na, nc, rp, col, cf)
    d_w = ctx.upload(np.frombuffer(wit, dtype=np.uint8))
    d_h = ctx.alloc(32 << R.log_m)
    lens = [np.diff(rp[k].astype(np.int64)) for k in range(3)]
    print("rows", nc, "log_m", R.log_m, "nnz", info["nnz"], "longest rows", [int(l.max()) for l in lens],
          "rows > 32 terms", [int((l > 32).sum()) for l in lens])
    ctx.profile_enable(2)
    for _ in range(3):
        ctx.prove_h_dev(R, d_w, d_h)
    ctx.profile_reset()
    t = time.perf_counter()
    for _ in range(10):
        ctx.prove_h_dev(R, d_w, d_h)
    print("H pipeline: %.3f ms" % (1e3 * (time.perf_counter() - t) / 10))
    p = ctx.profile_get()
    print({k: (round(v[0] / 10, 3), v[1] // 10) for k, v in p.items() if v[1]})

def probe_witness_jobs(argv):
    import za_amd
    from za_amd import synth
    ctx = za_amd.Context(0)
    sr = synth.SynthR1CS(22, 2, 0.97, 0x5A)
    R = sr.load(ctx)
    P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
    d_w = ctx.upload(sr.witness)
    P.prepare(R, d_w)
    d_h = ctx.alloc(32 << 22)
    ctx.prove_h_dev(R, d_w, d_h)
    for name, mask in (("l", 2), ("a", 4), ("b_g1", 8), ("b_g2", 16), ("all four", 0x1e), ("h", 1)):
        ts = []
        for i in range(6):
            t0 = time.perf_counter()
            ctx.prove_partials_dev(P, R, d_w, d_h, mask)
            ts.append(1e3 * (time.perf_counter() - t0))
        print("%-8s %.3f ms (min %.3f)" % (name, sum(ts[2:]) / 4, min(ts)))

def probe_g2_ratio(argv):
    import za_amd
    from za_amd import synth
    n = 1 << (int(argv[0]) if len(argv) > 0 else 21)
    ctx = za_amd.Context(0)
    d_b1 = synth.synth_bases_g1(ctx, n, 4242)
    d_b2 = synth.synth_bases_g2(ctx, n, 4243)
    d_s = synth.synth_scalars(ctx, n, 99)
    ctx.profile_enable(2)
    for name, fn, d_b in (("g1", ctx.msm_g1_dev, d_b1), ("g2", ctx.msm_g2_dev, d_b2)):
        for i in range(4):
            ctx.profile_reset()
            t0 = time.perf_counter()
            fn(d_b, d_s, n)
            dt = 1e3 * (time.perf_counter() - t0)
            print("%s %d: %.3f ms" % (name, i, dt), ctx.profile_get() if i == 3 else "")

def probe_ntt(argv):
    import torch, za_amd
    from za_amd import synth
    lg = int(argv[0]) if len(argv) > 0 else 22
    ctx = za_amd.Context(0)
    d = synth.synth_scalars(ctx, 1 << lg, 5)
    for _ in range(3):
        ctx.ntt_dev(d, lg, False, False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        ctx.ntt_dev(d, lg, False, False)
    torch.cuda.synchronize()
    print("ms per zg_ntt_fr_dev (3 passes + bitrev + 2 conversions):", 1e3 * (time.perf_counter() - t) / 20)
    ctx.free(d); ctx.close()

def probe_prove_timing(argv):
    import numpy as np
    import za_amd
    from za_amd import synth
    lg = int(argv[0]) if len(argv) > 0 else 22
    ctx = za_amd.Context(0)
    sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
    R = sr.load(ctx)
    P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
    pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
    d_w = ctx.upload(sr.witness)
    P.prepare(R, d_w)
    for mode in ("dev", "host", "dev", "host"):
        tm = []
        for i in range(13):
            t0 = time.perf_counter()
            if mode == "dev":
                ctx.prove_dev(P, R, d_w, 5, 7)
            else:
                ctx.prove(P, R, pin, 5, 7)
            tm.append([1e3 * (time.perf_counter() - t0)] + list(ctx.last_timings().values()))
        tm = np.array(tm[3:])
        print(mode, "wall %.3f" % tm[:, 0].mean(), "timings", np.round(tm[:, 1:].mean(axis=0), 3))

def probe_interference(argv):
    import za_amd
    from za_amd import synth
    mode = argv[0] if len(argv) > 0 else "none"
    ctxA, ctxB = za_amd.Context(0), za_amd.Context(0)
    sr = synth.SynthR1CS(22, 2, 0.97, 0x5A)
    RA, RB = sr.load(ctxA), sr.load(ctxB)
    P = ctxB.setup(RB, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
    dwA, dwB = ctxA.upload(sr.witness), ctxB.upload(sr.witness)
    P.prepare(RB, dwB)
    dhA = ctxA.alloc(32 << 22)
    dhB = ctxB.alloc(32 << 22)
    ctxB.prove_h_dev(RB, dwB, dhB)
    stop = False
    count = [0]
    def worker():
        while not stop:
            ctxB.prove_partials_dev(P, RB, dwB, dhB, 0x1e)
            count[0] += 1
    for _ in range(3):
        ctxA.prove_h_dev(RA, dwA, dhA)
    th = None
    if mode == "witness":
        th = threading.Thread(target=worker); th.start(); time.sleep(0.2)
    t = time.perf_counter()
    N = 40
    for _ in range(N):
        ctxA.prove_h_dev(RA, dwA, dhA)
    dt = 1e3 * (time.perf_counter() - t) / N
    stop = True
    if th: th.join()
    print("mode %s: H pipeline %.3f ms per run; witness phases run meanwhile: %d" % (mode, dt, count[0]))

def probe_numa(argv):
    nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
    def cpus(n):
        out = []
        for part in open(n + "/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
        return out
    print("nodes", [(os.path.basename(n), len(cpus(n))) for n in nodes])
    alloc_node, run_node = int(argv[0]), int(argv[1])
    os.sched_setaffinity(0, cpus(nodes[alloc_node]))
    import numpy as np
    import za_amd
    from za_amd import synth
    ctx = za_amd.Context(0)
    sr = synth.SynthR1CS(22, 2, 0.97, 0x5A)
    R = sr.load(ctx)
    P = synth.synth_params(ctx, (1 << 22) - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 7)
    pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
    os.sched_setaffinity(0, cpus(nodes[run_node]))
    ts = []
    for i in range(12):
        t0 = time.perf_counter(); ctx.prove(P, R, pin, 5, 7); t = time.perf_counter() - t0
        ts.append("%.2f/%dMB" % (1e3 * t, ctx.last_timings()["upload_bytes"] / 1e6))
    print("proofs", " ".join(ts))
    print("alloc node", alloc_node, "run node", run_node, "last proof %.3f ms" % (1e3 * t), ctx.last_timings()["upload_bytes"])

def probe_small_chains(argv):
    import numpy as np
    import za_amd
    from za_amd import synth, workloads
    what = argv[0] if argv else "eddsa"
    ctx = za_amd.Context(0)
    if what == "eddsa":
        n_in, n_aux, cons, wit = workloads.eddsa_poseidon_verifier()
        rp, cl, cf = workloads.csr_arrays(cons)
        R = ctx.r1cs_load(n_in, n_aux, len(cons), rp, cl, cf)
        wb = np.frombuffer(workloads.witness_bytes(wit), dtype=np.uint8)
    else:
        sr = synth.SynthR1CS(int(what), 2, 0.97, 0x5A)
        R = sr.load(ctx)
        wb = np.array(sr.witness, copy=True)
    P = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
    d_w = ctx.upload(wb)
    P.prepare(R, d_w)
    d_h = ctx.alloc(32 << R.log_m)
    for _ in range(3):
        ctx.prove_dev(P, R, d_w, 5, 7)
    for name, fn in (("R1CS + H (evaluation form)", lambda: ctx.prove_h_eval_dev(R, d_w, d_h)),):
        ts = []
        for i in range(24):
            t0 = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t0))
        print("%-28s median %.3f min %.3f ms" % (name, sorted(ts)[12], min(ts)))
    for name, mask in (("h", 1 | 0x20), ("l", 2), ("a", 4), ("b_g1", 8), ("b_g2", 16), ("b_g2 + h", 17 | 0x20), ("b_g2 + h + l", 19 | 0x20),
                       ("b_g2 + h + l + a", 23 | 0x20), ("l + a + b_g1", 14), ("all five", 0x1f | 0x20)):
        ts = []
        for i in range(24):
            t0 = time.perf_counter()
            ctx.prove_partials_dev(P, R, d_w, d_h, mask)
            ts.append(1e3 * (time.perf_counter() - t0))
        print("%-28s median %.3f min %.3f ms" % (name, sorted(ts)[12], min(ts)))
    ts = []
    for i in range(24):
        t0 = time.perf_counter(); ctx.prove_dev(P, R, d_w, 5, 7); ts.append(1e3 * (time.perf_counter() - t0))
    print("%-28s median %.3f min %.3f ms" % ("zg_prove_dev", sorted(ts)[12], min(ts)))

PROBES = {"small-chains": probe_small_chains, "h": probe_h, "h-sha": probe_h_sha, "witness-jobs": probe_witness_jobs, "g2-ratio": probe_g2_ratio, "ntt": probe_ntt, "prove-timing": probe_prove_timing, "interference": probe_interference, "numa": probe_numa}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in PROBES:
        print(__doc__)
        sys.exit(2)
    PROBES[sys.argv[1]](sys.argv[2:])
