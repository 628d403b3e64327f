#!/usr/bin/env python3
"""Benchmark of the za_amd Groth16/BN254 prover hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--log-m 22]

A "step" is one create_proof -- zg_prove: the witness comes from (pinned) HOST memory and its hand-over is part of the
step, as SURVEY 8(d) defines the metric (a mostly-0/1 witness is packed by a few host threads before it crosses PCIe) --
of the synthetic 2^log_m R1CS of config 4 (SHA256-Merkle-like witness: 97 % bits, SplitMix64 seed 0x5A) with the R1CS and
the proving key resident in HBM; r, s are fixed.  N > 1 is launched by torch.distributed.run, one rank per GPU: the MSMs
shard by base chunk, the NTT stays on rank 0 (za_amd/dist.py).  Rank 0 prints ONE JSON line.
`roofline` is for the dominant kernel, the G1 bucket accumulation, as launched by the h multiexp (every point of such a
launch reaches the kernel), from HIP events inside the timed region; `roofline.by_job_class` also carries the witness
jobs' launches and the blend of both (the figure a rocprofv3 --stats summary shows: one kernel name); `traffic` is filled
only for the workload the PMC passes in profiles/ were taken on.  `roofline_ntt` is the H pipeline against the same
roofline.  `cpu_baseline` times the oracle (a port of bellman's CPU algorithm) on the same workload on this box's host
cores.  Secondary keys: prove_dev_ms (witness already in HBM), prove_packed (the caller hands over the packed form),
g1_msm (the metric's second half: 2^26 resident G1 bases, scalars uploaded), dense_witness (same size, no 0/1 shortcuts),
sha256_merkle, shard_model (the per-rank phases of an 8-way sharded proof measured on this one GPU), concurrent_provers.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the five MSM streams + the main stream each want a hardware queue (ROCm default: 4 per process);
# must be set before the first HIP call (torch import)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_ISSUE_PEAK = 467.5e9 * 64   # lane v_mad_u64_u32 / s, tools/ubench_valu.hip on this chip
MADS_PER_MIXED_ADD = 8 * 162 + 2 * 126 - 81   # y3's two products share one reduction (fd.h mac2)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--provers", type=int, default=3,
                    help="extra line: throughput with this many concurrent provers on the GPU (1 = skip); the headline "
                         "value is always one proof at a time")
    ap.add_argument("--log-m", type=int, default=22, help="domain size: n_constraints + n_inputs = 2^log_m")
    ap.add_argument("--workload", default="synthetic", choices=["synthetic", "poseidon", "sha256", "eddsa"],
                    help="synthetic = config-4-shaped random R1CS (the headline); poseidon = circomlib Poseidon(t=6) hash "
                         "chain padded to 2^log_m (BASELINE config 3: use --log-m 20), generated in Python (~15 s at 2^20); "
                         "sha256 = bit-level SHA-256 Merkle inclusion of --depth levels (BASELINE config 4 on a real "
                         "circuit: depth 32 is 1.48 M constraints in a 2^21 domain, depth 90 fills 2^22); --log-m is "
                         "then derived from the circuit; eddsa = one EdDSA-Poseidon signature verification on Baby Jubjub "
                         "(BASELINE config 2's statement, 7.5 k constraints, the reference's signature as witness)")
    ap.add_argument("--depth", type=int, default=32, help="--workload sha256: levels of the Merkle path")
    ap.add_argument("--bit-fraction", type=float, default=0.97)
    ap.add_argument("--seed", type=lambda x: int(x, 0), default=0x5A)
    ap.add_argument("--cpu-log-m", type=int, default=22, help="size of the bounded CPU-baseline sample (2^22 = the GPU workload itself, ~10 s of CPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default: min(host cores, 64))")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo stages through host memory (debugging only)")
    ap.add_argument("--synthetic-key", action="store_true", help="skip the set-up; use bases with known discrete logs")
    ap.add_argument("--no-msm", action="store_true", help="skip the standalone G1 MSM line")
    ap.add_argument("--msm-log-n", type=int, default=26, help="standalone G1 MSM: 2^n bases in total (sharded over the ranks)")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-witness (bit fraction 0) line")
    ap.add_argument("--no-sha", action="store_true", help="skip the SHA-256 Merkle (depth 32, real circuit) line")
    ap.add_argument("--no-outer", action="store_true",
                    help="skip key_load / outer_prove / one_shot (the reference's per-proof flow: read the key checked, prove once)")
    ap.add_argument("--shard-model", type=int, default=8,
                    help="N = 1 only: also time the phases one rank of a proof sharded W ways would run (0 = skip)")
    return ap.parse_args()


def cpu_baseline(ctx, args):
    """Oracle prove (bellman-algorithm CPU port) on a 2^cpu_log_m instance of the same workload."""
    from oracle import coracle as co
    from za_amd import synth
    lg = min(args.cpu_log_m, args.log_m)
    if args.workload == "sha256":
        from za_amd import workloads

        class _W:
            pass
        sr = _W()
        (sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff, sr.witness,
         _info) = workloads.sha256_merkle(args.depth)
        sr.close = lambda: None
        r1 = ctx.r1cs_load(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
        lg = r1.log_m
    else:
        sr = synth.SynthR1CS(lg, 2, args.bit_fraction, args.seed)
        r1 = sr.load(ctx)
    m = 1 << lg
    n_a = sr.n_inputs + r1.a_aux_total
    n_b = r1.b_input_total + r1.b_aux_total
    if args.synthetic_key:
        P = synth.synth_params(ctx, m - 1, sr.n_aux, n_a, n_b, 1000)
    else:
        P = ctx.setup(r1, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
    # bring the key to the host for the CPU run
    import ctypes as C
    vk5 = np.zeros(448, dtype=np.uint8)
    ctx._check(ctx.lib.zg_params_download(P.h, C.c_int(-1), vk5.ctypes.data_as(C.c_void_p), C.c_size_t(448)))
    arrs = []
    for which, elem in ((0, 64), (1, 64), (2, 64), (3, 64), (4, 128)):
        tot = P.query_total(which)
        a = np.zeros(max(tot, 1) * elem, dtype=np.uint8)
        ctx._check(ctx.lib.zg_params_download(P.h, C.c_int(which), a.ctypes.data_as(C.c_void_p), C.c_size_t(tot * elem)))
        arrs.append(a[: tot * elem])
    mp = co.MontParams(vk5[0:64].tobytes(), vk5[64:128].tobytes(), vk5[128:256].tobytes(), vk5[256:320].tobytes(),
                       vk5[320:448].tobytes(), *arrs)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    # bellman's Worker uses one thread per host core -- and its parallel_fft does 2n multiplications PER THREAD before
    # the sub-transforms, so it stops scaling early: on this pool's 256-thread hosts the port needs 45 s at 256 threads
    # (H 42.5 s) against 9 s at 64.  The baseline is run at the thread count that favours the CPU (capped at 64).
    cores = os.cpu_count() or 1
    threads = min(cores, args.cpu_threads) if args.cpu_threads else min(cores, 64)
    t0 = time.time()
    proof_cpu, tm = co.prove(mp, csr, sr.witness, 5, 7, threads=threads, timings=True)
    dt = time.time() - t0
    proof_gpu = ctx.prove(P, r1, sr.witness, 5, 7)
    ok = proof_cpu == proof_gpu
    P.close(); r1.close()
    n = sr.n_constraints
    sr.close()
    return {"value": n / dt, "unit": "constraints/s", "cores": threads, "host_cores": cores, "kind": "port",
            "sample": "oracle/c create_proof (bellman-algorithm port: c = ceil(ln n) windows, the 8 multiexps' windows "
                      "in one %d-thread pool, radix-2 FFT) on the same workload at 2^%d constraints: "
                      "%.2f s (r1cs %.2f, H %.2f, MSM %.2f); proof bytes equal to the GPU's: %s"
                      % (threads, lg, dt, tm[0], tm[1], tm[2], ok)}


def outer_measurements(ctx, args, sr, r1, P, w_host):
    """The reference's own flow, which pays the key on EVERY proof (helper.rs:91-147: read_pk -> Parameters::read(pk,
    checked = true) -> witness -> generate_verified_proof): (a) key_load: Parameters::read of this workload's key, unchecked
    and checked (validated on the device, csrc/keyload.hip); (b) one_shot: zg_prove with ZG_OPT_ONE_SHOT (no per-key
    tables, bellman's seven transforms) on the resident key; (c) outer_prove: ONE libza2c.so prove() -- through the
    za-gpu driver as a child process -- on the workload written as a proving.key container + .binwitness."""
    import re
    import shutil
    import subprocess
    import tempfile
    import za_amd
    from za_amd import build as zb
    out = {}
    t0 = time.perf_counter()
    pbytes = P.write()
    t_write = time.perf_counter() - t0
    key_load = {"bytes": len(pbytes)}
    for name, checked in (("unchecked_s", False), ("checked_s", True), ("unchecked_again_s", False), ("checked_again_s", True)):
        t0 = time.perf_counter()
        P2, _ = ctx.params_load(pbytes, checked=checked)
        key_load[name] = time.perf_counter() - t0
        key_load[name.replace("_s", "_device_ms")] = ctx.last_load_timings()["queries_device_ms"]
        P2.close()
    key_load["points"] = {n: P.query_total(k) for k, n in enumerate(("h", "l", "a", "b_g1", "b_g2"))}
    key_load["note"] = ("zg_params_load from pageable host memory: the five queries are uploaded as they stand in the file and "
                        "decoded / validated by one kernel per query (on the curve; b_g2: order-r membership by the psi "
                        "endomorphism on lane pairs); *_device_ms = upload + kernels + verdict read-back; "
                        "Parameters::write of the key took %.2f s" % t_write)
    out["key_load"] = key_load
    # one-shot proofs on the resident key (before any table is built)
    ctx.set_option(za_amd.OPT_ONE_SHOT, 1)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        proof1 = ctx.prove(P, r1, w_host, 0x1234567, 0x7654321)
        ts.append(1e3 * (time.perf_counter() - t0))
    ctx.set_option(za_amd.OPT_ONE_SHOT, 0)
    out["one_shot"] = {"first_ms": ts[0], "ms": float(np.median(ts[1:])),
                       "proof_sha": __import__("hashlib").sha256(bytes(proof1)).hexdigest()[:16],
                       "note": "zg_prove with ZG_OPT_ONE_SHOT: no evaluation-form bases, no tables, bellman's seven transforms; "
                               "first_ms includes the workspace allocations of a fresh context"}
    # one prove() of the outer ABI on files
    tmp = tempfile.mkdtemp(prefix="za_bench_", dir=os.environ.get("TMPDIR") or "/tmp")
    try:
        pk_path, bw_path, proof_path = os.path.join(tmp, "proving.key"), os.path.join(tmp, "w.binwitness"), os.path.join(tmp, "proof.json")
        t0 = time.perf_counter()
        sr.write_proving_key(pk_path, pbytes)
        sr.write_binwitness(bw_path)
        t_files = time.perf_counter() - t0
        del pbytes
        runs = []
        for _ in range(2):      # the second run has the files in the page cache for certain
            t0 = time.perf_counter()
            r = subprocess.run([zb.DRIVER, "prove", "--pk", pk_path, "--witness", bw_path, "--proof", proof_path],
                               capture_output=True, text=True)
            wall = time.perf_counter() - t0
            tm = {k.strip(): float(v) for k, v in re.findall(r"^\s*([A-Za-z()+ ]+?) time: ([0-9.]+)s", r.stderr, flags=re.M)}
            sub = re.search(r"walk ([0-9.]+)s, queries checked on the GPU ([0-9.]+)s, constraints ([0-9.]+)s, R1CS to the GPU ([0-9.]+)s", r.stderr)
            runs.append((r.returncode, wall, tm, sub, r.stderr[-400:]))
        rc, wall, tm, sub, tail = runs[-1]
        if rc != 0:
            out["outer_prove"] = {"error": tail}
        else:
            pj = open(proof_path).read()
            proof_o, pub_o = za_amd.proof_from_json(pj)
            vk = P.write_vk()
            out["outer_prove"] = {
                "pk_bytes": os.path.getsize(pk_path), "binwitness_bytes": os.path.getsize(bw_path),
                "gpu_context_s": tm.get("GPU context"), "pk_read_s": tm.get("Proving key read"),
                "pk_read_parts_s": {"vk_and_walk": float(sub.group(1)), "queries_checked_on_gpu": float(sub.group(2)),
                                    "constraints_to_csr": float(sub.group(3)), "r1cs_to_gpu": float(sub.group(4))} if sub else None,
                "witness_s": tm.get("Witness read"), "check_s": tm.get("Constraint check"),
                "prove_one_shot_ms": 1e3 * tm["Proof generation"] if "Proof generation" in tm else None,
                "verify_ms": 1e3 * tm["Proof verification"] if "Proof verification" in tm else None,
                "total_s": tm.get("prove()"), "process_wall_s": wall, "first_run_process_wall_s": runs[0][1],
                "proof_verifies": bool(za_amd.verify(vk, proof_o, pub_o)),
                "note": "ONE prove() of libza2c.so (za-gpu prove as a child process: its own GPU context, the key read with "
                        "checked = true, constraint check, one-shot proof with fresh r, s, self-verification on the host, "
                        "proof.json), files in the page cache; total_s is the call, process_wall_s adds process start and "
                        "HIP initialisation; writing the two files took %.1f s" % t_files}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


VALU_PEAK_NOTE = ("tools/ubench_valu.hip on this chip (profiles/r02_ubench_fmul.txt): 467.5 G wave-level v_mad_u64_u32 / s "
                  "measured; the nominal figure 1024 SIMDs x 2.4 GHz / 4 cycles = 614 G is given beside it "
                  "(MI355X_MICROARCH.md has no integer issue rates)")
VALU_ISSUE_NOMINAL = 1024 * 2.4e9 / 4 * 64   # lane multiply-adds / s at the nominal clock, quarter rate


def _acc_class(prof, key, hbm=True):
    """hbm = False (the witness queries' launches): no HBM fraction -- SURVEY 8(d)'s 96 B are per point that is multiplied,
    and a launch of a witness job is credited every POSITION of its query although a zero scalar never reaches the kernel
    (and, for a folded key, the 2 m positions of the C rows, which are not survey points at all)."""
    ms, n, nbytes, units = prof.get(key, (0.0, 0, 0, 0))
    if not n:
        return None
    achieved = nbytes / (ms * 1e-3) / 1e9
    mads = units * MADS_PER_MIXED_ADD
    out = {"launches": n, "avg_launch_ms": ms / n, "mixed_adds_per_launch": units / n,
           "alu_frac_of_measured_issue_rate": mads / (ms * 1e-3) / VALU_ISSUE_PEAK,
           "alu_frac_of_nominal_issue_rate": mads / (ms * 1e-3) / VALU_ISSUE_NOMINAL}
    if hbm:
        out.update({"algorithmic_bytes_per_launch": nbytes / n, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS})
    return out


def acc_roofline(prof, pmc=None):
    """roofline object for k_accumulate<Fq> from the level-1 HIP-event profile of a timed region.  The kernel serves two
    classes of job: the h multiexp (`h_dense`: every point of the launch reaches the kernel -- the honest figure, and the one
    the top-level fields carry) and the witness queries' G1 jobs (`witness`: credited 96 B for every position of the
    query although a zero scalar never reaches the kernel).  `blended` is round 2's single number over both.
    pmc: {"traffic": bytes per launch, "valu_per_add": ..., "source": ...} measured for THIS workload, or None."""
    dense, wit = _acc_class(prof, "msm_accumulate_g1_dense"), _acc_class(prof, "msm_accumulate_g1", hbm=False)
    main = dense or _acc_class(prof, "msm_accumulate_g1")
    if not main:
        return None
    roof = {"bound": "hbm", "kernel": "k_accumulate<Fq> (G1 bucket accumulation), launches of the h multiexp" if dense
            else "k_accumulate<Fq> (G1 bucket accumulation)",
            "achieved": main["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": main["frac"],
            "frac_of_measured_copy_ceiling": main["achieved"] / 6290.0,
            "traffic": pmc["traffic"] if pmc else None, "traffic_source": pmc["source"] if pmc else None,
            "launches": main["launches"], "avg_launch_ms": main["avg_launch_ms"],
            "algorithmic_bytes_per_launch": main["algorithmic_bytes_per_launch"],
            "by_job_class": {"h_dense": dense, "witness": wit}}
    if dense and wit:
        ms = prof["msm_accumulate_g1_dense"][0] + prof["msm_accumulate_g1"][0]
        nb = prof["msm_accumulate_g1_dense"][2] + prof["msm_accumulate_g1"][2]
        nl = prof["msm_accumulate_g1_dense"][1] + prof["msm_accumulate_g1"][1]
        roof["by_job_class"]["blended"] = {"launches": nl, "avg_launch_ms": ms / nl,
                                           "note": "both classes are ONE kernel name: this average launch time is the one a "
                                                   "rocprofv3 --stats summary of the same command shows for k_accumulate<Fq> (%s); "
                                                   "no HBM fraction is given for it or for the witness class: only the h launches' "
                                                   "points all reach the kernel" % blended_note()}
    # secondary, integer-issue ceiling: the kernel is VALU bound.  One mixed add = 8 mul + 2 sqr
    # in the 9x29 field = 8*162 + 2*126 v_mad_u64_u32 (minus 81 where two products share a reduction).
    roof["alu_secondary"] = {"unit": "lane v_mad_u64_u32 / s", "peak_measured": VALU_ISSUE_PEAK, "peak_nominal": VALU_ISSUE_NOMINAL,
                             "peak_source": VALU_PEAK_NOTE, "mads_per_mixed_add": MADS_PER_MIXED_ADD,
                             "frac_of_measured": main["alu_frac_of_measured_issue_rate"],
                             "frac_of_nominal": main["alu_frac_of_nominal_issue_rate"]}
    if pmc and pmc.get("valu_per_add"):
        roof["alu_secondary"]["valu_instructions_per_mixed_add"] = pmc["valu_per_add"]
        roof["alu_secondary"]["valu_source"] = pmc["source"]
    return roof


def blended_note():
    """Where the committed rocprofv3 summary of this round puts the same average (profiles/rNN_m_kernel_stats_2p22_default_bench.csv
    against the events of the run it was taken under, rNN_m_bench_under_rocprof.json) -- read from the files, newest round first."""
    import csv
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_m_kernel_stats_2p22_default_bench.csv")), reverse=True):
        rnd = os.path.basename(f).split("_")[0]
        try:
            tot, calls = 0.0, 0
            for row in csv.DictReader(open(f)):
                if "k_accumulate<" in row["Name"] and "FqParams" in row["Name"] and "FqL" in row["Name"]:
                    tot += float(row["TotalDurationNs"])
                    calls += int(row["Calls"])
            under = json.load(open(os.path.join(ROOT, "profiles", "%s_m_bench_under_rocprof.json" % rnd)))
            ev = under["roofline"]["by_job_class"]["blended"]["avg_launch_ms"] * 1e3
            return "profiles/%s_m_*: %.1f us in the csv, %.1f us from the events of that same run" % (rnd, tot / calls / 1e3, ev)
        except Exception:
            continue
    return "no committed summary found under profiles/"


def host_pack_threads():
    """zg_prove's host-side packing threads (csrc/prover.hip host_pack_threads: the default of ZG_OPT_PACK_THREADS)."""
    hc = os.cpu_count() or 1
    return 0 if hc < 8 else min(16, hc // 2)


def pmc_for(args):
    """PMC figures are collected in separate rocprofv3 --pmc passes (tools/pmc_*.py) and only describe the workload they
    were taken on: the default one.  Any other workload gets None -- nothing is pasted across workloads."""
    if not (args.workload == "synthetic" and args.log_m == 22 and abs(args.bit_fraction - 0.97) < 1e-9 and args.seed == 0x5A):
        return None
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % rnd)))
            vi = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_valu_instructions.json" % rnd)))
        except Exception:
            continue
        t = tr.get("msm_accumulate_g1_dense_bytes_per_launch") or tr.get("msm_accumulate_g1_bytes_per_launch")
        return {"traffic": t, "valu_per_add": vi.get("k_accumulate_g1_valu_instructions_per_wave_mixed_add"),
                "source": "profiles/%s_pmc_traffic.json + profiles/%s_pmc_valu_instructions.json (separate rocprofv3 --pmc passes of "
                          "this workload, 2^22 config 4; not measured in this run)" % (rnd, rnd)}
    return None


def ntt_roofline(m, passes, kernel_ms, launches, h_ms_live, eval_form=True, c_folded=False):
    """The H pipeline against the HBM roofline: algorithmic 64 B per element per transform."""
    n_transforms = (4 if c_folded else 6) if eval_form else 7
    alg = n_transforms * 64 * m
    out = {"bound": "hbm", "kernel": ("k_ntt_pass / k_ntt_mid / k_ntt_tail3 (H in evaluation form%s: %%d transforms in %%d launches)"
                                      % (", C rows folded into the l multiexp; the launch count includes c's passes, which are queued and "
                                         "leave at once when the device decides that the proof folds" if c_folded else "")
                                      if eval_form else "k_ntt_pass (H in coefficient form, bellman's %d transforms, in %d launches)")
           % (n_transforms, passes), "algorithmic_bytes": alg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "h_pipeline_ms_in_proof": h_ms_live,
           "achieved_in_proof": alg / (h_ms_live * 1e-3) / 1e9 if h_ms_live else None,
           "frac_in_proof": alg / (h_ms_live * 1e-3) / 1e9 / HBM_PEAK_GBS if h_ms_live else None,
           "note": "h_pipeline_ms_in_proof = HIP-event time of the transforms inside the timed proofs (they share the GPU with "
                   "the witness multiexps); kernel_ms = sum of the transform kernels' own event times in the instrumented proofs"}
    if kernel_ms and launches:
        out.update({"kernel_ms": kernel_ms, "launches": launches, "achieved": alg / (kernel_ms * 1e-3) / 1e9,
                    "frac": alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`bench.py --gpus N` without a launcher: N ranks through `python -m torch.distributed.run` as a child process."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", os.environ.get("MASTER_PORT") or str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between the ranks on this image
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` the way the driver runs N = 1: start the N ranks as CHILD processes (one per GPU,
        # torch.distributed.run over RCCL) before anything here has touched the GPU or imported torch, relay their output
        # (rank 0 prints the JSON line) and leave with the launcher's exit code.  Never an exec.
        sys.exit(self_launch(args.gpus))
    import torch
    import za_amd
    from za_amd import synth
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (there is no CPU fallback)", file=sys.stderr)
        sys.exit(1)
    # ZA_BENCH_DEVICE pins every rank to one device: lets the N > 1 code path be exercised on a 1-GPU box
    # (with --dist-backend gloo; RCCL itself needs one GPU per rank)
    if "ZA_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["ZA_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = za_amd.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- workload (untimed set-up): R1CS + proving key resident in HBM, witness in pinned host memory
    t_setup = time.time()
    if args.workload == "poseidon":
        from za_amd import workloads

        class _W:      # the fields of synth.SynthR1CS the rest of the script uses
            pass
        sr = _W()
        sr.n_inputs, sr.n_aux, cons, wit, hashes = workloads.poseidon_chain(args.log_m)
        sr.n_constraints = len(cons)
        rp, cl, cf = workloads.csr_arrays(cons)
        sr.nnz = [len(c) for c in cl]
        sr.witness = workloads.witness_bytes(wit)
        sr.close = lambda: None
        r1 = ctx.r1cs_load(sr.n_inputs, sr.n_aux, sr.n_constraints, rp, cl, cf)
        n_perm = len(hashes)
        del cons, wit, rp, cl, cf
        if args.synthetic_key or world > 1:
            print("--workload poseidon runs on one GPU with a key from zg_setup", file=sys.stderr)
            sys.exit(2)
    elif args.workload == "eddsa":
        from za_amd import workloads

        class _W:
            pass
        sr = _W()
        sr.n_inputs, sr.n_aux, cons, wit = workloads.eddsa_poseidon_verifier()
        sr.n_constraints = len(cons)
        rp, cl, cf = workloads.csr_arrays(cons)
        sr.nnz = [len(c) for c in cl]
        sr.witness = workloads.witness_bytes(wit)
        sr.close = lambda: None
        r1 = ctx.r1cs_load(sr.n_inputs, sr.n_aux, sr.n_constraints, rp, cl, cf)
        args.log_m = r1.log_m
        del cons, wit, rp, cl, cf
        if args.synthetic_key or world > 1:
            print("--workload eddsa runs on one GPU with a key from zg_setup", file=sys.stderr)
            sys.exit(2)
    elif args.workload == "sha256":
        from za_amd import workloads

        class _W:
            pass
        sr = _W()
        sr.n_inputs, sr.n_aux, sr.n_constraints, rp, cl, cf, wit, sha_info = workloads.sha256_merkle(args.depth)
        sr.nnz = sha_info["nnz"]
        sr.witness = wit
        sr.close = lambda: None
        r1 = ctx.r1cs_load(sr.n_inputs, sr.n_aux, sr.n_constraints, rp, cl, cf)
        args.log_m = r1.log_m
        del rp, cl, cf, wit
        if args.synthetic_key or world > 1:
            print("--workload sha256 runs on one GPU with a key from zg_setup", file=sys.stderr)
            sys.exit(2)
    else:
        sr = synth.SynthR1CS(args.log_m, 2, args.bit_fraction, args.seed)
        r1 = sr.load(ctx)
    m = 1 << args.log_m
    n_a = sr.n_inputs + r1.a_aux_total
    n_b = r1.b_input_total + r1.b_aux_total
    full_key = None
    if args.synthetic_key:
        # bases = k_i * G with known discrete logs (tests use this form); same sizes, same prover work
        P = synth.synth_params(ctx, m - 1, sr.n_aux, n_a, n_b, 7000, rank=rank, world=world)
    else:
        # a valid proving key from the GPU trapdoor set-up (fixed toxic waste so runs are reproducible)
        P = ctx.setup(r1, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
        if world > 1:
            full = P
            P = synth.params_shard(ctx, full, rank, world)
            full.close()
        else:
            full_key = P
    w_host = ctx.host_alloc(len(sr.witness))     # pinned: the H2D copy inside every step runs at PCIe speed
    w_host[:] = sr.witness if isinstance(sr.witness, np.ndarray) else np.frombuffer(sr.witness, dtype=np.uint8)
    witness_bytes = int(w_host.size)
    d_w = ctx.upload(w_host)
    def mem_note(what):   # ZA_BENCH_MEM=1: free HBM at the set-up's milestones, to stderr
        if os.environ.get("ZA_BENCH_MEM"):
            f, t = torch.cuda.mem_get_info()
            print("[bench mem] %-18s free %.1f GB of %.1f" % (what, f / 1e9, t / 1e9), file=sys.stderr, flush=True)
    mem_note("key + witness")
    outer_line, t_outer = None, 0.0
    if world == 1 and not args.no_outer and args.workload == "synthetic" and not args.synthetic_key:
        t_o = time.time()
        try:
            outer_line = outer_measurements(ctx, args, sr, r1, P, w_host)
        except Exception as e:   # the bench line must still come out
            outer_line = {"error": repr(e)}
        t_outer = time.time() - t_o
    t_prep = time.time()
    P.prepare(r1, d_w)         # fixed-base window tables (of this rank's shard): once per key, not per proof
    t_prep = time.time() - t_prep
    mem_note("tables prepared")
    n_constraints = sr.n_constraints
    nnz = list(sr.nnz)
    n_aux = sr.n_aux
    sr.close()
    t_setup = time.time() - t_setup - t_outer
    r_blind, s_blind = 0x1234567, 0x7654321

    phase_acc = {}
    if world > 1:
        from za_amd.dist import GpuEngine, sharded_prove
        engine = GpuEngine(ctx, P, r1, d_w, world)
        upload_sent = [0]

        def step():
            # every rank's copy of the witness arrives over its own PCIe link, inside the step, the way zg_prove hands it
            # over (packed by host threads when it is mostly 0 / 1)
            engine.d_witness, sent = ctx.witness_to_device(w_host)
            upload_sent[0] = sent
            tm = {}
            out = sharded_prove(engine, r_blind, s_blind, timings=tm)
            for k, v in tm.items():
                phase_acc[k] = phase_acc.get(k, 0.0) + v
            return out
    else:
        def step():
            return ctx.prove(P, r1, w_host, r_blind, s_blind)

    proof = None
    # level 1: HIP events around the accumulation kernels only (the roofline's kernel).  Timing every kernel class
    # costs 0.5 ms per proof in event records; the full breakdown comes from a few extra untimed proofs below.
    ctx.profile_enable(1)         # on during warm-up too, so the events are created outside the timed region
    # Part of the untimed set-up, like the tables: a few proofs until the library's pools are warm (HIP events of the
    # profiling scopes are recycled one proof late, helper threads' stacks, pinned read-back buffers) -- without them one
    # of the first three timed steps carries ~1.5 ms of one-time work (`step_ms` shows the per-step times).
    t_first = time.perf_counter()
    proof = step()
    t_first = time.perf_counter() - t_first     # a key's first proof: the one-time work zg_params_prepare left (if any)
    for _ in range(3):
        proof = step()
    for _ in range(args.warmup):
        proof = step()
    ctx.profile_reset()
    phase_acc.clear()
    barrier()
    t0 = time.perf_counter()
    step_ms = []
    h_ms_acc = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        proof = step()
        step_ms.append(round(1e3 * (time.perf_counter() - ts), 3))
        if world == 1:
            h_ms_acc.append(ctx.last_timings()["h_ms"])   # (a ctypes read of six doubles, outside the step's clock)
    barrier()
    elapsed = time.perf_counter() - t0
    try:
        proof_stats = ctx.last_stats() if world == 1 else None   # of the last timed step: zg_prove counts its own launches
    except AttributeError:   # (ZA_AMD_LIB pointing at an older build of the library: A/B runs)
        proof_stats = None
    prof = ctx.profile_get()
    # the tables the TIMED proofs used (the secondary measurements below go through the plain phase API, which builds the plain
    # l query's table for a folded key on demand: it is not what zg_prove reads)
    key_tables = {name: {"kind": ("none", "window", "subset-sum")[P.table_info(k)[0]], "gb": P.table_info(k)[1] / 1e9}
                  for k, name in enumerate(("h", "l", "a", "b_g1", "b_g2", "l_with_c_rows")) if k < 5 or P.table_info(k)[0]}
    phases = dict(ctx.last_timings())
    upload_bytes = int(upload_sent[0]) if world > 1 else int(phases.get("upload_bytes", 0))
    phases.pop("upload_bytes", None)   # (a byte count, reported under its own key)
    # the form of H the timed proofs took: the key's (a sharded run: what the ranks agreed on)
    h_eval_form = bool(getattr(engine, "h_eval", False)) if world > 1 else bool(P.h_eval_info()[0])
    h_ms_live = float(np.mean(h_ms_acc)) if h_ms_acc else None
    phase_ms = {k: v / max(args.steps, 1) for k, v in phase_acc.items()}
    elapsed = max_over_ranks(elapsed)
    dist_line = None
    if world > 1:
        # the north-star's multi-GPU target is MSM scaling: per-proof MSM phase = witness MSMs + h MSM, max over ranks
        msm_phase = max_over_ranks(phase_ms.get("witness_msm", 0.0) + phase_ms.get("h_msm", 0.0))
        h_msm = max_over_ranks(phase_ms.get("h_msm", 0.0))
        dist_line = {"msm_phase_ms": msm_phase, "h_msm_ms": h_msm,
                     "witness_msm_ms_rank0_beside_ntt": phase_ms.get("witness_msm"),
                     "h_ntt_ms": phase_ms.get("h_total"), "scatter_ms": phase_ms.get("scatter"),
                     "gather_ms": phase_ms.get("gather"), "combine_ms": phase_ms.get("combine"),
                     "h_form": ("evaluation, C rows folded on rank 0 (4 transforms, transformed h bases)" if getattr(engine, "c_folded", False)
                                else "evaluation (6 transforms, transformed h bases)") if getattr(engine, "h_eval", False)
                               else "coefficient (7 transforms)",
                     "note": "host wall times per proof; h_ntt_ms = rank 0's R1CS evaluation + H transforms from start to "
                             "finish (its witness MSMs run beside them); msm_phase_ms = max over ranks of witness MSMs + "
                             "h MSM of the rank's 1/%d shard" % world}

    # secondary: the same proofs with the witness already in HBM (round 1's headline form)
    if world == 1:   # (untimed: a proof of the other entry point re-sizes the l multiexp's sort buffers once, msm.hip msm_scratch_trim)
        ctx.prove_dev(P, r1, d_w, r_blind, s_blind)
        torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    n_dev = max(3, args.steps // 2)
    if world == 1:
        for _ in range(n_dev):
            ctx.prove_dev(P, r1, d_w, r_blind, s_blind)
        torch.cuda.synchronize()
        prove_dev_ms = 1e3 * (time.perf_counter() - t1) / n_dev
    else:
        prove_dev_ms = None

    # secondary: zg_prove_packed -- the same witness handed over in packed form (2 bits per value + the values that are
    # not 0 / 1), from pinned host memory, unpacked on the device: what the upload costs when the glue packs it
    packed_line = None
    if world == 1:
        pk = za_amd.pack_witness(w_host)
        pinned = []
        for a in pk:
            raw = np.ascontiguousarray(a).view(np.uint8)
            buf = ctx.host_alloc(max(raw.size, 4))
            buf[:raw.size] = raw
            pinned.append(buf)
        packed = (pinned[0][:pk[0].size], pinned[1][:4 * pk[1].size].view(np.uint32), pinned[2][:pk[2].size])
        proof_p = ctx.prove_packed(P, r1, packed, r_blind, s_blind)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_dev):
            proof_p = ctx.prove_packed(P, r1, packed, r_blind, s_blind)
        torch.cuda.synchronize()
        packed_line = {"ms_per_proof": 1e3 * (time.perf_counter() - t1) / n_dev,
                       "packed_bytes": int(pk[0].size + 36 * pk[1].size), "witness_bytes": witness_bytes,
                       "same_proof": bytes(proof_p) == bytes(proof) if proof is not None else None,
                       "note": "zg_prove_packed: 2-bit classes + (index, value) of the values that are not 0 / 1, from "
                               "pinned host memory; the packing itself (zg_witness_pack) is the caller's, like producing "
                               "the 32-byte form"}
        for b in pinned:
            ctx.host_free(b)

    # per-class kernel times: a separate, untimed, fully instrumented pass
    ctx.profile_enable(2)
    ctx.profile_reset()
    n_break = 3
    for _ in range(n_break):
        step()
    barrier()
    prof_all = ctx.profile_get()
    ctx.profile_enable(0)

    conc_line = None
    if args.provers > 1 and world == 1:
        # throughput with several provers sharing the GPU (one zg_ctx + host thread each, same resident key, each
        # uploading its witness from pinned memory): one prover's upload and NTT phase fill the other's sort /
        # reduction tails
        import threading
        ctxs = [ctx] + [za_amd.Context(local_rank) for _ in range(args.provers - 1)]
        conc_error = None
        per = max(4, args.steps)
        outs = [None] * args.provers

        def work(i, k):
            for _ in range(k):
                outs[i] = ctxs[i].prove(P, r1, w_host, r_blind, s_blind)
        try:
            for i in range(args.provers):
                work(i, 1)
        except za_amd.ZaAmdError as e:
            # every prover needs its own workspace (sort buffers, evaluation vectors: 65 GB at 2^26 with a dense witness):
            # at the largest sizes three of them do not fit beside the key's tables
            args.provers = 1
            conc_error = repr(e)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i, per)) for i in range(args.provers if args.provers > 1 else 0)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        conc_line = {"skipped": conc_error} if args.provers == 1 else {"provers": args.provers, "proofs": per * args.provers, "ms_per_proof": 1e3 * el / (per * args.provers),
                     "constraints_per_s": n_constraints * per * args.provers / el, "witness": "host (pinned), uploaded per proof",
                     "identical_proofs": all(bytes(o) == bytes(proof) for o in outs)}
        for c in ctxs[1:]:
            c.close()

    shard_line = None
    if world == 1 and args.shard_model > 1 and full_key is not None and args.workload == "synthetic":
        # What ONE rank of a W-way sharded proof does, measured here: the MSM phase over a 1/W shard of every query
        # (with its own tables), and the unsharded R1CS evaluation + H pipeline that stays on rank 0.
        W = args.shard_model
        Pk = synth.params_shard(ctx, full_key, 0, W)
        Pk.prepare(r1, d_w)
        d_h = ctx.alloc(32 * (m + W))
        # the form a sharded run takes (za_amd/dist.py): evaluation form when the shard has its slice of the transformed h bases
        ev = Pk.h_eval_info()[0] and full_key.h_eval_info()[0]
        h_fn = ctx.prove_h_eval_dev if ev else ctx.prove_h_dev
        h_mask = 0x01 | (za_amd.ZG_MASK_H_EVAL if ev else 0)
        h_fn(r1, d_w, d_h)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            h_fn(r1, d_w, d_h)
            t2 = time.perf_counter()
            ctx.prove_partials_dev(Pk, r1, d_w, d_h, 0x1e)
            t3 = time.perf_counter()
            ctx.prove_partials_dev(Pk, r1, d_w, d_h, h_mask)      # rank 0's range starts at position 0
            t4 = time.perf_counter()
            ts.append((t2 - t1, t3 - t2, t4 - t3))
        ts = np.array(ts[1:]) * 1e3
        full_w = []
        for _ in range(3):
            t1 = time.perf_counter()
            ctx.prove_partials_dev(full_key, r1, d_w, d_h, 0x1e)
            t2 = time.perf_counter()
            ctx.prove_partials_dev(full_key, r1, d_w, d_h, h_mask)
            full_w.append((t2 - t1, time.perf_counter() - t2))
        full_w = np.array(full_w[1:]) * 1e3
        # The same phases as they run in the real flow (za_amd/dist.py): H queued, the witness multiexps beside it, the h
        # multiexp once H is there.  What the multiexps add to the rank's critical path is that wall time minus H's own.
        enq = ctx.prove_h_eval_enqueue_dev if ev else ctx.prove_h_enqueue_dev
        flow = {}
        for name, key in (("shard", Pk), ("unsharded", full_key)):
            tt = []
            for _ in range(4):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                enq(r1, d_w, d_h)
                ctx.prove_partials_dev(key, r1, d_w, d_h, 0x1e)
                ctx.prove_partials_dev(key, r1, d_w, d_h, h_mask)
                tt.append(time.perf_counter() - t1)
            flow[name] = 1e3 * float(np.mean(tt[1:]))
        h_alone = float(ts[:, 0].mean())
        # ... and with the C rows folded on rank 0 (zg_prove_h_eval_enqueue_fold_dev: four transforms, rank 0's l partial
        # carries the rows' multiexp; the other ranks' phases are the ones above)
        fold = None
        if ev and hasattr(ctx, "prove_h_eval_enqueue_fold_dev"):
            th, tf, was = [], [], False
            for _ in range(4):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                was = ctx.prove_h_eval_enqueue_fold_dev(Pk, r1, d_w, d_h)
                ctx.sync()
                th.append(time.perf_counter() - t1)
                ctx.prove_partials_dev(Pk, r1, d_w, d_h, 0x1e)          # (takes the armed rows)
            for _ in range(4):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ctx.prove_h_eval_enqueue_fold_dev(Pk, r1, d_w, d_h)
                ctx.prove_partials_dev(Pk, r1, d_w, d_h, 0x1e)
                ctx.prove_partials_dev(Pk, r1, d_w, d_h, h_mask)
                tf.append(time.perf_counter() - t1)
            fold = {"folded": bool(was), "h_ntt_ms": 1e3 * float(np.mean(th[1:])), "flow_ms": 1e3 * float(np.mean(tf[1:]))}
        shard_line = {"world": W, "h_form": "evaluation" if ev else "coefficient", "h_ntt_ms": h_alone, "witness_msm_ms": float(ts[:, 1].mean()),
                      "h_msm_ms": float(ts[:, 2].mean()), "msm_phase_ms": float(ts[:, 1:].sum(axis=1).mean()),
                      "unsharded_msm_phase_ms": float(full_w.sum(axis=1).mean()),
                      "msm_scaling_model": float(full_w.sum(axis=1).mean() / ts[:, 1:].sum(axis=1).mean()),
                      "flow_ms": flow["shard"], "unsharded_flow_ms": flow["unsharded"],
                      "msm_critical_path_ms": max(flow["shard"] - h_alone, 0.0),
                      "unsharded_msm_critical_path_ms": max(flow["unsharded"] - h_alone, 0.0),
                      "msm_critical_path_scaling": (flow["unsharded"] - h_alone) / max(flow["shard"] - h_alone, 1e-9),
                      "rank0_with_c_rows_folded": fold,
                      "note": "one GPU.  msm_phase_ms / msm_scaling_model: the phases run one after the other (no overlap), rank "
                              "0's shard of a proof sharded %d ways vs the same phases unsharded.  flow_ms: the phases as the "
                              "sharded flow runs them on rank 0 (H queued, witness multiexps beside it, h multiexp behind it); "
                              "msm_critical_path_ms = flow_ms - h_ntt_ms.  The scatter of (m*32 B)/%d per xGMI link is not "
                              "included.  No multi-GPU hardware measurement exists: these are one-GPU models" % (W, W)}
        ctx.free(d_h)
        Pk.close()

    sha_line = None
    if world == 1 and not args.no_sha and args.workload == "synthetic" and args.log_m >= 21 and not args.synthetic_key:
        # BASELINE config 4 as it is named -- a depth-32 SHA-256 Merkle inclusion proof on a bit-level circuit (1.48 M
        # constraints, 2^21 domain); root and witness checked against hashlib by the generator, proof checked by the
        # pairing equation with that root as the public input
        from za_amd import workloads
        ni, na, nc, rp, cl, cf, wit, info = workloads.sha256_merkle(32)
        rs = ctx.r1cs_load(ni, na, nc, rp, cl, cf)
        Ps, pbytes = ctx.setup(rs, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172)
        ws = ctx.host_alloc(len(wit))
        ws[:] = np.frombuffer(wit, dtype=np.uint8)
        for _ in range(2):
            proof_s = ctx.prove(Ps, rs, ws, r_blind, s_blind)
        torch.cuda.synchronize()
        ns = max(3, args.steps // 2)
        t1 = time.perf_counter()
        for _ in range(ns):
            proof_s = ctx.prove(Ps, rs, ws, r_blind, s_blind)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        root = info["root"]
        ok = za_amd.verify(pbytes, proof_s, [int.from_bytes(root[:16], "big"), int.from_bytes(root[16:], "big")])
        sha_line = {"depth": 32, "n_constraints": nc, "log_m": rs.log_m, "ms_per_proof": 1e3 * el / ns,
                    "constraints_per_s": nc * ns / el, "proof_verifies_for_the_hashlib_root": bool(ok),
                    "note": "za_amd/workloads.py::sha256_merkle (bit-level SHA-256, two compressions per level); zg_prove "
                            "from pinned host memory; --workload sha256 --depth 90 fills the 2^22 domain"}
        ctx.host_free(ws)
        Ps.close(); rs.close()
        del rp, cl, cf, wit, pbytes

    dense_line = None
    if world == 1 and not args.no_dense and args.workload == "synthetic" and args.bit_fraction > 0:
        # the same size with a witness of uniform field elements: no 0/1 shortcut in the R1CS evaluation, no subset-sum
        # tables, five full-width Pippenger jobs (window tables on all five queries)
        sd = synth.SynthR1CS(args.log_m, 2, 0.0, args.seed)
        rd = sd.load(ctx)
        Pd = ctx.setup(rd, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]
        wd = ctx.host_alloc(len(sd.witness))
        wd[:] = sd.witness
        for _ in range(2):
            ctx.prove(Pd, rd, wd, r_blind, s_blind)
        torch.cuda.synchronize()
        nd = max(3, args.steps // 2)
        t1 = time.perf_counter()
        for _ in range(nd):
            ctx.prove(Pd, rd, wd, r_blind, s_blind)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        dense_line = {"ms_per_proof": 1e3 * el / nd, "constraints_per_s": sd.n_constraints * nd / el,
                      "tables_gb": sum(Pd.table_info(k)[1] for k in range(5)) / 1e9,
                      "note": "bit fraction 0: every witness value and coefficient a uniform field element (config 5's shape at "
                              "this size); zg_prove from pinned host memory"}
        ctx.host_free(wd)
        Pd.close(); rd.close(); sd.close()

    msm_line = None
    if not args.no_msm:
        # second half of BASELINE.json's metric: G1 MSM points/s -- bases resident in HBM (device form + window table),
        # uniform scalars uploaded from pinned host memory inside the timed region.  N > 1: the 2^n bases are sharded
        # over the ranks, every rank uploads and multiplies its shard, the 96-byte partial sums are all-gathered and
        # added (RCCL has no point-add reduction).
        n_total = 1 << args.msm_log_n
        n_mine = n_total // world
        d_b = synth.synth_bases_g1(ctx, n_mine, 4242 + (rank << 32))
        d_s = synth.synth_scalars(ctx, n_mine, 99 + rank)
        H = ctx.msm_bases_g1_dev(d_b, n_mine, table_budget_gb=120.0)
        ctx.free(d_b)
        s_host = ctx.host_alloc(32 * n_mine)
        s_host[:] = ctx.d2h(d_s, 32 * n_mine)
        binfo = ctx.msm_bases_info(H)

        def msm_step(resident):
            part = ctx.msm_resident_dev(H, d_s, n_mine, True) if resident else ctx.msm_resident(H, s_host, n_mine, True)
            if world > 1:
                t = torch.from_numpy(np.frombuffer(part, dtype=np.uint8).copy()).to("cuda" if args.dist_backend == "nccl" else "cpu")
                gathered = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(gathered, t)
                if rank == 0:
                    return za_amd.g1_sum(torch.cat(gathered).cpu().numpy())
                return None
            return part
        msm_step(False); msm_step(True)
        ctx.profile_enable(1)
        ctx.profile_reset()
        k_msm = 5
        barrier()
        t1 = time.perf_counter()
        for _ in range(k_msm):
            msm_step(False)
        barrier()
        el_up = max_over_ranks(time.perf_counter() - t1)
        prof_msm = ctx.profile_get()
        ctx.profile_enable(0)
        barrier()
        t1 = time.perf_counter()
        for _ in range(k_msm):
            msm_step(True)
        barrier()
        el_res = max_over_ranks(time.perf_counter() - t1)
        msm_line = {"points_per_s": k_msm * n_total / el_up, "points": n_total, "unit": "points/s", "ms_per_msm": 1e3 * el_up / k_msm,
                    "points_per_s_scalars_resident": k_msm * n_total / el_res, "ms_per_msm_scalars_resident": 1e3 * el_res / k_msm,
                    "sharding": "%d ranks x 2^%d/%d bases" % (world, args.msm_log_n, world) if world > 1 else "single GPU",
                    "table": {"gb": binfo["table_bytes"] / 1e9, "c": binfo["c"], "digits": binfo["nwin"]},
                    "roofline": acc_roofline(prof_msm),   # (no PMC passes exist for this workload: traffic = null)
                    "note": "zg_msm_resident: 2^%d G1 bases resident (device form + fixed-base window table), uniform 253-bit "
                            "scalars uploaded from pinned host memory in 4 chunks overlapped with the 4 dependent parts of the "
                            "multiexp; *_scalars_resident = zg_msm_resident_dev" % args.msm_log_n}
        ctx.host_free(s_host)
        ctx.msm_bases_free(H)
        ctx.free(d_s)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(args.steps, 1)
        roof = acc_roofline(prof, pmc=pmc_for(args))
        ntt_ms, ntt_n = prof_all["ntt_pass"][0] / n_break, prof_all["ntt_pass"][1] / n_break
        c_folded = bool(proof_stats and proof_stats.get("c_folded"))
        roof_ntt = ntt_roofline(m, int(round(ntt_n)), ntt_ms, int(round(ntt_n)), h_ms_live, h_eval_form, c_folded) if world == 1 else None
        line = {
            "metric": "Groth16 prove constraints/sec (BN128)", "value": n_constraints * args.steps / elapsed,
            "unit": "constraints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32x8 Montgomery (256-bit modular integer)", "data": "synthetic",
            "config": {"workload": ("Poseidon(t=6, RF=8, RP=57, circomlib parameters) hash chain of %d permutations, config 3 "
                                    "shape, padded to n_constraints + n_inputs = 2^%d, proving key from zg_setup"
                                    % (n_perm, args.log_m)) if args.workload == "poseidon" else
                                   ("SHA-256 Merkle inclusion, depth %d (bit-level circuit, two compressions per level, root "
                                    "= 2 public inputs; witness and root checked against hashlib), config 4 on a real "
                                    "circuit, domain 2^%d, proving key from zg_setup" % (args.depth, args.log_m))
                                   if args.workload == "sha256" else
                                   ("EdDSA-Poseidon signature verification on Baby Jubjub (config 2's statement, circomlib "
                                    "parameters, the reference's test signature as witness), domain 2^%d, proving key "
                                    "from zg_setup" % args.log_m) if args.workload == "eddsa" else
                                   "synthetic R1CS, config 4 shape (SHA256-Merkle-like: %.0f%% bit witness, SplitMix64 "
                                   "seed 0x%x), n_constraints + n_inputs = 2^%d, %s"
                                   % (100 * args.bit_fraction, args.seed, args.log_m,
                                      "synthetic proving key with known discrete logs" if args.synthetic_key
                                      else "proving key from zg_setup (valid CRS, fixed toxic waste)"),
                       "step": "%s: witness (%d bytes, 32 B per value) in pinned host memory, handed over inside every step; "
                               "R1CS and key resident.  %s; %s"
                               % ("zg_prove" if world == 1 else "sharded proof (za_amd/dist.py), every rank takes the whole witness",
                                  witness_bytes,
                                  "shared between the copy engine (from the front, as it is) and %d packing host threads "
                                  "(from the back) until they meet (%d bytes sent in the last step)"
                                  % (host_pack_threads(), upload_bytes)
                                  if 0 < upload_bytes < witness_bytes else
                                  "sent over PCIe as it is in the last step (packing applies from 2^18 values, to a mostly-0/1 "
                                  "witness)",
                                  "H in evaluation form, the C rows' values folded into the l multiexp (4 transforms; h bases "
                                  "transformed once per key, twice)" if h_eval_form and c_folded else
                                  "H in evaluation form (6 transforms; h bases transformed once per key)" if h_eval_form else
                                  "H in coefficient form (bellman's 7 transforms: this key has no transformed h bases)"),
                       "n_constraints": n_constraints, "n_aux": n_aux, "nnz": nnz,
                       "parallelism": "msm-sharded x%d, ntt on rank 0" % world if world > 1 else "single GPU"},
            "roofline": roof,
            "roofline_ntt": roof_ntt,
            "step_ms": step_ms, "ms_median": float(np.median(step_ms)) if step_ms else None,
            # the whole proof against the HBM roofline with the survey's algorithmic bytes
            # 96(m-1) + 96 v + 96(l + d_A) + (96 + 160) d_B + 7 * 64 m (bellman's seven transforms)
            "prove_hbm": (lambda b: {"algorithmic_bytes": b, "achieved": b / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                                     "frac": b / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS})(
                96 * (m - 1) + 96 * n_aux + 96 * n_a + 256 * n_b + 448 * m),
            "prove_dev_ms": prove_dev_ms,
            "prove_packed": packed_line,
            "phases_ms_last_step": phases,
            "upload_bytes_last_step": upload_bytes,
            # kernel launches one zg_prove queues (hand-over kernels, R1CS evaluation, transforms, the five multiexps), and which
            # witness multiexps shared a digit sort (zg_last_stats; ZG_OPT_SHARE_SORT)
            "launches_per_proof": proof_stats["launches"] if proof_stats else None,
            "c_rows_folded": c_folded if proof_stats else None,
            "digit_sort_sharing": {"b_g1_behind_b_g2": proof_stats["b_g1_behind_b_g2"],
                                   "one_sort_over_witness_positions_mask": proof_stats["shared_sort_mask"]} if proof_stats else None,
            "kernel_ms_per_step": {k: v[0] / n_break for k, v in prof_all.items()},
            # the h multiexp's phases (HIP events on its streams inside the profiled proofs; from 2^22 points the job runs as two
            # parts on two streams, whose times are ADDED here: they overlap each other and the end of the transforms)
            "h_msm": {"sort_ms": prof_all["msm_digit_sort_dense"][0] / n_break,
                      "accumulate_ms": prof_all["msm_accumulate_g1_dense"][0] / n_break,
                      "reduce_ms": prof_all["msm_bucket_reduce_dense"][0] / n_break,
                      "accumulate_launches_per_proof": prof_all["msm_accumulate_g1_dense"][1] / n_break}
                     if "msm_digit_sort_dense" in prof_all else None,
            "setup_s": t_setup,
            "per_key_preparation": {"prepare_s": t_prep, "first_proof_ms": 1e3 * t_first,
                                    "tables_gb": sum(t["gb"] for t in key_tables.values()),
                                    "note": "zg_params_prepare (evaluation-form h bases, folded C-row bases, window / subset-sum "
                                            "tables) and the first proof after it; what ZG_OPT_ONE_SHOT skips"},
            "key_tables": key_tables,
            "proof_sha": __import__("hashlib").sha256(bytes(proof)).hexdigest()[:16] if proof is not None else None,
        }
        if outer_line:
            line.update(outer_line)
            osh = outer_line.get("one_shot")
            if osh and osh["ms"] > ms_per_step:
                extra = t_prep + max(t_first - 1e-3 * ms_per_step, 0.0)
                line["per_key_preparation"]["break_even_proofs"] = int(np.ceil(extra / (1e-3 * (osh["ms"] - ms_per_step))))
                line["per_key_preparation"]["break_even_note"] = (
                    "proofs after which prepare_s + the first proof's extra is paid back by ms_per_step against one_shot.ms "
                    "(%.2f vs %.2f ms)" % (ms_per_step, osh["ms"]))
        if dist_line:
            line["sharded_phases"] = dist_line
        if msm_line:
            line["g1_msm"] = msm_line
        if dense_line:
            line["dense_witness"] = dense_line
        if sha_line:
            line["sha256_merkle"] = sha_line
        if shard_line:
            line["shard_model"] = shard_line
        if conc_line:
            line["concurrent_provers"] = conc_line
        if not args.no_cpu_baseline and world == 1 and args.workload in ("synthetic", "sha256"):
            try:
                line["cpu_baseline"] = cpu_baseline(ctx, args)
            except Exception as e:  # the bench line must still come out
                line["cpu_baseline"] = {"error": repr(e)}
        elif world > 1:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # explicit teardown (keeps profilers that hook process exit happy)
    try:
        ctx.free(d_w)
        ctx.host_free(w_host)
        P.close()
        r1.close()
        ctx.close()
    except Exception:
        pass
    sys.stdout.flush()


if __name__ == "__main__":
    main()
