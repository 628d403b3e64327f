"""Stream / hardware-queue stress for the prover: N concurrent provers (one zg_ctx + host thread each) proving the
same statement K times, every proof compared with the first one.  Run as a CHILD process with GPU_MAX_HW_QUEUES set
in its environment (the variable only counts before the first HIP call of a process):
    GPU_MAX_HW_QUEUES=2 python tools/queue_stress.py --provers 3 --proofs 300 --log-m 16 [--torch-first] [--h-parts 3]
Prints one JSON line; exit code 0 only if every proof is byte-identical.  tests/test_gpu_queues.py drives it."""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--provers", type=int, default=3)
    ap.add_argument("--proofs", type=int, default=300)
    ap.add_argument("--log-m", type=int, default=16)
    ap.add_argument("--bit-fraction", type=float, default=0.97)
    ap.add_argument("--torch-first", action="store_true", help="let torch initialise HIP before zg_init (then "
                    "zg_init's own setenv of GPU_MAX_HW_QUEUES comes too late: the runtime default applies)")
    ap.add_argument("--h-parts", type=int, default=0, help="ZG_OPT_H_PARTS for the h multiexp (0 = the library's choice)")
    args = ap.parse_args()
    if args.torch_first:
        import torch
        torch.zeros(16, device="cuda").sum().item()
    import numpy as np
    import za_amd
    from za_amd import synth
    ctxs = [za_amd.Context(0) for _ in range(args.provers)]
    for c in ctxs:
        c.set_option(za_amd.OPT_H_PARTS, args.h_parts)
    ctx = ctxs[0]
    sr = synth.SynthR1CS(args.log_m, 2, args.bit_fraction, 0x5A)
    R = sr.load(ctx)
    P, _ = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)
    w = ctx.host_alloc(len(sr.witness))
    w[:] = sr.witness
    want = ctx.prove(P, R, w, 5, 7)
    bad = [0] * args.provers
    err = [None] * args.provers

    def work(i):
        try:
            for _ in range(args.proofs):
                if ctxs[i].prove(P, R, w, 5, 7) != want:
                    bad[i] += 1
        except Exception as e:  # reported below
            err[i] = repr(e)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(args.provers)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    out = {"provers": args.provers, "proofs_each": args.proofs, "log_m": args.log_m, "seconds": round(el, 3),
           "ms_per_proof": round(1e3 * el / (args.provers * args.proofs), 3), "mismatches": sum(bad), "errors": [e for e in err if e],
           "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "torch_first": args.torch_first, "h_parts": args.h_parts}
    print(json.dumps(out), flush=True)
    ctx.host_free(w)
    P.close(); R.close(); sr.close()
    for c in ctxs:
        c.close()
    return 0 if not sum(bad) and not any(err) else 1


if __name__ == "__main__":
    sys.exit(main())
