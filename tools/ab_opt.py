"""A/B of a zg_set_option value inside ONE process (the pool's boxes and host loads differ by more than most effects, so runs
of separate processes cannot be compared):

    python tools/ab_opt.py OPTION v1,v2,... [--log-m 22] [--rounds 8] [--proofs 8] [--bits 0.97] [--keys] [--preset 3=4,5=0]

OPTION is a ZG_OPT_* number (include/za_amd.h; 3 = packing threads, 10 = the copy engine's share of the hand-over, 11 = early
combination, 14 = digit-sort schedule, 18 = h parts ...).  The values are alternated round by round; zg_prove_dev (witness
resident) and zg_prove (witness handed over) medians / minima per value.  --keys: for an option that is read when a key's
tables are BUILT (16 = subset-sum group width, 17 = compact window table, 7 = table mask): one key per value, built under
that value, proofs alternated between the keys.  From 2^24 the key has bases with known discrete logs (no set-up)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import za_amd
from za_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("option", type=int)
ap.add_argument("values")
if len(sys.argv) > 2 and sys.argv[2][:1] == "-" and sys.argv[2][1:2].isdigit():   # a list that starts with a negative value
    sys.argv[2] = " " + sys.argv[2]
ap.add_argument("--log-m", type=int, default=22)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--proofs", type=int, default=8)
ap.add_argument("--bits", type=float, default=0.97)
ap.add_argument("--keys", action="store_true")
ap.add_argument("--preset", default="")
a = ap.parse_args()
values = [int(v) for v in a.values.strip().split(",")]
ctx = za_amd.Context(0)
for kv in a.preset.split(","):
    if kv:
        ctx.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
lg = a.log_m
sr = synth.SynthR1CS(lg, 2, a.bits, 0x5A if a.bits > 0 else 0x26)
R = sr.load(ctx)


def make_key():
    if lg >= 24:
        return synth.synth_params(ctx, (1 << lg) - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 7000)
    return ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172, want_bytes=False)[0]


pin = ctx.host_alloc(len(sr.witness)); pin[:] = sr.witness
d_w = ctx.upload(sr.witness)
keys, ref = {}, None
shared = None if a.keys else make_key()
for v in values:
    ctx.set_option(a.option, v)
    keys[v] = make_key() if a.keys else shared
    for _ in range(3):
        p = ctx.prove(keys[v], R, pin, 5, 7)
    assert ref is None or p == ref, "proof bytes differ between the values"
    ref = p
    if a.keys:
        print("option %d = %d: tables %s" % (a.option, v, [keys[v].table_info(k) for k in range(5)]), flush=True)
dev = {v: [] for v in values}
host = {v: [] for v in values}
for _ in range(a.rounds):
    for v in values:
        ctx.set_option(a.option, v)   # (with --keys too: an option may be read both when the tables are built and per proof)
        ctx.prove_dev(keys[v], R, d_w, 5, 7)
        for _ in range(a.proofs):
            t0 = time.perf_counter(); ctx.prove_dev(keys[v], R, d_w, 5, 7); dev[v].append(1e3 * (time.perf_counter() - t0))
        for _ in range(a.proofs):
            t0 = time.perf_counter(); ctx.prove(keys[v], R, pin, 5, 7); host[v].append(1e3 * (time.perf_counter() - t0))
for v in values:
    print("option %d = %-6d prove_dev median %.3f min %.3f | zg_prove median %.3f min %.3f max %.3f" %
          (a.option, v, np.median(dev[v]), np.min(dev[v]), np.median(host[v]), np.min(host[v]), np.max(host[v])))
