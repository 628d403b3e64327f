"""every assignment of (h, l, a, b_g1, b_g2) to the side streams 0-4, one process each (run BEFORE anything touches the GPU here)"""
import itertools, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
perms = list(itertools.permutations(range(5)))
if len(sys.argv) > 1:
    perms = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for p in perms:
    env = dict(os.environ, ZA_AMD_JSLOTS=",".join(map(str, p)))
    r = subprocess.run([sys.executable, os.path.join(here, "r4_exp31_child.py")], env=env, capture_output=True, text=True, timeout=120)
    line = [l for l in r.stdout.splitlines() if "median" in l]
    print(line[0] if line else "FAILED %s %s" % (p, r.stderr[-200:]), flush=True)
