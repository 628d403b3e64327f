"""assignments of (h, l, a, b_g1, b_g2) to the side streams 0-4, one process each:  r4_exp33.py "<sizes>" [perm ...]"""
import itertools, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
sizes = sys.argv[1].split()
perms = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or list(itertools.permutations(range(5)))
for p in perms:
    env = dict(os.environ, ZA_AMD_JSLOTS=",".join(map(str, p)))
    r = subprocess.run([sys.executable, os.path.join(here, "r4_exp33_child.py")] + sizes, env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if "median" in l]
    print(line[0] if line else "FAILED %s %s" % (p, r.stderr[-200:]), flush=True)
