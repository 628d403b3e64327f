"""Twiddles with their Shoup companions for tools/ubench_shoup.hip: w (canonical), ws = floor(w 2^261 / r), and
wm = w 2^261 mod r as 8 x u32 (the Montgomery-form operand the reference product takes) -> tools/ubench_shoup_tw.inc"""
import random
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
random.seed(7)
def limbs29(x): return [(x >> (29 * i)) & ((1 << 29) - 1) for i in range(9)]
def words32(x): return [(x >> (32 * i)) & 0xffffffff for i in range(8)]
ws_rows, wm_rows = [], []
for k in range(64):
    w = random.randrange(1, R) if k else R - 1
    ws = (w << 261) // R
    wm = (w << 261) % R
    ws_rows.append("{{%s}, {%s}}" % (", ".join("0x%08xu" % v for v in limbs29(w)), ", ".join("0x%08xu" % v for v in limbs29(ws))))
    wm_rows.append("{%s}" % ", ".join("0x%08xu" % v for v in words32(wm)))
open(__import__("os").path.join(__import__("os").path.dirname(__file__), "ubench_shoup_tw.inc"), "w").write(
    "static const Tw TWS[64] = {\n  %s};\nstatic const uint32_t TWM[64][8] = {\n  %s};\n" % (",\n  ".join(ws_rows), ",\n  ".join(wm_rows)))
