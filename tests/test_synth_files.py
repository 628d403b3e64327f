"""The synthetic circuits as the files za leaves on disk (libza_amd_synth.so: zg_synth_pk_container_write,
zg_synth_binwitness_write): the proving.key container of prover/src/groth16/format.rs:223-252 and the evaluator's
.binwitness dump (compiler/src/tester/embeeded.rs:47-57) -- what libza2c's prove() takes at bench sizes.  CPU only."""
import numpy as np

from oracle import bn254 as bn
from oracle import groth16 as g


def test_container_and_binwitness_writers_against_the_oracle_reader(lib_built, tmp_path):
    from za_amd import synth
    sr = synth.SynthR1CS(9, 3, 0.5, 77)
    body = bytes(range(256)) * 3                      # opaque here: the container only carries it
    pk = tmp_path / "proving.key"
    sr.write_proving_key(pk, body)
    buf = pk.read_bytes()
    qeqs, ignore, off = g.pk_container_read(buf)
    assert ignore == [] and buf[off:] == body and len(qeqs) == sr.n_constraints
    assert buf[:len(g.pk_container_write(qeqs, [], body))] == g.pk_container_write(qeqs, [], body)   # byte for byte
    for k in range(3):
        rp, col, co = sr.row_ptr[k], sr.col[k], sr.coeff[k]
        for i in (0, 1, sr.n_constraints // 2, sr.n_constraints - 1):
            want = []
            for e in range(int(rp[i]), int(rp[i + 1])):
                c = bn.from_le32(bytes(co[32 * e:32 * e + 32]))
                want.append((int(col[e]), (-c) % bn.R if k == 2 else c))     # za's QEQ is a*b + c = 0
            assert qeqs[i][k] == want
    bw = tmp_path / "w.binwitness"
    sr.write_binwitness(bw)
    raw = bw.read_bytes()
    nv = sr.n_inputs + sr.n_aux
    assert len(raw) == 32 * (nv + 1)

    def w32(b):   # FS::write_256_w32 (fs.rs:116-126): 32-bit words from the least significant, each big-endian
        return sum(int.from_bytes(b[4 * j:4 * j + 4], "big") << (32 * j) for j in range(8))
    assert w32(raw[:32]) == nv
    w = np.frombuffer(bytes(sr.witness), dtype=np.uint8)
    for i in (0, 1, 5, nv - 1):
        assert w32(raw[32 * (i + 1):32 * (i + 2)]) == bn.from_le32(bytes(w[32 * i:32 * i + 32]))
    sr.close()
