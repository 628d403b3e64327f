"""create_proof on the GPU (zg_prove*) -- bit-exact proof bytes against the golden fixtures and the C
oracle, error behaviour mirroring bellman's SynthesisError cases, and the sharded phase API."""
import ctypes as C

import numpy as np
import pytest

import za_amd

from oracle import bn254 as bn
from oracle import coracle as co
from oracle import groth16 as g
from tests.util import download_params, from_mont_bytes, golden, r1cs_from_fixture, table_infos, table_kinds

pytestmark = pytest.mark.gpu


def _load_fixture(ctx, fx):
    r1 = r1cs_from_fixture(fx)
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    P, used = ctx.params_load(bytes.fromhex(fx["params"]), checked=True)
    assert used == len(fx["params"]) // 2
    w = co.witness_bytes([int(x, 16) for x in fx["witness"]])
    return R, P, w, csr


@pytest.mark.parametrize("name", ["proof_factor.json", "proof_chain7.json"])
def test_prove_matches_golden_bytes_and_json(ctx, name):
    import za_amd
    fx = golden(name)
    R, P, w, _ = _load_fixture(ctx, fx)
    assert [R.a_aux_total, R.b_input_total, R.b_aux_total] == [sum(fx["a_aux_density"]), sum(fx["b_input_density"]),
                                                                sum(fx["b_aux_density"])]
    proof = ctx.prove(P, R, w, int(fx["r"], 16), int(fx["s"], 16))
    assert proof.hex() == fx["proof"]
    wi = [int(x, 16) for x in fx["witness"]]
    assert za_amd.proof_json(proof, wi[1:fx["n_inputs"]]) == fx["proof_json"]
    a, b, c = R.eval(w)
    assert from_mont_bytes(a, bn.R) == [int(x, 16) for x in fx["a"]]
    assert from_mont_bytes(b, bn.R) == [int(x, 16) for x in fx["b"]]
    assert from_mont_bytes(c, bn.R) == [int(x, 16) for x in fx["c"]]


def test_prove_from_za_pk_container(ctx):
    """proving.key as `za setup` writes it (format.rs:223-252): skip the header, read the bellman body."""
    import za_amd
    fx = golden("proof_factor.json")
    pk = bytes.fromhex(fx["pk_container"])
    off = za_amd.pk_params_offset(pk)
    P, used = ctx.params_load(pk[off:])
    assert off + used == len(pk)
    R, _, w, _ = _load_fixture(ctx, fx)
    assert ctx.prove(P, R, w, 5, 7).hex() == fx["proof"]


def test_proof_verifies_under_the_pairing_equation(ctx):
    """The reference's acceptance criterion (prover.rs:291-305) applied to the GPU's proof with fresh r, s."""
    fx = golden("proof_chain7.json")
    R, P, w, _ = _load_fixture(ctx, fx)
    pb = ctx.prove(P, R, w, 0xDEADBEEF, 0xFEEDFACE)
    v = [int.from_bytes(pb[32 * i:32 * i + 32], "big") for i in range(8)]
    proof = ((bn.Fq(v[0]), bn.Fq(v[1])), (bn.Fq2(v[2], v[3]), bn.Fq2(v[4], v[5])), (bn.Fq(v[6]), bn.Fq(v[7])))
    params, _ = g.params_read(bytes.fromhex(fx["params"]))
    wi = [int(x, 16) for x in fx["witness"]]
    pub = wi[1:fx["n_inputs"]]
    assert g.verify_proof(params.vk, proof, pub)
    assert not g.verify_proof(params.vk, proof, [(pub[0] + 1) % bn.R] + pub[1:])


def test_random_blinding_still_verifies(ctx):
    fx = golden("proof_factor.json")
    R, P, w, _ = _load_fixture(ctx, fx)
    out = np.zeros(256, dtype=np.uint8)
    wa = np.frombuffer(w, dtype=np.uint8)
    ctx._check(ctx.lib.zg_prove_random(ctx.h, P.h, R.h, wa.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    pb = out.tobytes()
    assert pb.hex() != fx["proof"]
    v = [int.from_bytes(pb[32 * i:32 * i + 32], "big") for i in range(8)]
    proof = ((bn.Fq(v[0]), bn.Fq(v[1])), (bn.Fq2(v[2], v[3]), bn.Fq2(v[4], v[5])), (bn.Fq(v[6]), bn.Fq(v[7])))
    params, _ = g.params_read(bytes.fromhex(fx["params"]))
    assert g.verify_proof(params.vk, proof, [6])
    assert not g.verify_proof(params.vk, proof, [7])


@pytest.mark.parametrize("lg", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 17, 19])
def test_h_in_evaluation_form_gives_the_coefficient_forms_proof(ctx, lg):
    """zg_prove* takes H in evaluation form (six transforms; the h query's bases transformed once per key by a DFT over
    the group, heval.hip); ZG_OPT_H_EVAL = 0 keeps bellman's icoset_fft + coefficient-form multiexp.  Same group element,
    so the same bytes -- and the C oracle's (which follows bellman step by step).  Covers every pass structure: one tile
    (2^1 .. 2^10), two passes (2^11 .. 2^16), three (2^17 ..)."""
    import os
    import random
    from oracle.circuits import chain_r1cs, prescribed_h_r1cs
    from za_amd import synth
    if lg <= 9:
        if lg <= 6:   # every variable constrained whatever the size; H = prescribed random coefficients
            rng = random.Random(lg)
            r1, w = prescribed_h_r1cs(lg, [rng.randrange(bn.R) for _ in range((1 << lg) - 1)])
        else:
            r1, w = chain_r1cs(lg, n_inputs=2, seed=900 + lg, bit_fraction=0.5)
        csr = co.CsrR1CS.from_r1cs(r1)
        wb = co.witness_bytes(w)
        sr = None
    else:
        sr = synth.SynthR1CS(lg, 2, 0.5, 900 + lg)
        csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
        wb = sr.witness
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    assert R.log_m == lg
    P, pbytes = ctx.setup(R, 0x1f3a, 0x2b4c, 0x3d5e, 0x4f60, 0x5172)
    r, s = 0x1234567 + lg, 0x7654321
    got = ctx.prove(P, R, wb, r, s)            # evaluation form (the default)
    assert ctx.prove(P, R, wb, r, s) == got
    ctx.set_option(za_amd.OPT_H_EVAL, 0)
    try:
        P0, _ = ctx.params_load(pbytes, checked=False)
        assert ctx.prove(P0, R, wb, r, s) == got   # coefficient form
        P0.close()
    finally:
        ctx.set_option(za_amd.OPT_H_EVAL, 1)
    if lg <= 17:
        assert co.prove(download_params(ctx, P), csr, wb, r, s, threads=8) == got
    P.close(); R.close()
    if sr is not None:
        sr.close()


def test_zg_prove_packs_a_bit_heavy_witness_on_the_host(ctx, opts):
    """zg_prove turns a mostly-0/1 witness into the packed form on a few host threads before the upload (3 % of the
    bytes cross PCIe); a witness of field elements, or one with a dense stretch, is uploaded as it is.  Same proof as
    zg_prove_dev on the raw array and as the C oracle either way."""
    from za_amd import synth
    opts(za_amd.OPT_PACK_MIN_LOG_N, 16)    # (by default only witnesses of 2^18 values and more are packed)
    for lg, bits, expect_packed in ((18, 0.97, True), (18, 0.5, False), (17, 1.0, True)):
        sr = synth.SynthR1CS(lg, 3, bits, 0x77 + lg)
        R = sr.load(ctx)
        nv = sr.n_inputs + sr.n_aux
        P = synth.synth_params(ctx, (1 << lg) - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 31)
        csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
        want = co.prove(download_params(ctx, P), csr, sr.witness, 11, 13, threads=8)
        pin = ctx.host_alloc(32 * nv)
        pin[:] = sr.witness
        assert ctx.prove(P, R, pin, 11, 13) == want
        sent = ctx.last_timings()["upload_bytes"]
        assert (sent < 32 * nv / 2) == expect_packed, (lg, bits, sent, 32 * nv)
        d_w = ctx.upload(sr.witness)
        assert ctx.prove_dev(P, R, d_w, 11, 13) == want
        ctx.free(d_w)
        if expect_packed and bits < 1.0:
            # one stretch of 2^16 values made dense: the packing attempt is abandoned, the proof (of a now unsatisfied
            # system -- still a well-defined group element) equals the raw-upload one
            w2 = np.array(sr.witness, copy=True).reshape(-1, 32)
            rng = np.random.default_rng(5)
            lo = 3 << 16
            w2[lo:lo + (1 << 16), :31] = np.frombuffer(rng.bytes(31 << 16), dtype=np.uint8).reshape(-1, 31)
            w2[lo:lo + (1 << 16), 31] = 0
            pin[:] = w2.reshape(-1)
            got = ctx.prove(P, R, pin, 11, 13)
            assert ctx.last_timings()["upload_bytes"] == 32 * nv
            d_w = ctx.upload(w2.reshape(-1))
            assert ctx.prove_dev(P, R, d_w, 11, 13) == got
            ctx.free(d_w)
        ctx.host_free(pin)
        P.close(); R.close(); sr.close()


def test_witness_to_device_hands_over_like_zg_prove(ctx, opts):
    """zg_witness_to_device (the hand-over of zg_prove on its own, for sharded callers): the context's buffer holds the
    witness byte for byte whether it crossed PCIe packed or as it is, and the phase entry points prove from it."""
    from za_amd import synth
    opts(za_amd.OPT_PACK_MIN_LOG_N, 16)
    for lg, bits in ((17, 0.97), (17, 0.3)):
        sr = synth.SynthR1CS(lg, 2, bits, 0x31)
        R = sr.load(ctx)
        nv = sr.n_inputs + sr.n_aux
        P = synth.synth_params(ctx, (1 << lg) - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 5)
        pin = ctx.host_alloc(32 * nv)
        pin[:] = sr.witness
        want = ctx.prove(P, R, pin, 9, 10)
        d_w, sent = ctx.witness_to_device(pin)
        assert (sent < 32 * nv // 2) == (bits > 0.9), (bits, sent)
        assert bytes(ctx.d2h(d_w, 32 * nv)) == bytes(sr.witness)
        assert ctx.prove_dev(P, R, d_w, 9, 10) == want
        ctx.host_free(pin)
        P.close(); R.close(); sr.close()
    with pytest.raises(Exception):
        ctx.witness_to_device(np.zeros(0, dtype=np.uint8))


def test_back_to_back_handovers_do_not_share_a_pinned_stage(ctx, opts):
    """zg_witness_to_device is queued, not waited for: the packed form of the next witness must not be built in pinned
    bytes whose copy is still in flight.  Hand-overs of two different witnesses alternate without a synchronisation in
    between (two pinned stages, each reused only behind the event of its last copy); what the device buffer holds at the end,
    and the proof from it, are the last witness's."""
    from za_amd import synth
    opts(za_amd.OPT_PACK_MIN_LOG_N, 16)
    lg = 18
    sr = synth.SynthR1CS(lg, 2, 0.97, 0x51)
    R = sr.load(ctx)
    nv = sr.n_inputs + sr.n_aux
    P = synth.synth_params(ctx, (1 << lg) - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 9)
    w1 = np.array(sr.witness, copy=True)
    w2 = w1.copy().reshape(-1, 32)
    w2[5::7, 0] ^= 1           # another 0/1 pattern and ...
    w2[11::1013, 1:9] = 0x5a   # ... other field elements (still packable: well below 8 %)
    w2 = w2.reshape(-1)
    pins = [ctx.host_alloc(32 * nv), ctx.host_alloc(32 * nv)]
    pins[0][:] = w1
    pins[1][:] = w2
    want = [ctx.prove(P, R, pins[k], 3, 5) for k in (0, 1)]
    assert want[0] != want[1]
    for rounds in (2, 3, 7):
        for k in range(rounds):
            d_w, sent = ctx.witness_to_device(pins[k & 1])
            assert sent < 32 * nv // 2
        last = (rounds - 1) & 1
        assert bytes(ctx.d2h(d_w, 32 * nv)) == bytes(pins[last])
        assert ctx.prove_dev(P, R, d_w, 3, 5) == want[last]
    for p in pins:
        ctx.host_free(p)
    P.close(); R.close(); sr.close()


def test_copy_engine_and_packing_threads_share_the_witness(ctx, opts):
    """The hand-over of a pinned, bit-heavy witness: the copy engine takes the array as it is from the front, piece by
    piece, while the host threads pack it from the back, until the two meet (ZG_OPT_PACK_RAW_PERMILLE caps the copier's
    share; where they meet depends on the host's load, so only bounds are checked).  What the device buffer holds is the
    witness, byte for byte, whatever the split -- also when the packed part turns out not to pack after pieces are already
    on their way, and for a witness in pageable memory, which is packed whole; the proof is the C oracle's."""
    from za_amd import synth
    opts(za_amd.OPT_PACK_MIN_LOG_N, 16)
    lg = 18
    sr = synth.SynthR1CS(lg, 2, 0.97, 0x52)
    R = sr.load(ctx)
    nv = sr.n_inputs + sr.n_aux
    P = synth.synth_params(ctx, (1 << lg) - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 9)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    want = co.prove(download_params(ctx, P), csr, sr.witness, 3, 5, threads=8)
    pin = ctx.host_alloc(32 * nv)
    pin[:] = sr.witness
    base = None
    for permille in (0, 70, 200, 500, 900, 0):
        opts(za_amd.OPT_PACK_RAW_PERMILLE, permille)
        for _ in range(3):
            d_w, sent = ctx.witness_to_device(pin)
            assert bytes(ctx.d2h(d_w, 32 * nv)) == bytes(pin), permille
            cap = (nv * permille // 1000) & ~((1 << 14) - 1)      # the copier takes whole 2^14-value slices
            if permille == 0:
                assert base in (None, sent)
                base = sent
            else:
                assert base - cap // 4 - 36 * (cap // 8) <= sent <= 32 * cap + base, (permille, sent, cap, base)
        assert ctx.prove(P, R, pin, 3, 5) == want
    # pageable memory: packed whole (that copy would be staged by the thread that feeds the copy engine)
    opts(za_amd.OPT_PACK_RAW_PERMILLE, 500)
    d_w, sent = ctx.witness_to_device(np.array(sr.witness, copy=True))
    assert sent == base
    assert bytes(ctx.d2h(d_w, 32 * nv)) == bytes(sr.witness)
    # a dense stretch near the end, where the packing threads start: given up with pieces already queued, the rest
    # follows as it is
    w2 = np.array(sr.witness, copy=True).reshape(-1, 32)
    rng = np.random.default_rng(6)
    lo = 3 << 16
    w2[lo:lo + (1 << 16), :31] = np.frombuffer(rng.bytes(31 << 16), dtype=np.uint8).reshape(-1, 31)
    w2[lo:lo + (1 << 16), 31] = 0
    pin[:] = w2.reshape(-1)
    d_w, sent = ctx.witness_to_device(pin)
    assert sent == 32 * nv
    assert bytes(ctx.d2h(d_w, 32 * nv)) == bytes(pin)
    ctx.host_free(pin)
    P.close(); R.close(); sr.close()


@pytest.mark.parametrize("name", ["proof_factor.json", "proof_chain7.json"])
def test_combination_beside_the_h_multiexp_or_at_the_end(ctx, opts, name):
    """ZG_OPT_COMBINE_EARLY: A, B and C without its h term are computed by the thread that finishes the last witness
    multiexp (while h is still on the GPU), or everything at the end -- the golden proof bytes either way, also for r = s = 0
    (A = alpha + sum a: no scalar multiplication at all) and r = s = the largest scalar; an r beyond the group order is
    the same refusal in both orders (prover.rs:173 takes Fr values: nothing larger exists there)."""
    import za_amd
    fx = golden(name)
    R, P, w, csr = _load_fixture(ctx, fx)
    r0, s0 = int(fx["r"], 16), int(fx["s"], 16)
    pk = download_params(ctx, P)
    for early in (1, 0, 1):
        opts(za_amd.OPT_COMBINE_EARLY, early)
        assert ctx.prove(P, R, w, r0, s0).hex() == fx["proof"]
        d_w = ctx.upload(w)
        assert ctx.prove_dev(P, R, d_w, r0, s0).hex() == fx["proof"]
        ctx.free(d_w)
        for r, s in ((0, 0), (bn.R - 1, bn.R - 1), (1, 0)):
            assert ctx.prove(P, R, w, r, s) == co.prove(pk, csr, w, r, s), (early, r, s)
        for r, s in ((bn.R, 1), (1, bn.R), (2**256 - 1, 3)):
            with pytest.raises(za_amd.ZaAmdError) as e:
                ctx.prove(P, R, w, r, s)
            assert e.value.code == za_amd.ZG_E_BAD_FORMAT, (early, r, s)
    P.close(); R.close()


def test_zg_prove_falls_back_to_the_plain_upload_while_packing_is_slow(lib_built):
    """The packing pass competes with whatever else runs on the host: two passes in a row slower than their budget (here:
    a budget of zero) and the next proofs hand the witness over unpacked (ZG_OPT_PACK = 2; the default never looks at the
    clock).  Same proof bytes either way; a pass that allocates is not judged."""
    import za_amd
    from za_amd import synth
    ctx = za_amd.Context(0)
    ctx.set_option(za_amd.OPT_PACK, 2)                     # the packing pass judged against the plain upload's time ...
    ctx.set_option(za_amd.OPT_PACK_BUDGET_PERMILLE, 0)     # ... of which it may take nothing
    lg = 20
    sr = synth.SynthR1CS(lg, 2, 0.97, 0x99)
    R = sr.load(ctx)
    nv = sr.n_inputs + sr.n_aux
    P = synth.synth_params(ctx, (1 << lg) - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 77)
    pin = ctx.host_alloc(32 * nv)
    pin[:] = sr.witness
    proofs, sent = [], []
    for _ in range(6):
        proofs.append(bytes(ctx.prove(P, R, pin, 3, 4)))
        sent.append(int(ctx.last_timings()["upload_bytes"]))
    assert len(set(proofs)) == 1
    # (the first pass into each of the two pinned stages allocates and is not judged; the next two are over budget)
    assert all(b < 32 * nv // 2 for b in sent[:4]) and sent[4] == sent[5] == 32 * nv, sent
    d_w = ctx.upload(sr.witness)
    assert bytes(ctx.prove_dev(P, R, d_w, 3, 4)) == proofs[0]
    ctx.free(d_w); ctx.host_free(pin)
    P.close(); R.close(); sr.close(); ctx.close()


def test_one_shot_context_skips_the_per_key_precomputation(lib_built):
    """ZG_OPT_ONE_SHOT (za's `prove` command: read the key, prove once, exit): no evaluation-form h bases, no tables --
    bellman's own schedule -- and the same proof bytes as a context that builds them."""
    import za_amd
    from za_amd import synth
    lg = 15
    sr = synth.SynthR1CS(lg, 2, 0.9, 1515)
    c1 = za_amd.Context(0)
    c1.set_one_shot()
    R = sr.load(c1)
    sizes = ((1 << lg) - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    P = synth.synth_params(c1, *sizes, 77)
    a = c1.prove(P, R, sr.witness, 31, 32)
    assert [P.table_info(k)[0] for k in range(5)] == [0] * 5 and P.h_eval_info()[0] is False
    c2 = za_amd.Context(0)
    R2 = sr.load(c2)
    assert c2.prove(P, R2, sr.witness, 31, 32) == a            # the same key in an ordinary context: tables appear
    assert P.table_info(0)[0] == 1 and P.h_eval_info()[0] is True
    assert c1.prove(P, R, sr.witness, 31, 32) == a             # ... and the one-shot context uses what is there
    P.close(); R.close(); R2.close(); sr.close(); c1.close(); c2.close()


def test_witness_is_validated_like_a_typed_assignment(ctx):
    """bellman's `Fr` cannot hold a value >= r and its variable 0 is CS::one(): the raw 32-byte witness of zg_prove*
    is checked on the device (k_witness_classes) and refused with BadFormat instead of giving a wrong proof."""
    import za_amd
    fx = golden("proof_chain7.json")
    R, P, w, _ = _load_fixture(ctx, fx)
    r_, s_ = int(fx["r"], 16), int(fx["s"], 16)
    good = ctx.prove(P, R, w, r_, s_)
    assert good.hex() == fx["proof"]
    nv = len(w) // 32
    for pos in (1, nv // 2, nv - 1):
        for bad in (bn.R, bn.R + 1, 1 << 254, (1 << 256) - 1):
            wb = bytearray(w)
            wb[32 * pos:32 * pos + 32] = bad.to_bytes(32, "little")
            with pytest.raises(za_amd.ZaAmdError) as e:
                ctx.prove(P, R, bytes(wb), r_, s_)
            assert e.value.code == za_amd.ZG_E_BAD_FORMAT and ("value %d " % pos) in str(e.value), str(e.value)
            d_w = ctx.upload(bytes(wb))
            with pytest.raises(za_amd.ZaAmdError) as e:
                ctx.prove_dev(P, R, d_w, r_, s_)
            assert e.value.code == za_amd.ZG_E_BAD_FORMAT
            # the phase API of a rank that does not evaluate the R1CS
            with pytest.raises(za_amd.ZaAmdError) as e:
                ctx.prove_partials_dev(P, R, d_w, 0, mask=0x1e)
            assert e.value.code == za_amd.ZG_E_BAD_FORMAT
            ctx.free(d_w)
    # r - 1 is the largest canonical value: accepted (the proof is then simply one of an unsatisfied system)
    wb = bytearray(w)
    wb[32 * (nv - 1):32 * nv] = (bn.R - 1).to_bytes(32, "little")
    assert len(ctx.prove(P, R, bytes(wb), r_, s_)) == 256
    # variable 0 must be ONE
    for v0 in (0, 2, bn.R - 1):
        wb = bytearray(w)
        wb[0:32] = v0.to_bytes(32, "little")
        with pytest.raises(za_amd.ZaAmdError) as e:
            ctx.prove(P, R, bytes(wb), r_, s_)
        assert e.value.code == za_amd.ZG_E_BAD_FORMAT and "ONE" in str(e.value)
    # the packed form: a listed value >= r is caught by the same device pass
    classes, idx, vals = za_amd.pack_witness(w)
    if idx.size:
        vb = np.array(vals, copy=True)
        vb[0:32] = np.frombuffer(bn.R.to_bytes(32, "little"), dtype=np.uint8)
        with pytest.raises(za_amd.ZaAmdError) as e:
            ctx.prove_packed(P, R, (classes, idx, vb), r_, s_)
        assert e.value.code == za_amd.ZG_E_BAD_FORMAT
    # and a good witness still proves to the golden bytes afterwards (no sticky state)
    assert ctx.prove(P, R, w, r_, s_) == good


def test_error_behaviour(ctx):
    import za_amd
    fx = golden("proof_chain7.json")
    R, P, w, csr = _load_fixture(ctx, fx)
    # AssignmentMissing: no witness
    out = np.zeros(256, dtype=np.uint8)
    rs = np.zeros(32, dtype=np.uint8)
    rc = ctx.lib.zg_prove(ctx.h, P.h, R.h, None, rs.ctypes.data_as(C.c_void_p), rs.ctypes.data_as(C.c_void_p),
                          out.ctypes.data_as(C.c_void_p))
    assert rc == za_amd.ZG_E_ASSIGNMENT_MISSING
    # key / circuit mismatch: multiexp's query-size assertion
    fx2 = golden("proof_factor.json")
    R2, P2, w2, _ = _load_fixture(ctx, fx2)
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.prove(P2, R, w, 1, 1)
    assert e.value.code == za_amd.ZG_E_QUERY_SIZE
    # UnexpectedIdentity: delta_g1 = identity
    params, _ = g.params_read(bytes.fromhex(fx["params"]))
    params.vk.delta_g1 = None
    Pz, _ = ctx.params_load(g.params_write(params))
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.prove(Pz, R, w, 1, 1)
    assert e.value.code == za_amd.ZG_E_UNEXPECTED_IDENTITY
    # malformed key: IoError
    bad = bytearray(bytes.fromhex(fx["params"])); bad[0] |= 0x80
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.params_load(bytes(bad))
    assert e.value.code == za_amd.ZG_E_IO
    # PolynomialDegreeTooLarge: constraints + inputs need a domain of 2^28 (Fr::S = 28)
    rp = [np.zeros(1, dtype=np.uint64)] * 3
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.r1cs_load(2, 2, (1 << 27) + 1, rp, [np.zeros(1, np.uint32)] * 3, [np.zeros(32, np.uint8)] * 3)
    assert e.value.code == za_amd.ZG_E_POLYNOMIAL_DEGREE_TOO_LARGE


@pytest.mark.parametrize("lg,bitf", [(10, 0.9), (12, 0.0), (14, 0.97)])
def test_synthetic_circuits_against_c_oracle(ctx, lg, bitf):
    from za_amd import synth
    sr = synth.SynthR1CS(lg, 2, bitf, 1000 + lg)
    R = sr.load(ctx)
    m = 1 << lg
    P = synth.synth_params(ctx, m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 50 + lg)
    mp = download_params(ctx, P)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    r, s = 0x1111 + lg, 0x2222 + lg
    exp = co.prove(mp, csr, sr.witness, r, s, threads=16)
    got = ctx.prove(P, R, sr.witness, r, s)
    assert exp == got
    sr.close()


def test_sharded_phases_equal_the_unsharded_proof(ctx):
    """world = 3 shards driven sequentially on one GPU: h -> partials per shard -> combine."""
    from za_amd import synth
    lg = 12
    sr = synth.SynthR1CS(lg, 3, 0.8, 4242)
    R = sr.load(ctx)
    m = 1 << lg
    sizes = (m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    P1 = synth.synth_params(ctx, *sizes, 900)
    d_w = ctx.upload(sr.witness)
    r, s = 12345, 67890
    whole = ctx.prove_dev(P1, R, d_w, r, s)
    world = 3
    d_h = ctx.alloc(32 * (m + world))
    ctx.h2d(d_h, np.zeros(32 * (m + world), dtype=np.uint8))
    ctx.prove_h_dev(R, d_w, d_h)
    parts = []
    for rank in range(world):
        Pk = synth.synth_params(ctx, *sizes, 900, rank=rank, world=world)
        lo, hi = Pk.shard_range(0)
        assert Pk.total_h == m - 1
        parts.append(ctx.prove_partials_dev(Pk, R, d_w, d_h + 32 * lo))
    got = ctx.prove_combine(P1, np.concatenate(parts), world, r, s)
    assert got == whole
    sr.close()


def test_sharded_phases_in_evaluation_form(ctx):
    """The same with H in evaluation form: every shard holds its slice of the transformed h bases (cut from the full key
    by zg_params_shard, or made by zg_params_load(rank, world) from the whole bellman key) and takes its range of the
    scalars e; world = 3 (ranges that are no powers of two) and 8."""
    import za_amd
    from tests import golden_1k as k1
    from tests.util import golden
    from za_amd import synth
    lg = 13
    sr = synth.SynthR1CS(lg, 3, 0.8, 4243)
    R = sr.load(ctx)
    m = 1 << lg
    sizes = (m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    P1 = synth.synth_params(ctx, *sizes, 901)
    d_w = ctx.upload(sr.witness)
    r, s = 12345, 67890
    whole = ctx.prove_dev(P1, R, d_w, r, s)
    assert P1.h_eval_info()[:2] == (True, lg)
    d_e = ctx.alloc(32 * m)
    ctx.prove_h_eval_dev(R, d_w, d_e)
    for world in (3, 8):
        parts, covered = [], 0
        for rank in range(world):
            Pk = synth.params_shard(ctx, P1, rank, world)
            if rank == 1:
                Pk.prepare(R, d_w)       # tables up front, as bench.py does for its shard (the slice, not m points)
            ready, lgk, lo, hi = Pk.h_eval_info()
            assert ready and lgk == lg and lo == covered
            covered = hi
            parts.append(ctx.prove_partials_dev(Pk, R, d_w, d_e + 32 * lo, mask=0x1f | za_amd.ZG_MASK_H_EVAL))
            Pk.close()
        assert covered == m
        assert ctx.prove_combine(P1, np.concatenate(parts), world, r, s) == whole
    # a key without the transformed bases refuses the flag instead of mixing the forms
    Pc = synth.synth_params(ctx, *sizes, 901, rank=0, world=2)
    assert Pc.h_eval_info()[0] is False
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.prove_partials_dev(Pc, R, d_w, d_e, mask=0x01 | za_amd.ZG_MASK_H_EVAL)
    assert e.value.code == za_amd.ZG_E_INVALID_ARG
    ctx.free(d_e); ctx.free(d_w)
    sr.close()
    # zg_params_load(rank, world) on a bellman key (|h| = 2^10 - 1): the golden 2^10 proof from three sharded loads
    fx = golden("proof_chain10.json")
    r1, w = k1.chain10_circuit(fx)
    csr = co.CsrR1CS.from_r1cs(r1)
    R2 = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    pbytes = k1.chain10_params()
    d_w = ctx.upload(co.witness_bytes(w))
    d_e = ctx.alloc(32 << 10)
    ctx.prove_h_eval_dev(R2, d_w, d_e)
    parts = []
    for rank in range(3):
        Pk, _ = ctx.params_load(pbytes, checked=False, rank=rank, world=3)
        ready, lgk, lo, hi = Pk.h_eval_info()
        assert ready and lgk == 10
        parts.append(ctx.prove_partials_dev(Pk, R2, d_w, d_e + 32 * lo, mask=0x1f | za_amd.ZG_MASK_H_EVAL))
        Pk.close()
    Pf, _ = ctx.params_load(pbytes, checked=False)
    assert ctx.prove_combine(Pf, np.concatenate(parts), 3, int(fx["r"], 16), int(fx["s"], 16)).hex() == fx["proof"]
    ctx.free(d_e); ctx.free(d_w)


def test_full_size_proof_matches_discrete_log_expectation(ctx):
    """2^20 constraints (config 3 size): with every base's discrete log known, the proof's three
    elements are k*G for scalars computed independently (H from the C oracle)."""
    from za_amd import synth
    lg = 20
    sr = synth.SynthR1CS(lg, 2, 0.97, 0x5A)
    R = sr.load(ctx)
    m = 1 << lg
    n_a, n_b = sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total
    seed = 7000
    P = synth.synth_params(ctx, m - 1, sr.n_aux, n_a, n_b, seed)
    r, s = 0x1234567, 0x7654321
    got = ctx.prove(P, R, sr.witness, r, s)
    # independent expectation
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    a, b, c, ad, bi, bd = co.r1cs_eval(csr, sr.witness, threads=32)
    h = co.h_coeffs(a, b, c, sr.n_constraints + sr.n_inputs, lg, threads=32)
    wv = np.frombuffer(bytes(sr.witness), dtype="<u8").reshape(-1, 4)
    hv = np.frombuffer(bytes(h), dtype="<u8").reshape(-1, 4)

    def dot(vals, ks):
        tot = 0
        for row, k in zip(vals.tolist(), ks.tolist()):
            tot += (row[0] | (row[1] << 64) | (row[2] << 128) | (row[3] << 192)) * k
        return tot % bn.R
    ni = sr.n_inputs
    a_idx = np.concatenate([np.arange(ni), ni + np.nonzero(ad)[0]])
    b_idx = np.concatenate([np.nonzero(bi)[0], ni + np.nonzero(bd)[0]])
    assert len(a_idx) == n_a and len(b_idx) == n_b
    sh = dot(hv, synth.splitmix64_array(seed + 1, m - 1))
    sl = dot(wv[ni:], synth.splitmix64_array(seed + 2, sr.n_aux))
    sa = dot(wv[a_idx], synth.splitmix64_array(seed + 3, n_a))
    sb = dot(wv[b_idx], synth.splitmix64_array(seed + 4, n_b))
    alpha, beta, delta = 3, 5, 7
    eA = (alpha + sa + r * delta) % bn.R
    eB = (beta + sb + s * delta) % bn.R
    eC = (r * s * delta + s * alpha + r * beta + s * sa + r * sb + sh + sl) % bn.R
    A = co.g1_mul(bn.g1_to_mont_bytes(bn.G1_GEN), eA)
    B = co.g2_mul(bn.g2_to_mont_bytes(bn.G2_GEN), eB)
    Cc = co.g1_mul(bn.g1_to_mont_bytes(bn.G1_GEN), eC)
    pa, pb, pc = bn.g1_from_mont_bytes(A), bn.g2_from_mont_bytes(B), bn.g1_from_mont_bytes(Cc)
    assert got == g.proof_bytes((pa, pb, pc))
    sr.close()


def test_gpu_engine_through_torch_distributed_world1(ctx):
    """za_amd.dist.GpuEngine + sharded_prove on a real GPU (single-rank process group): exercises the
    torch-tensor plumbing (H buffer, partials tensor) that bench.py --gpus N uses on every rank."""
    import os
    import torch
    import torch.distributed as dist
    from za_amd import synth
    from za_amd.dist import GpuEngine, sharded_prove
    lg = 12
    sr = synth.SynthR1CS(lg, 2, 0.9, 777)
    R = sr.load(ctx)
    m = 1 << lg
    P = synth.synth_params(ctx, m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 31)
    d_w = ctx.upload(sr.witness)
    expect = ctx.prove_dev(P, R, d_w, 11, 22)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        torch.cuda.set_device(0)
        eng = GpuEngine(ctx, P, R, d_w, 1)
        got = sharded_prove(eng, 11, 22)
        assert got == expect
        # a second call reuses the cached H buffer
        assert sharded_prove(eng, 11, 22) == expect
    finally:
        if created:
            dist.destroy_process_group()
    sr.close()


@pytest.mark.parametrize("name", ["proof_factor.json", "proof_chain7.json"])
def test_prove_from_za_proving_key_and_binwitness(ctx, name):
    """a whole `proving.key` (constraints + ignore_signals + bellman Parameters,
    format.rs:223-293) and the evaluator's .binwitness dump give the same proof as the flattened inputs."""
    fx = golden(name)
    R, P = ctx.pk_load(bytes.fromhex(fx["pk_container"]))
    assert (R.n_inputs, R.n_aux, R.n_constraints) == (fx["n_inputs"], fx["n_aux"], len(fx["constraints"]))
    w = ctx.witness_from_binwitness(R, bytes.fromhex(fx["binwitness"]))
    assert bytes(w) == co.witness_bytes([int(x, 16) for x in fx["witness"]])
    proof = ctx.prove(P, R, w, int(fx["r"], 16), int(fx["s"], 16))
    assert proof.hex() == fx["proof"]


def test_pk_load_rejects_inconsistent_containers(ctx):
    import za_amd
    fx = golden("proof_chain7.json")
    pk = bytearray(bytes.fromhex(fx["pk_container"]))
    with pytest.raises(za_amd.ZaAmdError):
        ctx.pk_load(bytes(pk[:200]))
    R, P = ctx.pk_load(bytes(pk))
    bw = bytes.fromhex(fx["binwitness"])
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.witness_from_binwitness(R, bw[:-32])
    assert e.value.code == za_amd.ZG_E_BAD_FORMAT if hasattr(za_amd, "ZG_E_BAD_FORMAT") else True


def test_r1cs_load_host_passes_on_threads(ctx):
    """zg_r1cs_load's host passes (range check, densities by mention, index lists, long rows) run on threads from 2^16 entries
    (round 6): the first out-of-range column is reported in matrix, then entry order, whatever slice finds it first; the
    densities and the evaluation of a 2^17-row circuit equal numpy's."""
    import za_amd
    rng = np.random.default_rng(5)
    n_in, n_aux, nc = 3, 150_000, 1 << 17
    nv = n_in + n_aux
    row_ptr, col, coeff = [], [], []
    for k in range(3):
        per_row = rng.integers(0, 4, nc)
        per_row[7] = 300                                   # a long row (evaluated by its own kernel)
        rp = np.zeros(nc + 1, dtype=np.uint64)
        rp[1:] = np.cumsum(per_row)
        nnz = int(rp[-1])
        c = rng.integers(0, nv // 2 if k == 1 else nv, nnz).astype(np.uint32)   # (b mentions only half of the variables)
        v = np.zeros((nnz, 32), dtype=np.uint8)
        v[:, 0] = rng.integers(1, 5, nnz)
        row_ptr.append(rp); col.append(c); coeff.append(v.reshape(-1))
    assert all(len(c) > (1 << 16) for c in col)
    R = ctx.r1cs_load(n_in, n_aux, nc, row_ptr, col, coeff)
    a_aux = np.unique(col[0][col[0] >= n_in]).size
    b_in, b_aux = np.unique(col[1][col[1] < n_in]).size, np.unique(col[1][col[1] >= n_in]).size
    assert (R.a_aux_total, R.b_input_total, R.b_aux_total) == (a_aux, b_in, b_aux)
    w = np.zeros((nv, 32), dtype=np.uint8)
    w[:, 0] = rng.integers(0, 3, nv)
    w[0, 0] = 1
    abc = R.eval(w.reshape(-1))
    for k in range(3):
        prod = coeff[k].reshape(-1, 32)[:, 0].astype(np.int64) * w[col[k], 0]
        csum = np.concatenate(([0], np.cumsum(prod)))
        want = csum[row_ptr[k][1:].astype(np.int64)] - csum[row_ptr[k][:-1].astype(np.int64)]
        got = from_mont_bytes(abc[k], bn.R)[:nc]
        assert got == [int(x) for x in want], k
    # two offending columns: one deep inside matrix 2, one later in matrix 1 -- matrix 1's is the one reported
    bad = [c.copy() for c in col]
    bad[2][len(bad[2]) - 5] = nv + 9
    bad[1][70_000] = nv + 3
    bad[1][90_000] = nv + 4
    with pytest.raises(za_amd.ZaAmdError):
        ctx.r1cs_load(n_in, n_aux, nc, row_ptr, bad, coeff)
    assert ctx.lib.zg_last_error(ctx.h).decode() == "matrix 1: variable index %d out of range" % (nv + 3)


def test_pk_load_reports_the_key_before_the_constraints(ctx):
    """zg_pk_load reads the constraint sections on a helper thread while the device checks the queries (round 6): a container
    whose ignore_signals names a public signal is refused with the parser's message, and when the key is bad as well the
    key's error comes first -- as when the two steps ran one after the other."""
    import za_amd
    fx = golden("proof_chain7.json")
    pk = bytes.fromhex(fx["pk_container"])
    qeqs, ignore, off = g.pk_container_read(pk)
    assert 1 not in ignore
    bad_ignore = g.pk_container_write(qeqs, [1] + list(ignore), pk[off:])
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.pk_load(bad_ignore)
    assert "public signal 1 is in ignore_signals" in ctx.lib.zg_last_error(ctx.h).decode(), str(e.value)
    body = bytearray(pk[off:])
    first_h = 64 + 64 + 128 + 128 + 64 + 128 + 4 + 64 * (fx["n_inputs"]) + 4   # vk, |ic|, ic, |h|
    body[first_h + 40] ^= 1                                                       # a coordinate of the first h point
    both = g.pk_container_write(qeqs, [1] + list(ignore), bytes(body))
    with pytest.raises(za_amd.ZaAmdError):
        ctx.pk_load(both)
    assert ctx.lib.zg_last_error(ctx.h).decode().startswith("h: point is not on the curve")
    R, P = ctx.pk_load(pk)   # ... and the context still loads the good container
    assert R.n_constraints == len(fx["constraints"])


@pytest.mark.parametrize("name", ["proof_factor.json", "proof_chain7.json"])
def test_gpu_setup_reproduces_the_golden_key(ctx, name):
    """generate_parameters with the fixture's toxic waste gives the same
    Parameters::write bytes as the oracle, and proving with the resident key gives the golden proof."""
    fx = golden(name)
    r1 = r1cs_from_fixture(fx)
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    td = [int(x, 16) for x in fx["trapdoor"]]
    P, pbytes = ctx.setup(R, *td)
    assert pbytes.hex() == fx["params"]
    w = co.witness_bytes([int(x, 16) for x in fx["witness"]])
    assert ctx.prove(P, R, w, int(fx["r"], 16), int(fx["s"], 16)).hex() == fx["proof"]


def test_gpu_setup_then_prove_verifies_under_the_pairing(ctx):
    """setup -> prove -> verify round trip (the reference's prover test, prover.rs:225-306) at 2^11 rows,
    everything heavy on the GPU, verification by the oracle's pairing."""
    from oracle.circuits import chain_r1cs
    r1, w = chain_r1cs(11, n_inputs=3, seed=5, bit_fraction=0.7)
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    P, pbytes = ctx.setup(R, 0x1111, 0x2222, 0x3333, 0x4444, 0x5555)
    params, used = g.params_read(pbytes)
    assert used == len(pbytes)
    pb = ctx.prove(P, R, co.witness_bytes(w), 0xABCDEF, 0x123456)
    v = [int.from_bytes(pb[32 * i:32 * i + 32], "big") for i in range(8)]
    proof = ((bn.Fq(v[0]), bn.Fq(v[1])), (bn.Fq2(v[2], v[3]), bn.Fq2(v[4], v[5])), (bn.Fq(v[6]), bn.Fq(v[7])))
    pub = w[1:r1.num_inputs]
    assert g.verify_proof(params.vk, proof, pub)
    assert not g.verify_proof(params.vk, proof, [pub[0], (pub[1] + 1) % bn.R])
    # the oracle, proving with the GPU-made key, produces the same bytes
    mp = co.MontParams.from_params(params)
    assert co.prove(mp, csr, co.witness_bytes(w), 0xABCDEF, 0x123456, threads=8) == pb


def test_setup_rejects_unconstrained_variable(ctx):
    import za_amd
    # variable aux[1] appears in no constraint: bellman's generator returns UnconstrainedVariable
    r1 = g.R1CS(2, 3, [([(2, 1)], [(2, 1)], [(4, 1)])])
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.setup(R, 3, 5, 7, 11, 13)
    assert e.value.code == za_amd.ZG_E_UNCONSTRAINED_VARIABLE


def test_setup_prove_verify_entirely_through_the_abi(ctx):
    """za's setup -> prove -> verify cycle (prover.rs:225-306) through the C ABI only: GPU set-up and proof,
    host pairing check (zg_verify), wrong public input rejected."""
    import za_amd
    from za_amd import synth
    sr = synth.SynthR1CS(13, 4, 0.9, 2024)
    R = sr.load(ctx)
    P, pbytes = ctx.setup(R, 101, 202, 303, 404, 505)
    proof = ctx.prove(P, R, sr.witness, 0x77, 0x99)
    w = np.frombuffer(bytes(sr.witness), dtype=np.uint8).reshape(-1, 32)
    pub = [int.from_bytes(bytes(w[i]), "little") for i in range(1, sr.n_inputs)]
    assert za_amd.verify(pbytes, proof, pub) is True
    assert za_amd.verify(pbytes, proof, [pub[0] ^ 1] + pub[1:]) is False
    sr.close()


def test_ragged_circuit_shapes_against_the_python_oracle(ctx):
    """Edge shapes: empty linear combinations, a variable repeated inside one LC, a constant row (ONE * ONE = ONE),
    more inputs than constraints, an aux variable used only in C.  Key from the GPU set-up, proof vs the oracle."""
    R_ = bn.R
    w = [1, 5, 7, 3, 4, 12, 35, 6]          # ONE, two public inputs, five aux
    # var: 0 ONE, 1 x=5, 2 y=7 | 3 a=3, 4 b=4, 5 c=12, 6 d=35, 7 e=6
    cons = [
        ([(3, 1)], [(4, 1)], [(5, 1)]),                         # a*b = c
        ([(1, 1)], [(2, 1)], [(6, 1)]),                         # x*y = d
        ([(3, 1), (3, 1)], [(0, 1)], [(7, 1)]),                 # (a + a)*ONE = e   (duplicate term)
        ([], [(5, 1)], []),                                     # 0 * c = 0         (empty A and C)
        ([(0, 1)], [(0, 1)], [(0, 1)]),                         # ONE*ONE = ONE
        ([(7, 2)], [(0, 1)], [(5, 1)]),                         # 2e * ONE = c
    ]
    r1 = g.R1CS(3, 5, cons)
    assert r1.is_satisfied(w)
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    td = (0x11, 0x22, 0x33, 0x44, 0x55)
    P, pbytes = ctx.setup(R, *td)
    params, _ = g.generate_parameters(r1, *td)
    assert pbytes == g.params_write(params)
    exp = g.create_proof(r1, params, w, 9, 10)
    got = ctx.prove(P, R, co.witness_bytes(w), 9, 10)
    assert got == g.proof_bytes(exp)
    import za_amd
    assert za_amd.verify(pbytes, got, w[1:3])


def test_setup_rejects_a_key_that_would_not_match_the_density_maps(ctx):
    """A zero coefficient is still a *mention* for bellman's prover (density bit set) while the generator drops the
    identity point: the sharp edge of SURVEY section 8a.  zg_setup refuses such a circuit instead of emitting a key
    whose a/b queries are misaligned."""
    import za_amd
    r1 = g.R1CS(2, 2, [([(2, 1), (3, 0)], [(3, 1)], [(1, 1)])])   # aux 1 mentioned in A with coefficient 0
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.setup(R, 3, 5, 7, 11, 13)
    assert e.value.code == za_amd.ZG_E_BAD_FORMAT


def test_witness_of_all_zero_aux_and_of_all_ones(ctx):
    """Degenerate scalar sets: every MSM bucket empty except ONE's, and the all-ones bucket path."""
    from oracle.circuits import chain_r1cs
    for val in (0, 1):
        n_aux = 6
        cons = [([(1 + i, 1)], [(2 + i, 1)], [(1 + i, 1)]) for i in range(0, n_aux)]   # v*v' = v for v in {0,1}
        r1 = g.R1CS(1, n_aux + 1, cons)
        w = [1] + [val] * (n_aux + 1)
        assert r1.is_satisfied(w)
        csr = co.CsrR1CS.from_r1cs(r1)
        R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
        td = (5, 6, 7, 8, 9)
        P, pbytes = ctx.setup(R, *td)
        params, _ = g.generate_parameters(r1, *td)
        assert pbytes == g.params_write(params)
        exp = g.create_proof(r1, params, w, 3, 4)
        assert ctx.prove(P, R, co.witness_bytes(w), 3, 4) == g.proof_bytes(exp)


@pytest.mark.parametrize("lg", [12, 14, 15, 20])
def test_poseidon_chain_circuit_setup_prove_verify(ctx, lg):
    """BASELINE config 2's nominal size (2^14; the EdDSA-Poseidon statement itself is
    test_eddsa_poseidon_verifier_circuit_setup_prove_verify) and config 3 (circomlib Poseidon t=6 chain up to 2^20 rows: dense 6-term mix rows, no bit signals, so every
    scalar takes the full-width Pippenger path): GPU set-up, GPU proof equal to the C oracle's bytes with the same
    key, host pairing check on the public hash."""
    import za_amd
    from oracle.circuits import poseidon_chain_r1cs
    r1, w, hs = poseidon_chain_r1cs(lg)
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    P, pbytes = ctx.setup(R, 0xA1, 0xB2, 0xC3, 0xD4, 0xE5)
    wb = co.witness_bytes(w)
    got = ctx.prove(P, R, wb, 0x1234567, 0x7654321)
    mp = download_params(ctx, P)
    assert co.prove(mp, csr, wb, 0x1234567, 0x7654321, threads=32) == got
    assert za_amd.verify(pbytes, got, [hs[-1]]) is True
    assert za_amd.verify(pbytes, got, [hs[0]]) is (len(hs) == 1)


def test_fixed_base_window_table_gives_the_same_proof_as_the_plain_schedule(ctx, monkeypatch, opts):
    """The h multiexp through the per-key window table (one bucket set, digits folded) and through the plain
    per-window schedule (ZG_OPT_TABLES = 0) must agree byte for byte; both against the C oracle."""
    from za_amd import synth
    lg = 13
    sr = synth.SynthR1CS(lg, 2, 0.0, 77)      # a dense witness: every query qualifies for a window table
    R = sr.load(ctx)
    m = 1 << lg
    sizes = (m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    P_tab = synth.synth_params(ctx, *sizes, 31)
    d_w = ctx.upload(sr.witness)
    P_tab.prepare(R, d_w)          # h, l, a, b_g1 and b_g2 tables, sized from this witness
    kinds = table_kinds(P_tab)
    assert kinds == [1, 1, 1, 1, 1], kinds
    with_table = ctx.prove_dev(P_tab, R, d_w, 5, 6)
    ctx.free(d_w)
    opts(za_amd.OPT_TABLES, 0)
    P_plain = synth.synth_params(ctx, *sizes, 31)
    plain = ctx.prove(P_plain, R, sr.witness, 5, 6)
    assert with_table == plain
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    assert co.prove(download_params(ctx, P_plain), csr, sr.witness, 5, 6, threads=8) == plain
    sr.close()


def test_two_contexts_prove_concurrently_with_one_resident_key(ctx):
    """Two zg_ctx handles on two host threads share the R1CS and key handles (read-only after zg_params_prepare):
    every proof equals the single-threaded one."""
    import threading
    import za_amd
    from za_amd import synth
    lg = 14
    sr = synth.SynthR1CS(lg, 2, 0.9, 4321)
    R = sr.load(ctx)
    m = 1 << lg
    P = synth.synth_params(ctx, m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 61)
    d_w = ctx.upload(sr.witness)
    P.prepare(R, d_w)
    want = ctx.prove_dev(P, R, d_w, 11, 13)
    other = za_amd.Context(0)
    got = {0: [], 1: []}

    def work(i, c):
        for _ in range(4):
            got[i].append(c.prove_dev(P, R, d_w, 11, 13))
    th = [threading.Thread(target=work, args=(0, ctx)), threading.Thread(target=work, args=(1, other))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    other.close()
    ctx.free(d_w)
    assert all(bytes(p) == bytes(want) for p in got[0] + got[1]) and len(got[0]) == len(got[1]) == 4
    sr.close()


def test_r1cs_small_value_path_edge_coefficients(ctx):
    """The prove path's evaluator sums terms with a +-2^s coefficient (s < 64) on a 0/1 witness value as integers.
    A bit-packing circuit exercises every shift, both signs, the 2^64 and r - 2^64 coefficients that must take the
    generic path, rows mixing both kinds, a row whose fast terms sum to a negative number, and a long generic row
    (> 64 terms: the in-row reduction).  Key from the GPU set-up, proof against the C oracle."""
    R_ = bn.R
    rng = np.random.default_rng(99)
    nbits = 66
    bits = [int(b) for b in rng.integers(0, 2, nbits)]
    bits[0] = 1
    w = [1, 0]                      # ONE, public: packed value
    var_bit = []
    for b in bits:
        w.append(b); var_bit.append(len(w) - 1)
    cons = []

    def newvar(v):
        w.append(v % R_); return len(w) - 1
    # bit constraints b*(b-1) = 0  ->  b * b = b
    for v in var_bit:
        cons.append(([(v, 1)], [(v, 1)], [(v, 1)]))
    # packed = sum 2^s b_s for s < 66 (s = 64, 65 are generic coefficients)
    packed = sum(b << s for s, b in enumerate(bits))
    p_var = newvar(packed)
    cons.append(([(v, 1 << s) for s, v in enumerate(var_bit)], [(0, 1)], [(p_var, 1)]))
    # negative powers: neg = -sum 2^s b_s (s < 64) - 2^64 b_64
    neg = -sum(b << s for s, b in enumerate(bits[:65]))
    n_var = newvar(neg)
    cons.append(([(v, (R_ - (1 << s)) % R_) for s, v in enumerate(var_bit[:65])], [(0, 1)], [(n_var, 1)]))
    # mixed signs with a negative total on the fast path plus one generic term
    tot = bits[1] - (bits[2] << 63) - (bits[3] << 5) + 7 * packed
    m_var = newvar(tot)
    cons.append(([(var_bit[1], 1), (var_bit[2], R_ - (1 << 63)), (var_bit[3], R_ - 32), (p_var, 7)], [(0, 1)], [(m_var, 1)]))
    # a long generic row: 100 random coefficients on non-bit variables
    big = [newvar(int(rng.integers(2, 1 << 62)) * int(rng.integers(2, 1 << 62))) for _ in range(100)]
    coefs = [int(rng.integers(3, 1 << 62)) * 3 + 1 for _ in range(100)]
    l_var = newvar(sum(c * w[v] for c, v in zip(coefs, big)))
    cons.append(([(v, c) for c, v in zip(coefs, big)], [(0, 1)], [(l_var, 1)]))
    # the public output, then padding up to a 2^9 domain so the tiled evaluator is used
    w[1] = packed
    cons.append(([(p_var, 1)], [(0, 1)], [(1, 1)]))
    k = 2
    while len(cons) + 2 < 512:
        cons.append(([(k, 1)], [(0, 1)], [(k, 1)]))
        k = 2 + (k - 1) % (len(w) - 2)
    r1 = g.R1CS(2, len(w) - 2, cons)
    assert r1.is_satisfied(w)
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    assert R.log_m == 9
    P, pbytes = ctx.setup(R, 0x21, 0x32, 0x43, 0x54, 0x65)
    wb = co.witness_bytes(w)
    got = ctx.prove(P, R, wb, 77, 88)
    assert co.prove(download_params(ctx, P), csr, wb, 77, 88, threads=4) == got
    import za_amd
    assert za_amd.verify(pbytes, got, [packed]) is True


def test_table_budget_falls_back_to_the_plain_schedule(ctx, monkeypatch, opts):
    """ZG_OPT_TABLES_MAX_MB too small for any table: same proof through the per-window schedule."""
    from za_amd import synth
    lg = 12
    sr = synth.SynthR1CS(lg, 2, 0.0, 5)
    R = sr.load(ctx)
    m = 1 << lg
    sizes = (m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    P1 = synth.synth_params(ctx, *sizes, 17)
    a = ctx.prove(P1, R, sr.witness, 3, 4)
    opts(za_amd.OPT_TABLES_MAX_MB, 1); opts(za_amd.OPT_TABLES, 31)
    P2 = synth.synth_params(ctx, *sizes, 17)
    b = ctx.prove(P2, R, sr.witness, 3, 4)
    assert a == b
    sr.close()


@pytest.mark.parametrize("parts", ["2", "3", "4"])
def test_h_job_in_parts_gives_the_same_proof(ctx, monkeypatch, parts, opts):
    """The h multiexp split into dependent parts over consecutive base ranges (ZG_OPT_H_PARTS) against the single
    job, with and without the window table."""
    from za_amd import synth
    lg = 15
    sr = synth.SynthR1CS(lg, 2, 0.9, 808)
    R = sr.load(ctx)
    m = 1 << lg
    sizes = (m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    P = synth.synth_params(ctx, *sizes, 23)
    opts(za_amd.OPT_H_PARTS, 1)
    want = ctx.prove(P, R, sr.witness, 21, 22)
    opts(za_amd.OPT_H_PARTS, int(parts))
    assert ctx.prove(P, R, sr.witness, 21, 22) == want
    # the same with the last transform kernel launched range by range and every part gated on its own range
    opts(za_amd.OPT_H_PIPE, 1)
    assert ctx.prove(P, R, sr.witness, 21, 22) == want
    opts(za_amd.OPT_H_PIPE, 0)
    opts(za_amd.OPT_TABLES, 0)
    P2 = synth.synth_params(ctx, *sizes, 23)
    assert ctx.prove(P2, R, sr.witness, 21, 22) == want
    sr.close()


@pytest.mark.parametrize("qw", ["4", "8"])
def test_subset_sum_tables_with_opposite_and_equal_points(ctx, monkeypatch, opts, qw):
    """An all-0/1 witness takes the subset-sum tables for l and a (groups of four or of eight positions).  Rows
    (u - v + w) * 1 = w give the a and l queries pairs of opposite points in neighbouring positions (their subset sum is
    the identity), rows (x + y) * 1 = z pairs of equal points; proof against the C oracle, the plain schedule, and the
    pairing check."""
    import za_amd
    opts(za_amd.OPT_SMALL_LOG_N, 0)   # (jobs of this size otherwise take the small-job mode: window tables)
    opts(za_amd.OPT_SUBSET_W, int(qw))
    rng = np.random.default_rng(7)
    w = [1, 1]                       # ONE, one public input (a constant 1)
    cons = []
    for _ in range(260):
        b = int(rng.integers(0, 2)); c = int(rng.integers(0, 2))
        u = len(w); w.append(b)
        v = len(w); w.append(b)
        x = len(w); w.append(c)
        cons.append(([(u, 1), (v, bn.R - 1), (x, 1)], [(0, 1)], [(x, 1)]))       # opposite points for u, v
        p, q = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        while p + q == 2:
            p, q = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        y = len(w); w.append(p)
        z = len(w); w.append(q)
        s_ = len(w); w.append(p + q)                                                # 0 or 1
        cons.append(([(y, 1), (z, 1)], [(0, 1)], [(s_, 1)]))                        # equal points for y, z
    cons.append(([(1, 1)], [(0, 1)], [(1, 1)]))
    r1 = g.R1CS(2, len(w) - 2, cons)
    assert r1.is_satisfied(w) and r1.num_aux >= 1024
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    P, pbytes = ctx.setup(R, 0x91, 0x82, 0x73, 0x64, 0x55)
    wb = co.witness_bytes(w)
    d_w = ctx.upload(np.frombuffer(wb, dtype=np.uint8))
    P.prepare(R, d_w)
    assert table_kinds(P)[1:3] == [2, 2]        # subset-sum tables for l and a
    got = ctx.prove_dev(P, R, d_w, 1234, 5678)
    assert co.prove(download_params(ctx, P), csr, wb, 1234, 5678, threads=4) == got
    assert za_amd.verify(pbytes, got, [1]) is True
    opts(za_amd.OPT_TABLES, 0)
    P2, _ = ctx.setup(R, 0x91, 0x82, 0x73, 0x64, 0x55)
    assert ctx.prove_dev(P2, R, d_w, 1234, 5678) == got
    ctx.free(d_w)


def test_compact_window_table_over_the_field_element_positions_and_its_fallback(ctx, opts):
    """A bit-heavy query's subset-sum table carries a window table over the positions that held field elements in the
    witness it was built from (their digits then share one bucket set).  Same proof as the C oracle; a LATER witness with a
    field element at a position that was a bit takes the per-window path for that proof (k_classify counts such
    positions) -- again the oracle's bytes; and the same proofs with the compact rows switched off (ZG_OPT_COMPACT_TABLE = 0)."""
    import za_amd
    from za_amd import synth
    opts(za_amd.OPT_SMALL_LOG_N, 0)   # (queries of this size otherwise take the small-job mode: window tables)
    lg = 15
    sr = synth.SynthR1CS(lg, 2, 0.95, 4242)
    R = sr.load(ctx)
    m = 1 << lg
    sizes = (m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    w1 = np.array(sr.witness, copy=True)
    # a second witness: some aux values that are bits in w1 become field elements (the system is no longer satisfied; the
    # proof is still a well-defined group element, and the oracle computes the same one)
    w2 = w1.copy().reshape(-1, 32)
    rng = np.random.default_rng(9)
    changed = 0
    for v in range(sr.n_inputs + 100, sr.n_inputs + sr.n_aux, 997):
        if not w2[v, 1:].any() and w2[v, 0] <= 1:
            w2[v, :31] = np.frombuffer(rng.bytes(31), dtype=np.uint8)
            changed += 1
    assert changed >= 8
    w2 = w2.reshape(-1)
    proofs = {}
    for cw in ("1", "0"):
        opts(za_amd.OPT_COMPACT_TABLE, int(cw))
        P = synth.synth_params(ctx, *sizes, 61)
        d_w = ctx.upload(w1)
        P.prepare(R, d_w)
        kinds = table_infos(P)
        assert [k[0] for k in kinds[1:]] == [2, 2, 2, 2], kinds       # subset-sum tables for l, a, b_g1, b_g2
        proofs[cw] = (ctx.prove_dev(P, R, d_w, 77, 88), ctx.prove(P, R, w2, 77, 88), ctx.prove_dev(P, R, d_w, 77, 88),
                      [k[1] for k in kinds])
        ctx.free(d_w)
        if cw == "1":
            mp = download_params(ctx, P)
        P.close()
    assert proofs["1"][:3] == proofs["0"][:3]
    # the l query's table carries the compact rows (its 5 % of field elements are enough of them; the shorter a / b queries
    # may fall below the 1024 the rows are built from)
    assert proofs["1"][3][1] > proofs["0"][3][1], (proofs["1"][3], proofs["0"][3])
    assert proofs["1"][0] == proofs["1"][2] == co.prove(mp, csr, w1, 77, 88, threads=8)
    assert proofs["1"][1] == co.prove(mp, csr, w2, 77, 88, threads=8)
    R.close(); sr.close()


def test_h_coefficient_equal_to_one_single_constraint(ctx):
    """The dense (h) multiexp must not lose a scalar equal to 1 (round-1 defect: k_part sent it to the ones bin, which
    a dense job never reduces).  x * (2 ONE) = 2x with x = 3 has H = [1]: m = 2, one h base."""
    import za_amd
    r1 = g.R1CS(1, 1, [([(1, 1)], [(0, 2)], [(1, 2)])])
    w = [1, 3]
    assert g.h_coefficients(r1, w) == [1]
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    td = (0x15, 0x26, 0x37, 0x48, 0x59)
    P, pbytes = ctx.setup(R, *td)
    params, _ = g.generate_parameters(r1, *td)
    assert pbytes == g.params_write(params)
    exp = g.create_proof(r1, params, w, 21, 34)
    got = ctx.prove(P, R, co.witness_bytes(w), 21, 34)
    assert got == g.proof_bytes(exp)
    assert za_amd.verify(pbytes, got, []) is True


@pytest.mark.parametrize("lg,fbw,parts", [(5, None, None), (12, None, None), (12, "0", None), (13, None, "2"), (13, "0", "3")])
def test_dense_h_job_with_zero_one_and_extreme_coefficients(ctx, monkeypatch, lg, fbw, parts, opts):
    """H prescribed coefficient by coefficient (oracle/circuits.py::prescribed_h_r1cs): runs of 0 and 1, r - 1, 2,
    2^253-ish and random values, through the window-table schedule, the plain per-window schedule and the h job in
    parts -- proof bytes against the C oracle with the same (GPU-made) key."""
    import random
    import za_amd
    from oracle.circuits import prescribed_h_r1cs
    if fbw is not None:
        opts(za_amd.OPT_TABLES, int(fbw))
    if parts is not None:
        opts(za_amd.OPT_H_PARTS, int(parts))
    rng = random.Random(1000 + lg)
    m = 1 << lg
    pool = [0, 1, 1, 1, 2, bn.R - 1, bn.R - 2, (1 << 253) + 1, 1 << 20, (1 << 20) - 1]
    h = [rng.choice(pool) if rng.random() < 0.7 else rng.randrange(bn.R) for _ in range(m - 1)]
    h[0] = 1
    h[-1] = 1
    for k in range(8, min(40, m - 1)):
        h[k] = 1 if k < 24 else 0          # a run of ones and a run of zeros inside one sort tile
    r1, w = prescribed_h_r1cs(lg, h)
    csr = co.CsrR1CS.from_r1cs(r1)
    wb = co.witness_bytes(w)
    a, b, c, *_ = co.r1cs_eval(csr, wb, threads=4)
    hb = co.h_coeffs(a, b, c, m, lg, threads=4)
    assert [int.from_bytes(bytes(hb[32 * i:32 * i + 32]), "little") for i in range(m - 1)] == h
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    P, pbytes = ctx.setup(R, 0x1A, 0x2B, 0x3C, 0x4D, 0x5E)
    got = ctx.prove(P, R, wb, 0x777, 0x999)
    assert co.prove(download_params(ctx, P), csr, wb, 0x777, 0x999, threads=8) == got
    assert za_amd.verify(pbytes, got, []) is True



def test_one_key_many_witnesses_with_moving_statistics(ctx):
    """Consecutive proofs with ONE key whose witnesses have very different shares of 0 / 1 / other values: the scratch
    of a multiexp slot is re-used from proof to proof and the witness multiexps take the evaluator's class bytes, so
    every transition (bits -> field elements -> zeros -> ones -> mixed -> bits) must still give the oracle's bytes."""
    import random
    n_aux = (1 << 13) - 2
    cons = [([(1 + i, 1)], [(0, 1)], [(1 + i, 1)]) for i in range(n_aux)]      # x * ONE = x: any assignment satisfies
    r1 = g.R1CS(1, n_aux, cons)
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    P, _ = ctx.setup(R, 11, 12, 13, 14, 15, want_bytes=False)
    mp = download_params(ctx, P)
    rng = random.Random(77)
    bits = lambda p1: [1 if rng.random() < p1 else 0 for _ in range(n_aux)]
    field = lambda: [rng.randrange(bn.R) for _ in range(n_aux)]
    mixed = lambda p: [rng.randrange(bn.R) if rng.random() < p else rng.randrange(2) for _ in range(n_aux)]
    series = [bits(0.5), field(), [0] * n_aux, [1] * n_aux, mixed(0.03), bits(0.9), mixed(0.5), [0] * (n_aux - 1) + [5],
              bits(0.5), field(), field()]
    for k, aux in enumerate(series):
        w = co.witness_bytes([1] + aux)
        r, s = 100 + k, 200 + k
        assert ctx.prove(P, R, w, r, s) == co.prove(mp, csr, w, r, s, threads=8), "witness %d of the series" % k
    P.close(); R.close()


def test_sha256_merkle_circuit_setup_prove_verify(ctx):
    """BASELINE config 4 on a real circuit (za_amd/workloads.py::sha256_merkle, depth 2: 92 k bit-level constraints,
    long addition rows, > 99 % bit witness): constraint check on the GPU, GPU set-up, proof equal to the C oracle's
    bytes, pairing check with the hashlib root as the public input and rejection of another root."""
    import za_amd
    from za_amd import workloads
    ni, na, nc, rp, col, cf, wit, info = workloads.sha256_merkle(2, seed=3)
    R = ctx.r1cs_load(ni, na, nc, rp, col, cf)
    assert R.check(wit) is None
    P, pbytes = ctx.setup(R, 0xA1, 0xB2, 0xC3, 0xD4, 0xE5)
    got = ctx.prove(P, R, wit, 0x1234567, 0x7654321)
    csr = co.CsrR1CS(ni, na, nc, rp, col, cf)
    assert co.prove(download_params(ctx, P), csr, wit, 0x1234567, 0x7654321, threads=16) == got
    root = info["root"]
    pub = [int.from_bytes(root[:16], "big"), int.from_bytes(root[16:], "big")]
    assert za_amd.verify(pbytes, got, pub) is True
    assert za_amd.verify(pbytes, got, [pub[0], pub[1] ^ 1]) is False
    # queries of this size (below 2^19 positions) are latency-bound jobs: window tables, not subset sums
    assert table_kinds(P)[1:] == [1, 1, 1, 1]
    P.close(); R.close()


def test_long_rows_take_the_wave_per_row_evaluator(ctx, opts):
    """Rows with more than 32 terms (k_r1cs_eval_long: one wave per row) next to short ones, in all three matrices, with
    field-element coefficients and values (Montgomery path), +-2^s coefficients on bits (integer path), and lengths
    around the threshold and around the wave size."""
    import random
    rng = random.Random(4242)
    n_x = 300
    xs = [rng.randrange(bn.R) if i % 3 else rng.randrange(2) for i in range(n_x)]
    cons, aux = [], list(xs)
    var = lambda i: 1 + i                                  # variable 0 = ONE, aux from 1
    for i in range(n_x):                                   # every x is constrained: x * ONE = x
        cons.append(([(var(i), 1)], [(0, 1)], [(var(i), 1)]))
    for length in (31, 32, 33, 63, 64, 65, 129, 200):
        idx = rng.sample(range(n_x), length)
        coef = [rng.choice([1, 2, bn.R - 4, 1 << 20, rng.randrange(bn.R)]) for _ in idx]
        y = sum(c * xs[i] for c, i in zip(coef, idx)) % bn.R
        aux.append(y)
        yv = len(aux)                                      # its variable index
        lc = [(var(i), c) for c, i in zip(coef, idx)]
        cons.append((lc, [(0, 1)], [(yv, 1)]))             # long A
        cons.append(([(0, 1)], lc, [(yv, 1)]))             # long B
        cons.append(([(yv, 1)], [(0, 1)], lc))             # long C
    r1 = g.R1CS(1, len(aux), cons)
    w = [1] + aux
    assert r1.is_satisfied(w)
    csr = co.CsrR1CS.from_r1cs(r1)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    assert R.log_m >= 8
    wb = co.witness_bytes(w)
    assert R.check(wb) is None
    P, _ = ctx.setup(R, 21, 22, 23, 24, 25, want_bytes=False)
    exp = co.prove(download_params(ctx, P), csr, wb, 8, 9, threads=4)
    assert ctx.prove(P, R, wb, 8, 9) == exp
    P.close()
    # ... and with the C rows leaving the evaluator as scalars of the l multiexp (ZG_OPT_C_FOLD: the long C rows go through
    # the wave-per-row kernel's other exit)
    opts(za_amd.OPT_C_FOLD, 2)
    P, _ = ctx.setup(R, 21, 22, 23, 24, 25, want_bytes=False)
    assert ctx.prove(P, R, wb, 8, 9) == exp and ctx.last_stats()["c_folded"]
    P.close(); R.close()


def test_prove_packed_equals_prove(ctx):
    """zg_prove_packed (2-bit classes + the listed other values) against zg_prove on the same witnesses: a golden
    fixture, a 97 %-bit synthetic circuit, a witness without a single bit, and the malformed-input refusals."""
    import za_amd
    from za_amd import synth
    fx = golden("proof_chain7.json")
    R, P, w, _ = _load_fixture(ctx, fx)
    pk = za_amd.pack_witness(w)
    assert ctx.prove_packed(P, R, pk, int(fx["r"], 16), int(fx["s"], 16)).hex() == fx["proof"]
    for bitf, lg in ((0.97, 13), (0.0, 10), (1.0, 11)):
        sr = synth.SynthR1CS(lg, 3, bitf, 99 + lg)
        Rs = sr.load(ctx)
        m = 1 << lg
        Ps = synth.synth_params(ctx, m - 1, sr.n_aux, sr.n_inputs + Rs.a_aux_total, Rs.b_input_total + Rs.b_aux_total, 7)
        packed = za_amd.pack_witness(sr.witness)
        nv = sr.n_inputs + sr.n_aux
        assert packed[0].size == (nv + 3) // 4 and packed[2].size == 32 * packed[1].size
        assert ctx.prove_packed(Ps, Rs, packed, 3, 4) == ctx.prove(Ps, Rs, sr.witness, 3, 4)
        if bitf == 0.97:
            cls, idx, vals = [a.copy() for a in packed]
            bad = cls.copy(); bad[5] |= 0x0C                                  # class 3
            with pytest.raises(za_amd.ZaAmdError) as e:
                ctx.prove_packed(Ps, Rs, (bad, idx, vals), 3, 4)
            assert e.value.code == 20
            with pytest.raises(za_amd.ZaAmdError):                             # one listed value too few
                ctx.prove_packed(Ps, Rs, (cls, idx[:-1], vals[:-32]), 3, 4)
            sw = idx.copy(); sw[[0, 1]] = sw[[1, 0]]                           # not increasing
            with pytest.raises(za_amd.ZaAmdError):
                ctx.prove_packed(Ps, Rs, (cls, sw, vals), 3, 4)
            notone = cls.copy(); notone[0] &= 0xFC                             # variable 0 = 0
            with pytest.raises(za_amd.ZaAmdError):
                ctx.prove_packed(Ps, Rs, (notone, idx, vals), 3, 4)
        Ps.close(); Rs.close(); sr.close()


def test_eddsa_poseidon_verifier_circuit_setup_prove_verify(ctx):
    """BASELINE config 2 on its own statement (za_amd/workloads.py::eddsa_poseidon_verifier: 7.5 k constraints, witness =
    the reference's signature): GPU set-up, GPU proof equal to the C oracle's bytes, pairing check with (Ax, Ay, M) as
    public inputs and rejection of another message."""
    import za_amd
    from za_amd import workloads
    ni, na, cons, w = workloads.eddsa_poseidon_verifier()
    rp, cl, cf = workloads.csr_arrays(cons)
    R = ctx.r1cs_load(ni, na, len(cons), rp, cl, cf)
    wb = workloads.witness_bytes(w)
    assert R.check(wb) is None and R.log_m == 13
    P, pbytes = ctx.setup(R, 0xA1, 0xB2, 0xC3, 0xD4, 0xE5)
    got = ctx.prove(P, R, wb, 0x1234567, 0x7654321)
    csr = co.CsrR1CS(ni, na, len(cons), rp, cl, cf)
    assert co.prove(download_params(ctx, P), csr, wb, 0x1234567, 0x7654321, threads=8) == got
    kat = workloads.EDDSA_KAT
    assert za_amd.verify(pbytes, got, [kat["Ax"], kat["Ay"], kat["M"]]) is True
    assert za_amd.verify(pbytes, got, [kat["Ax"], kat["Ay"], kat["M"] + 1]) is False
    P.close(); R.close()


@pytest.mark.parametrize("lg,bitf,n_in", [(12, 0.9, 3), (13, 0.0, 2), (17, 0.0, 2), (17, 0.5, 5), (15, 0.97, 2)])
def test_witness_queries_share_digit_sorts(ctx, opts, lg, bitf, n_in):
    """l, a, b_g1 and b_g2 multiply the same scalars.  ZG_OPT_SHARE_SORT = 1 (default): b_g1 runs over the sorted (position,
    digit) pairs of b_g2 -- same positions, tables of one layout, whatever their kind (window, subset sums + compact rows,
    none).  = 2: with window tables of one plan ALL four read ONE sort over the witness positions through their position
    maps (msm.hip MsmShare) -- the a and b queries of the synthetic recipe have ragged densities (b about one half), so
    entries of variables outside a query's density are skipped.  = 0: a sort per query.  Every schedule gives the C oracle's
    bytes: small-job mode (2^12, 2^13), a dense witness beyond it (2^17), a half-bit witness, and a bit-heavy one (2^15 with
    the small-job mode off: subset sums)."""
    from za_amd import synth
    if bitf == 0.97:
        opts(za_amd.OPT_SMALL_LOG_N, 0)
    sr = synth.SynthR1CS(lg, n_in, bitf, 7700 + lg)
    R = sr.load(ctx)
    m = 1 << lg
    sizes = (m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    assert sizes[3] < sizes[1]            # the b queries do not have full density
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    r, s = 0x5151 + lg, 0x6262 + lg
    exp = None
    for share in (1, 2, 0):
        opts(za_amd.OPT_SHARE_SORT, share)
        P = synth.synth_params(ctx, *sizes, 90 + lg)      # (the tables' common layout is decided when they are built)
        if exp is None:
            exp = co.prove(download_params(ctx, P), csr, sr.witness, r, s, threads=16)
        for _ in range(2):
            assert ctx.prove(P, R, sr.witness, r, s) == exp
        st = ctx.last_stats()
        kinds = [P.table_info(k)[0] for k in range(5)]
        if share == 0:
            assert st["shared_sort_mask"] == 0 and not st["b_g1_behind_b_g2"]
        elif share == 1:
            # (not for queries in small-job mode -- latency-bound chains: up to 2^16 field elements among up to ~2^19 positions)
            assert st["b_g1_behind_b_g2"] == ((lg, bitf) in ((17, 0.0), (15, 0.97))), (st, kinds)
        elif bitf in (0.9, 0.0):
            # all four witness queries on one sort (a folded l query -- ZG_OPT_C_FOLD forced over the suite -- sorts its own)
            assert st["shared_sort_mask"] == (0b11100 if st["c_folded"] else 0b11110), (st, kinds)
        P.close()
    sr.close()


def test_shared_digit_sort_with_sharded_keys(ctx, opts):
    """A rank of a sharded proof sorts the variables ITS chunks of the queries span (a chunk of dense positions maps to a
    variable range through the index lists): three shards, each with the shared sort, against the unsharded proof and
    against the same shards with a sort per query."""
    from za_amd import synth
    lg = 13
    sr = synth.SynthR1CS(lg, 3, 0.0, 5151)
    R = sr.load(ctx)
    m = 1 << lg
    sizes = (m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    P1 = synth.synth_params(ctx, *sizes, 910)
    d_w = ctx.upload(sr.witness)
    r, s = 4321, 8765
    whole = ctx.prove_dev(P1, R, d_w, r, s)
    world = 3
    d_h = ctx.alloc(32 * (m + world))
    ctx.h2d(d_h, np.zeros(32 * (m + world), dtype=np.uint8))
    ctx.prove_h_dev(R, d_w, d_h)
    for share in (2, 1, 0):
        opts(za_amd.OPT_SHARE_SORT, share)
        parts = []
        for rank in range(world):
            Pk = synth.synth_params(ctx, *sizes, 910, rank=rank, world=world)
            lo, hi = Pk.shard_range(0)
            parts.append(ctx.prove_partials_dev(Pk, R, d_w, d_h + 32 * lo))
            st = ctx.last_stats()
            assert (st["shared_sort_mask"] != 0) == (share == 2) and not st["b_g1_behind_b_g2"]   # (2^13: small-job mode)
        assert ctx.prove_combine(P1, np.concatenate(parts), world, r, s) == whole
    sr.close()


@pytest.mark.parametrize("lg,bitf,n_in", [(10, 0.97, 2), (12, 0.9, 3), (13, 0.0, 2), (16, 0.97, 5), (19, 0.97, 2)])
def test_c_rows_folded_into_the_l_multiexp(ctx, opts, lg, bitf, n_in):
    """ZG_OPT_C_FOLD: the quotient (a b - c) / Z is linear in c, so the C term of the h multiexp is a multiexp of the C rows'
    VALUES (no transform) against a second transformed basis kept behind the l query: four transforms instead of six, one l
    query of n_aux + m points.  Mode 2 folds for every key, mode 1 where the rows evaluate to bits (decided with the key's
    first witness), mode 0 never -- the proof bytes are the C oracle's in every mode, from a host witness and from a device
    witness, for a witness that satisfies the circuit and for one that does not (bellman interpolates (a b - c) / Z on the
    coset whatever it is; so do both forms here)."""
    from za_amd import synth
    sr = synth.SynthR1CS(lg, n_in, bitf, 8800 + lg)
    R = sr.load(ctx)
    m = 1 << lg
    sizes = (m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    r, s = 0x7171 + lg, 0x8282 + lg
    bad = np.array(sr.witness, copy=True)
    bad[32 * (sr.n_inputs + 5):32 * (sr.n_inputs + 6)] = 0
    bad[32 * (sr.n_inputs + 5)] = 3                       # one aux value changed: rows that read it no longer hold
    exp = exp_bad = None
    opts(za_amd.OPT_C_FOLD_MIN_LOG_M, 10)
    for mode in (2, 1, 0):
        opts(za_amd.OPT_C_FOLD, mode)
        P = synth.synth_params(ctx, *sizes, 70 + lg)      # (decided per key, when its tables are built)
        if exp is None:
            pk = download_params(ctx, P)
            exp = co.prove(pk, csr, sr.witness, r, s, threads=16)
            exp_bad = co.prove(pk, csr, bad, r, s, threads=16)
            assert exp != exp_bad
        for _ in range(2):
            assert ctx.prove(P, R, sr.witness, r, s) == exp
        folded = ctx.last_stats()["c_folded"]
        if mode == 2:
            assert folded
        elif mode == 0:
            assert not folded
        elif bitf == 0.0:
            assert not folded                             # field elements everywhere: six transforms stay
        elif bitf == 0.97:
            assert folded                                 # few non-zero digits: XOR rows evaluate to 0 or 2 (90 % bits: borderline)
        d_w = ctx.upload(sr.witness)
        assert ctx.prove_dev(P, R, d_w, r, s) == exp
        assert ctx.last_stats()["c_folded"] == folded
        assert ctx.prove(P, R, bad, r, s) == exp_bad
        assert ctx.prove(P, R, sr.witness, r, s) == exp
        ctx.free(d_w)
        P.close()
    sr.close()


@pytest.mark.parametrize("lg,bitf,pack", [(20, 0.0, 1), (20, 0.97, 0), (21, 0.5, 1)])
def test_l_multiexp_in_parts_behind_a_witness_handed_over_in_chunks(ctx, opts, lg, bitf, pack):
    """ZG_OPT_UPLOAD_GATES: a witness of 2^20 values or more that goes over as it is (field elements; or bits with the
    packing switched off, which puts subset-sum tables under the range jobs) is copied in three chunks, each classified behind
    its copy, and the l multiexp runs in three parts gated on them.  Bytes against the C oracle's with the gates on and
    off, twice each (cached tables, the second hand-over), and zg_last_stats says which way the proof ran; a witness with a
    value >= r in the LAST chunk is still refused."""
    from za_amd import synth
    sr = synth.SynthR1CS(lg, 2, bitf, 7300 + lg)
    R = sr.load(ctx)
    m = 1 << lg
    P = synth.synth_params(ctx, m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 370 + lg)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    pk = download_params(ctx, P)
    r, s = 0x3131 + lg, 0x4242 + lg
    exp = co.prove(pk, csr, sr.witness, r, s, threads=16)
    opts(za_amd.OPT_PACK, pack)
    opts(za_amd.OPT_C_FOLD, 0)                      # (a folded l query waits for the evaluation anyway: not this path)
    pin = ctx.host_alloc(len(sr.witness))
    pin[:] = sr.witness
    for gates in (20, 0, 20):                       # (the default applies it from 2^23 values: A/B in profiles/r06_exp_upload_gates.txt)
        opts(za_amd.OPT_UPLOAD_GATES, gates)
        for src in (pin, sr.witness):               # pinned and pageable host memory
            assert ctx.prove(P, R, src, r, s) == exp
            assert ctx.last_stats()["l_in_parts_behind_the_upload"] == bool(gates)
    bad = np.array(sr.witness, copy=True)
    nv = sr.n_inputs + sr.n_aux
    bad[32 * (nv - 5):32 * (nv - 4)] = np.frombuffer((bn.R + 1).to_bytes(32, "little"), dtype=np.uint8)
    with pytest.raises(za_amd.ZaAmdError) as e:
        ctx.prove(P, R, bad, r, s)
    assert e.value.code == za_amd.ZG_E_BAD_FORMAT
    assert ctx.prove(P, R, pin, r, s) == exp        # ... and the context is fine afterwards
    d_w = ctx.upload(sr.witness)
    assert ctx.prove_dev(P, R, d_w, r, s) == exp and not ctx.last_stats()["l_in_parts_behind_the_upload"]
    ctx.free(d_w); ctx.host_free(pin)
    P.close(); R.close(); sr.close()


@pytest.mark.parametrize("lg,share", [(16, 0.25), (18, 1.0), (20, 0.25)])
def test_witness_with_one_constant_in_many_positions(ctx, lg, share):
    """a prover service takes witnesses it does not choose.  `share` of the aux values overwritten by
    ONE non-bit constant (65537): every such scalar is the same digits in the same buckets of all four witness queries (and an
    unsatisfied witness, which bellman proves all the same).  Bytes against the C oracle's, host and device witness, twice
    (cached tables)."""
    from za_amd import synth
    sr = synth.SynthR1CS(lg, 2, 0.97, 5150 + lg)
    R = sr.load(ctx)
    m = 1 << lg
    P = synth.synth_params(ctx, m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total, 270 + lg)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    nv = sr.n_inputs + sr.n_aux
    w = np.array(sr.witness, copy=True).reshape(nv, 32)
    ctx.prove(P, R, w.reshape(-1), 3, 4)                     # the key's tables are made for the circuit's own witness
    rng = np.random.default_rng(lg)
    idx = np.arange(sr.n_inputs, nv) if share >= 1.0 else sr.n_inputs + rng.choice(sr.n_aux, size=int(sr.n_aux * share), replace=False)
    w[idx, :] = 0
    w[idx, 0] = 1
    w[idx, 2] = 1                                            # 65537
    wb = w.reshape(-1)
    pk = download_params(ctx, P)
    r, s = 0x1717 + lg, 0x2828 + lg
    exp = co.prove(pk, csr, wb, r, s, threads=16)
    for _ in range(2):
        assert ctx.prove(P, R, wb, r, s) == exp
    d_w = ctx.upload(wb)
    assert ctx.prove_dev(P, R, d_w, r, s) == exp
    ctx.free(d_w)
    assert ctx.prove(P, R, sr.witness, 3, 4) == co.prove(pk, csr, sr.witness, 3, 4, threads=16)
    P.close(); R.close(); sr.close()


@pytest.mark.parametrize("lg", [12, 16, 20])
def test_the_fold_is_decided_per_proof_on_the_device(ctx, opts, lg):
    """VERDICT r5 / ADVICE r5: a key folds its C rows because the FIRST witness it saw had rows of bits; a later witness whose
    rows hold field elements must not send m dense scalars into the l multiexp.  With ZG_OPT_C_FOLD = 1 every proof's rows
    are counted by the evaluator and k_fold_decide takes four or six transforms for THAT proof on the device (the rows'
    values are struck from the l multiexp when it is six): a bits / dense / bits / dense sequence through one key, host and
    device witnesses and the phase API, against the C oracle's bytes, with zg_last_stats saying which form ran.  Mode 2
    keeps every proof folded."""
    from za_amd import synth
    sr = synth.SynthR1CS(lg, 2, 0.97, 4242 + lg)
    R = sr.load(ctx)
    m = 1 << lg
    sizes = (m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    csr = co.CsrR1CS(sr.n_inputs, sr.n_aux, sr.n_constraints, sr.row_ptr, sr.col, sr.coeff)
    nv = sr.n_inputs + sr.n_aux
    rng = np.random.default_rng(99 + lg)
    dense = rng.integers(0, 256, size=32 * nv, dtype=np.uint8)
    dense[31::32] = 0                                        # < 2^248: canonical
    dense[:32] = 0
    dense[0] = 1                                             # variable 0 is ONE
    bits = np.array(sr.witness, copy=True)
    r, s = 0x5151 + lg, 0x6262 + lg
    opts(za_amd.OPT_C_FOLD_MIN_LOG_M, 10)
    opts(za_amd.OPT_C_FOLD, 1)
    P = synth.synth_params(ctx, *sizes, 170 + lg)
    pk = download_params(ctx, P)
    exp = {"bits": co.prove(pk, csr, bits, r, s, threads=16), "dense": co.prove(pk, csr, dense, r, s, threads=16)}
    wit = {"bits": bits, "dense": dense}
    for name in ("bits", "dense", "bits", "dense", "dense", "bits"):
        assert ctx.prove(P, R, wit[name], r, s) == exp[name], name
        assert ctx.last_stats()["c_folded"] == (name == "bits"), name
    for name in ("dense", "bits"):
        d_w = ctx.upload(wit[name])
        assert ctx.prove_dev(P, R, d_w, r, s) == exp[name]
        assert ctx.last_stats()["c_folded"] == (name == "bits")
        # the phase API: the pipeline is queued with the rows armed; which form ran is known after the partials
        d_e = ctx.alloc(32 * m)
        armed = ctx.prove_h_eval_enqueue_fold_dev(P, R, d_w, d_e)
        assert armed
        w0 = ctx.prove_partials_dev(P, R, d_w, 0, mask=0x1e)
        assert ctx.last_stats()["c_folded"] == (name == "bits")
        ctx.sync()
        hp = ctx.prove_partials_dev(P, R, d_w, d_e, mask=0x01 | za_amd.ZG_MASK_H_EVAL)
        part = np.array(w0, copy=True)
        part[:96] = hp[:96]
        assert ctx.prove_combine(P, part, 1, r, s) == exp[name]
        ctx.free(d_e); ctx.free(d_w)
    opts(za_amd.OPT_C_FOLD, 2)                               # forced: no guard, the dense rows go through the l multiexp
    assert ctx.prove(P, R, dense, r, s) == exp["dense"] and ctx.last_stats()["c_folded"]
    opts(za_amd.OPT_C_FOLD, 0)
    assert ctx.prove(P, R, bits, r, s) == exp["bits"] and not ctx.last_stats()["c_folded"]
    P.close()
    sr.close()


@pytest.mark.parametrize("lg,bitf", [(13, 0.97), (12, 0.0)])
def test_sharded_phases_with_the_c_rows_folded_on_rank_0(ctx, opts, lg, bitf):
    """zg_prove_h_eval_enqueue_fold_dev: the rank that runs the transforms of a sharded proof drops C's two as well -- its
    shard of the key took the bases of the folded C rows when it was cut (or loaded), the scalars it scatters are a b / Z,
    and its l partial carries the rows' multiexp; the other ranks' calls are what they were.  Three shards and the unsharded
    key through the same entry point, in every mode, against the unsharded proof; then zg_params_load(rank, world) on a
    bellman key against the golden 2^10 proof."""
    from tests import golden_1k as k1
    from tests.util import golden
    from za_amd import synth
    sr = synth.SynthR1CS(lg, 3, bitf, 6100 + lg)
    R = sr.load(ctx)
    m = 1 << lg
    sizes = (m - 1, sr.n_aux, sr.n_inputs + R.a_aux_total, R.b_input_total + R.b_aux_total)
    d_w = ctx.upload(sr.witness)
    d_e = ctx.alloc(32 * m)
    r, s = 424242, 171717
    opts(za_amd.OPT_C_FOLD_MIN_LOG_M, 8)
    whole = None

    def sharded(keys, world, then_plain=False):
        folded = ctx.prove_h_eval_enqueue_fold_dev(keys[0], R, d_w, d_e)
        if then_plain:         # a pipeline queued the plain way behind it: nothing stays armed, d_e holds the six-transform scalars
            ctx.prove_h_eval_enqueue_dev(R, d_w, d_e)
            folded = False
        w0 = ctx.prove_partials_dev(keys[0], R, d_w, 0, mask=0x1e)      # rank 0's witness multiexps beside its transforms
        assert ctx.last_stats()["c_folded"] == folded
        ctx.sync()
        parts = []
        for rank, Pk in enumerate(keys):
            lo = Pk.h_eval_info()[2]
            hp = ctx.prove_partials_dev(Pk, R, d_w, d_e + 32 * lo, mask=0x01 | za_amd.ZG_MASK_H_EVAL)
            wp = w0 if rank == 0 else ctx.prove_partials_dev(Pk, R, d_w, 0, mask=0x1e)
            part = np.array(wp, copy=True)
            part[:96] = hp[:96]
            parts.append(part)
        return folded, np.concatenate(parts)

    for mode in (2, 1, 0):
        opts(za_amd.OPT_C_FOLD, mode)
        P1 = synth.synth_params(ctx, *sizes, 777)
        if whole is None:
            whole = ctx.prove_dev(P1, R, d_w, r, s)
        want_fold = mode == 2 or (mode == 1 and bitf == 0.97)
        for _ in range(2):
            folded, allp = sharded([P1], 1)                              # the unsharded key through the phase API
            assert folded == want_fold and ctx.prove_combine(P1, allp, 1, r, s) == whole
        keys = [synth.params_shard(ctx, P1, rank, 3) for rank in range(3)]
        for _ in range(2):
            folded, allp = sharded(keys, 3)
            assert folded == want_fold and ctx.prove_combine(P1, allp, 3, r, s) == whole
        folded, allp = sharded(keys, 3, then_plain=True)
        assert not folded and ctx.prove_combine(P1, allp, 3, r, s) == whole
        with pytest.raises(za_amd.ZaAmdError) as e:          # (argument checks as the plain entry's)
            ctx.prove_h_eval_enqueue_fold_dev(keys[0], R, 0, d_e)
        assert e.value.code == za_amd.ZG_E_ASSIGNMENT_MISSING
        for Pk in keys:
            Pk.close()
        P1.close()
    ctx.free(d_e); ctx.free(d_w)
    sr.close()
    if lg != 13:
        return
    opts(za_amd.OPT_C_FOLD, 2)
    fx = golden("proof_chain10.json")
    r1, w = k1.chain10_circuit(fx)
    csr = co.CsrR1CS.from_r1cs(r1)
    R2 = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    pbytes = k1.chain10_params()
    d_w = ctx.upload(co.witness_bytes(w))
    d_e = ctx.alloc(32 << 10)
    keys = [ctx.params_load(pbytes, checked=False, rank=rank, world=3)[0] for rank in range(3)]
    R, hold = R2, R
    folded, allp = sharded(keys, 3)
    Pf, _ = ctx.params_load(pbytes, checked=False)
    assert folded and ctx.prove_combine(Pf, allp, 3, int(fx["r"], 16), int(fx["s"], 16)).hex() == fx["proof"]
    ctx.free(d_e); ctx.free(d_w)
