"""world_size-2 gloo test of the one-process-per-GPU orchestration (za_amd/dist.py) on CPU.

The per-rank compute engine is backed by the oracle here (tests may use it; the product never does);
what is under test is the sharding, the H scatter, the partial-sum all-gather and the combine order.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bn254 as bn
from oracle import coracle as co
from oracle import groth16 as g
from tests.util import golden, r1cs_from_fixture


class OracleEngine:
    """Same interface as za_amd.dist.GpuEngine, computed with the Python / C oracle on CPU tensors."""

    def __init__(self, fx, rank, world, h_eval=False, c_fold=False):
        from za_amd.dist import shard_range
        self.device = torch.device("cpu")
        self.rank, self.world = rank, world
        self.r1 = r1cs_from_fixture(fx)
        self.w = [int(x, 16) for x in fx["witness"]]
        self.params, _ = g.params_read(bytes.fromhex(fx["params"]))
        self.m = 1 << self.r1.domain_log()
        self.n_h = len(self.params.h)
        self.h_chunk_entries = (self.n_h + world - 1) // world
        self.shard_range = shard_range
        _, _, _, self.ad, self.bi, self.bd = g.eval_constraints(self.r1, self.w)
        # H in evaluation form (za_amd.dist.GpuEngine's second mode): the transformed h bases, chunked over m
        self.h_eval = h_eval
        self.h_mask_extra = 0
        self._agreed = False
        # the rank that runs the transforms folds its C rows (za_amd.dist.GpuEngine.start_h ->
        # zg_prove_h_eval_enqueue_fold_dev): it scatters a b / Z, and its l partial carries the rows' multiexp
        self.c_fold = c_fold and rank == 0
        self.c_folded = False

    def agree_h_form(self, group=None):
        """As GpuEngine.agree_h_form: evaluation form only if every rank has the transformed bases."""
        if self._agreed:
            return
        t = torch.tensor([1 if self.h_eval else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        self.h_eval = bool(int(t.item()))
        if self.h_eval:
            self.h_chunk_entries = (self.m + self.world - 1) // self.world
            self.h_mask_extra = 0x20
            self.h_eval_bases = g.h_eval_bases_fft(self.params.h, self.r1.domain_log())
        self._agreed = True

    def start_h(self):
        self.c_folded = self.h_eval and self.c_fold       # (decided when the pipeline is queued, as the library does)

    def finish_h(self):
        if self.h_eval:
            e = g.h_eval_scalars_without_c(self.r1, self.w) if self.c_folded else g.h_eval_scalars(self.r1, self.w)
            n = self.world * self.h_chunk_entries
            e = e + [0] * (n - len(e))
            return torch.frombuffer(bytearray(b"".join(bn.le32(x) for x in e[:n])), dtype=torch.uint8)
        h = g.h_coefficients(self.r1, self.w) + [0]
        n = self.world * self.h_chunk_entries
        h = h + [0] * (max(n, self.m) - len(h))
        buf = b"".join(bn.le32(x) for x in h[:n])
        return torch.frombuffer(bytearray(buf), dtype=torch.uint8)

    def _jac(self, P, fb):
        if P is None:
            return b"\0" * (3 * fb)
        enc = bn.g1_to_mont_bytes(P) if fb == 32 else bn.g2_to_mont_bytes(P)
        one = bn.le32(bn.mont(1, bn.Q)) + (b"\0" * 32 if fb == 64 else b"")
        return enc + one

    def partials(self, h_shard, mask):
        p, ni = self.params, self.r1.num_inputs
        h_acc = None
        if mask & 1:
            hs = [bn.from_le32(bytes(h_shard[32 * i:32 * i + 32].tolist())) for i in range(h_shard.numel() // 32)]
            if mask & 0x20:
                assert self.h_eval
                lo, hi = self.shard_range(self.m, self.rank, self.world)
                h_acc = bn.msm_naive(self.h_eval_bases[lo:hi], hs[:max(hi - lo, 0)])
            else:
                assert not self.h_eval
                lo, hi = self.shard_range(self.n_h, self.rank, self.world)
                hi = min(hi, self.m - 1)
                h_acc = bn.msm_naive(p.h[lo:hi], hs[:max(hi - lo, 0)])
        if not (mask & 0x1e):
            buf = self._jac(h_acc, 32) + self._jac(None, 32) * 3 + self._jac(None, 64)
            return torch.frombuffer(bytearray(buf), dtype=torch.uint8)
        aux = self.w[ni:]
        lo, hi = self.shard_range(len(p.l), self.rank, self.world)
        l_acc = bn.msm_naive(p.l[lo:hi], aux[lo:hi])
        if self.c_folded:                                   # rank 0: + sum_j c_j C_j over ALL the rows
            c = g.eval_constraints(self.r1, self.w)[2]
            Cj = g.c_fold_bases_fft(p.h, self.r1.domain_log())
            l_acc = bn.pt_add(l_acc, bn.msm_naive(Cj[:len(c)], c))
        a_idx = list(range(ni)) + [ni + i for i, d in enumerate(self.ad) if d]
        b_idx = [i for i, d in enumerate(self.bi) if d] + [ni + i for i, d in enumerate(self.bd) if d]
        lo, hi = self.shard_range(len(p.a), self.rank, self.world)
        a_acc = bn.msm_naive(p.a[lo:hi], [self.w[i] for i in a_idx[lo:hi]])
        lo, hi = self.shard_range(len(p.b_g1), self.rank, self.world)
        b1 = bn.msm_naive(p.b_g1[lo:hi], [self.w[i] for i in b_idx[lo:hi]])
        b2 = bn.msm_naive(p.b_g2[lo:hi], [self.w[i] for i in b_idx[lo:hi]])
        buf = self._jac(h_acc, 32) + self._jac(l_acc, 32) + self._jac(a_acc, 32) + self._jac(b1, 32) + self._jac(b2, 64)
        assert len(buf) == 576
        return torch.frombuffer(bytearray(buf), dtype=torch.uint8)

    def combine(self, partials_all, world, r, s):
        pa = bytes(partials_all)
        acc = [None] * 5
        for k in range(world):
            q = pa[576 * k:576 * (k + 1)]
            pts = [None if q[64 + 96 * i:96 + 96 * i] == b"\0" * 32 else bn.g1_from_mont_bytes(q[96 * i:96 * i + 64])
                   for i in range(4)]
            pts.append(None if q[384 + 128:384 + 192] == b"\0" * 64 else bn.g2_from_mont_bytes(q[384:384 + 128]))
            acc = [bn.pt_add(x, y) for x, y in zip(acc, pts)]
        h_acc, l_acc, a_ans, b1_ans, b2_ans = acc
        vk = self.params.vk
        A = bn.pt_add(bn.pt_add(bn.pt_mul(vk.delta_g1, r), vk.alpha_g1), a_ans)
        B = bn.pt_add(bn.pt_add(bn.pt_mul(vk.delta_g2, s), vk.beta_g2), b2_ans)
        Cc = bn.pt_add(bn.pt_add(bn.pt_mul(vk.delta_g1, r * s % bn.R), bn.pt_mul(vk.alpha_g1, s)), bn.pt_mul(vk.beta_g1, r))
        Cc = bn.pt_add(Cc, bn.pt_mul(a_ans, s))
        Cc = bn.pt_add(Cc, bn.pt_mul(b1_ans, r))
        Cc = bn.pt_add(bn.pt_add(Cc, h_acc), l_acc)
        return g.proof_bytes((A, B, Cc))


def _worker(rank, world, port, out_path, form="coefficient"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from za_amd.dist import sharded_prove
        fx = golden("proof_chain7.json")
        # "evaluation": every rank has the transformed bases; "mixed": rank 1 has not -- all ranks must fall back
        has_eval = form in ("evaluation", "folded") or (form == "mixed" and rank == 0)
        eng = OracleEngine(fx, rank, world, h_eval=has_eval, c_fold=form == "folded")
        proof = sharded_prove(eng, int(fx["r"], 16), int(fx["s"], 16))
        assert eng.h_eval == (form in ("evaluation", "folded"))   # (the coefficient engine never had the bases; mixed ranks agree on "no")
        assert eng.c_folded == (form == "folded" and rank == 0)
        if rank == 0:
            with open(out_path, "wb") as f:
                f.write(proof)
        else:
            assert proof is None
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2])
def test_sharded_prove_over_gloo_matches_golden(tmp_path, world):
    out = str(tmp_path / "proof.bin")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    fx = golden("proof_chain7.json")
    assert open(out, "rb").read().hex() == fx["proof"]


@pytest.mark.parametrize("form", ["evaluation", "mixed", "folded"])
def test_sharded_prove_with_h_in_evaluation_form(tmp_path, form):
    """The orchestration's second mode: rank 0 scatters the coset evaluations e in ranges of ceil(m / world), the ranks
    multiply them into their slices of the transformed h bases (mask bit 0x20) -- same golden proof; and when one rank
    lacks those bases, every rank agrees on the coefficient form.  "folded": rank 0 runs four transforms -- it scatters
    a b / Z and its l partial carries the multiexp of the C rows' values (the other rank does what it did)."""
    out = str(tmp_path / "proof.bin")
    mp.spawn(_worker, args=(2, _free_port(), out, form), nprocs=2, join=True)
    assert open(out, "rb").read().hex() == golden("proof_chain7.json")["proof"]


def test_shard_ranges_partition_exactly():
    from za_amd.dist import shard_range
    for total in (0, 1, 7, 127, 128, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
