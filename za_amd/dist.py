"""One-process-per-GPU sharding of create_proof.

The MSMs shard by base/scalar chunk: every rank keeps its chunk of the h, l, a, b_g1, b_g2 queries
resident in HBM.  The NTT pipeline stays on rank 0, whose H scalars are scattered in equal chunks
(the one bulk exchange, (m*32 B)/world per peer over xGMI).  Each rank computes five partial sums
(576 bytes: the witness MSMs while the NTT is still running, the h MSM after the scatter) which are
all-gathered; rank 0 adds them and applies r, s.  RCCL cannot add curve
points, so the "reduce" is gather + local point adds.

`engine` abstracts the per-rank compute so the same orchestration runs on GPUs (GpuEngine, RCCL)
and in the world_size-2 gloo tests on CPU (tests/test_dist_gloo.py supplies an oracle-backed engine).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    chunk = (total + world - 1) // world
    lo = min(total, rank * chunk)
    return lo, min(total, lo + chunk)


H_MASK, WITNESS_MASK = 0x01, 0x1e


def _merge_partials(h_part, w_part, out=None):
    """h sum from one buffer, l / a / b_g1 / b_g2 from the other (576-byte layout of zg_prove_partials_*)."""
    if out is None:
        out = w_part.clone()
    else:
        out.copy_(w_part)
    out[:96] = h_part[:96]
    return out


def sharded_prove(engine, r, s, group=None, timings=None, force_collectives=False):
    """Collective: every rank calls it; rank 0 returns the 256-byte proof, the others None.
    force_collectives: run the scatter, the all-gather and the form agreement's all-reduce even in a world of one rank
    (tests/test_gpu_rccl.py: RCCL's first contact with this code does not have to wait for an 8-GPU node).
    timings (optional dict) receives this rank's host wall times in ms: witness_msm, h_wait (rank 0: what is left
    of R1CS evaluation + 7 NTTs after the witness MSMs returned), scatter, h_msm, gather, combine.

    engine.h_chunk_entries          number of H scalars per rank (ceil(n_h / world))
    engine.start_h()                rank 0: queue R1CS evaluation + the 7 NTTs, do not wait
    engine.finish_h()               rank 0: wait; uint8 tensor [world * chunk * 32] (zero padded) on engine.device
    engine.partials(h_shard, mask)  uint8 tensor [576] on engine.device (masked-out sums = identity)
    engine.combine(partials, world, r, s) -> bytes  (rank 0)

    The witness MSMs (l, a, b_g1, b_g2) need no H, so every rank runs its shards of them while rank 0's NTT
    pipeline is in flight; only the h MSM waits for the scatter.
    """
    import time
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    collect = world > 1 or force_collectives
    if hasattr(engine, "agree_h_form"):
        # evaluation or coefficient form of H: the same on every rank (once per engine)
        if force_collectives:
            engine.agree_h_form(group, force=True)
        else:
            engine.agree_h_form(group)
    nbytes = engine.h_chunk_entries * 32
    tm = {}
    t0 = time.perf_counter()
    if rank == 0:
        engine.start_h()
    w_part = engine.partials(None, WITNESS_MASK)
    t1 = time.perf_counter()
    tm["witness_msm"] = 1e3 * (t1 - t0)
    mine = torch.empty(nbytes, dtype=torch.uint8, device=engine.device)
    if rank == 0:
        h_all = engine.finish_h()
        tm["h_wait"] = 1e3 * (time.perf_counter() - t1)
        tm["h_total"] = 1e3 * (time.perf_counter() - t0)
        assert h_all.numel() == world * nbytes
        scatter_list = list(h_all.view(world, nbytes).unbind(0))
    else:
        scatter_list = None
    # gloo (CPU tests, or a debugging run of several ranks on one GPU) moves bytes through host memory
    staged = collect and dist.get_backend(group) == "gloo" and mine.is_cuda
    t2 = time.perf_counter()
    if collect:
        if staged:
            host = torch.empty(nbytes, dtype=torch.uint8)
            dist.scatter(host, [t.cpu() for t in scatter_list] if rank == 0 else None, src=0, group=group)
            mine.copy_(host)
        else:
            dist.scatter(mine, scatter_list, src=0, group=group)
        # the library launches on its own HIP streams: make the received chunk visible to them
        if mine.is_cuda:
            torch.cuda.current_stream(mine.device).synchronize()
    else:
        mine = scatter_list[0]
    t3 = time.perf_counter()
    tm["scatter"] = 1e3 * (t3 - t2)
    # The five sums are finished on the HOST (a Horner pass over c - 1 bit planes per multiexp: DESIGN.md section 5), so
    # the 576 bytes start in host memory: merged there, then ONE non-blocking copy from a pinned buffer into the device
    # tensor the all-gather reads (RCCL) -- no synchronisation on this rank between its last multiexp and the collective;
    # gloo takes the host tensor as it is.
    h_part = engine.partials(mine, H_MASK | getattr(engine, "h_mask_extra", 0))
    t4 = time.perf_counter()
    tm["h_msm"] = 1e3 * (t4 - t3)
    if collect and dist.get_backend(group) != "gloo" and hasattr(engine, "partials_to_device"):
        part = engine.partials_to_device(_merge_partials(h_part.cpu(), w_part.cpu(), out=engine.partials_pinned()))
    else:
        part = _merge_partials(h_part, w_part)
        if staged:
            part = part.cpu()
    if collect:
        gathered = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(gathered, part, group=group)
    else:
        gathered = [part]
    t5 = time.perf_counter()
    tm["gather"] = 1e3 * (t5 - t4)
    if timings is not None:
        timings.update(tm)
    if rank != 0:
        return None
    allp = torch.cat(gathered).cpu().numpy()
    proof = engine.combine(allp, world, r, s)
    if timings is not None:
        timings["combine"] = 1e3 * (time.perf_counter() - t5)
    return proof


def za_amd_mask_h_eval():
    from . import ZG_MASK_H_EVAL
    return ZG_MASK_H_EVAL


class GpuEngine:
    """Per-rank compute on one MI355X through the C ABI phase entry points."""

    def __init__(self, ctx, params, r1cs, d_witness, world):
        self.ctx, self.params, self.r1cs, self.d_witness, self.world = ctx, params, r1cs, d_witness, world
        self.device = torch.device("cuda", ctx.device)
        m = 1 << r1cs.log_m
        lo, hi = params.shard_range(0)
        # chunking is defined over the h query length (>= m-1); every rank derives the same chunk
        self.n_h_total = self._total_h()
        self.h_chunk_entries = (self.n_h_total + world - 1) // world
        self.m = m
        self._h_all = None
        # H in evaluation form (six transforms on rank 0 instead of seven) when this rank's key has its slice of the
        # transformed h bases; agree_h_form() makes it the decision of ALL ranks
        ready, lg, lo_e, hi_e = params.h_eval_info()
        self.h_eval = bool(ready and lg == r1cs.log_m)
        self._agreed = False
        self._set_form()

    def _set_form(self):
        if self.h_eval:
            self.h_chunk_entries = (self.m + self.world - 1) // self.world
            self.h_mask_extra = za_amd_mask_h_eval()
        else:
            self.h_chunk_entries = (self.n_h_total + self.world - 1) // self.world
            self.h_mask_extra = 0

    def agree_h_form(self, group=None, force=False):
        if self._agreed:
            return
        if dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
            t = torch.tensor([1 if self.h_eval else 0], dtype=torch.int32,
                             device=self.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            self.h_eval = bool(int(t.item()))
            self._set_form()
        self._agreed = True

    def _total_h(self):
        # all ranks hold the same total; recover it from this rank's shard bounds is not enough, so the
        # caller stores it on the params object (synth_params / params_load know it)
        return self.params.total_h

    def start_h(self):
        n = self.world * self.h_chunk_entries
        need = max(n, self.m)
        if self._h_all is None or self._h_all.numel() != need * 32:
            self._h_all = torch.zeros(need * 32, dtype=torch.uint8, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()
        if self.h_eval:
            # (names the key: rank 0's shard may hold the bases of the folded C rows -- then four transforms instead of six,
            # and this rank's l partial carries the C term; nothing changes for the other ranks)
            self.c_folded = self.ctx.prove_h_eval_enqueue_fold_dev(self.params, self.r1cs, self.d_witness, self._h_all.data_ptr())
        else:
            self.ctx.prove_h_enqueue_dev(self.r1cs, self.d_witness, self._h_all.data_ptr())

    def finish_h(self):
        self.ctx.sync()
        return self._h_all[: self.world * self.h_chunk_entries * 32]

    def partials(self, h_shard, mask):
        """The five partial sums of this rank's shards (576 bytes) as a HOST tensor: the library finishes them on the host."""
        ptr = 0 if h_shard is None else h_shard.data_ptr()
        out = self.ctx.prove_partials_dev(self.params, self.r1cs, self.d_witness, ptr, mask)
        return torch.from_numpy(out)

    def partials_pinned(self):
        if getattr(self, "_pinned", None) is None:
            self._pinned = torch.empty(576, dtype=torch.uint8).pin_memory()
            self._part_dev = torch.empty(576, dtype=torch.uint8, device=self.device)
        return self._pinned

    def partials_to_device(self, pinned):
        self._part_dev.copy_(pinned, non_blocking=True)    # queued on torch's stream, in front of the collective
        return self._part_dev

    def combine(self, partials_all, world, r, s):
        return self.ctx.prove_combine(self.params, np.ascontiguousarray(partials_all), world, r, s)
