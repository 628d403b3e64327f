"""Child process of tests/test_gpu_rccl.py: RCCL (torch.distributed backend "nccl") initialised before anything else touches
the GPU, then the sharded proof flow of za_amd/dist.py with its collectives forced at world size 1 -- dist.scatter and
dist.all_gather on uint8 device tensors, the int32 all_reduce(MIN) of the form agreement, the device_id= path -- on the golden
2^10 circuit.  Prints RCCL_OK <proof hex> per H form."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    # one collective of each kind and dtype the flow uses, on its own, first: a failure names the call
    t = torch.arange(96, dtype=torch.uint8, device="cuda")
    out = torch.empty_like(t)
    dist.scatter(out, [t], src=0)
    assert bytes(out.cpu().numpy()) == bytes(range(96))
    g = [torch.empty_like(t)]
    dist.all_gather(g, t)
    assert bytes(g[0].cpu().numpy()) == bytes(range(96))
    f = torch.tensor([1], dtype=torch.int32, device="cuda")
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    assert int(f.item()) == 1
    mx = torch.tensor([2.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)        # bench.py's max-over-ranks
    dist.barrier()
    import za_amd
    from oracle import coracle as co
    from tests import golden_1k as k1
    from tests.util import golden
    from za_amd.dist import GpuEngine, sharded_prove
    fx = golden("proof_chain10.json")
    r1, w = k1.chain10_circuit(fx)
    csr = co.CsrR1CS.from_r1cs(r1)
    ctx = za_amd.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    R = ctx.r1cs_load(csr.n_inputs, csr.n_aux, csr.n_constraints, csr.row_ptr, csr.col, csr.coeff)
    pbytes = k1.chain10_params()
    r, s = int(fx["r"], 16), int(fx["s"], 16)
    d_w = ctx.upload(co.witness_bytes(w))
    for h_eval in (1, 0):
        ctx.set_option(za_amd.OPT_H_EVAL, h_eval)
        P, _ = ctx.params_load(pbytes, checked=True)
        P.prepare(R, d_w)
        eng = GpuEngine(ctx, P, R, d_w, 1)
        tm = {}
        proof = sharded_prove(eng, r, s, timings=tm, force_collectives=True)
        assert bytes(proof).hex() == fx["proof"], "sharded flow over RCCL (h_eval=%d) gives another proof" % h_eval
        assert eng.h_eval == bool(h_eval)
        print("RCCL_OK h_eval=%d scatter %.3f ms gather %.3f ms %s" % (h_eval, tm["scatter"], tm["gather"], bytes(proof).hex()[:16]),
              flush=True)
        P.close()
    print("RCCL_VERSION", torch.cuda.nccl.version(), flush=True)
    ctx.free(d_w)
    R.close()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
