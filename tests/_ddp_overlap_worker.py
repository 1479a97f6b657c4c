"""Worker of tests/test_round4_gpu.py::test_two_ranks_on_one_gpu_overlap_equals_flat_allreduce: launched by torch.distributed.run with
two ranks that share the one GPU of the box (gloo: RCCL needs one device per rank).  Each rank computes the gradient of ITS shard,
exchanges it three ways - flat all-reduce on the caller's stream, chunked on the side stream behind an eager backward, chunked
behind a hipGraph replay - and checks that the three results are bit-identical and equal on both ranks."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rna-mpnn_amd"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rnampnn.model.rnampnn import RNAMPNN, CapturedTrainStep
    from rnampnn.utils import synth
    torch.cuda.set_device(0)
    model = RNAMPNN(precision="bf16", num_res_neighbours=30, num_res_mpnn_layers=3, padding_len=64)
    sd = synth.closed_form_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to("cuda:0").train()
    c, m, y = (torch.from_numpy(x).cuda() for x in synth.synth_batch([24, 17, 30, 12], first_index=70 + 100 * rank))
    seed = 11 + rank
    model.loss_and_grad(y, c, m, seed=seed)
    own = model.flat_grad.clone()
    model.allreduce_gradients()                              # flat, caller's stream
    torch.cuda.synchronize()
    g_flat = model.flat_grad.clone()
    assert not torch.equal(g_flat, own), "the exchange changed nothing: both ranks computed the same gradient?"
    model.enable_allreduce_overlap(True)
    model.loss_and_grad(y, c, m, seed=seed)
    model.allreduce_gradients()                              # chunked, side stream, fresh events
    torch.cuda.synchronize()
    assert torch.equal(model.flat_grad, g_flat), float((model.flat_grad - g_flat).abs().max())
    cap = CapturedTrainStep(model, 4, 30)
    for _ in range(2):
        cap(y, c, m, seed=seed)
        model.allreduce_gradients()                          # chunked, side stream, NO events from the replay
        torch.cuda.synchronize()
        assert torch.equal(model.flat_grad, g_flat), float((model.flat_grad - g_flat).abs().max())
    both = [torch.empty_like(g_flat) for _ in range(world)]
    dist.all_gather(both, g_flat)
    assert all(torch.equal(b, g_flat) for b in both)
    dist.barrier()
    if rank == 0:
        print("DDP_OVERLAP_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
