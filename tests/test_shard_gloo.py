"""N > 1 host logic on CPU: world_size-2 gloo processes exercise the sharding helpers the
multi-GPU bench uses (whole RNAs per rank, global max_len for T_norm, job-level reductions) and
check, with the CPU oracle standing in for the kernels, that T_norm makes a sharded run
reproduce the single-rank logits."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lens, ret):
    import sys
    sys.path.insert(0, os.path.join(REPO, "rna-mpnn_amd"))
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from rnampnn.utils import shard, synth
    from rnampnn.model._schema import DEFAULT_HPARAMS, state_dict_shapes
    from oracle import rnampnn_oracle as O
    mine = shard.balanced_shards(lens, world)[rank]
    my_lens = [lens[i] for i in mine]
    coords = np.zeros((len(mine), max(my_lens), 7, 3), np.float32)
    mask = np.zeros((len(mine), max(my_lens)), np.float32)
    for row, i in enumerate(mine):
        coords[row, :lens[i]] = synth.synth_rna(lens[i], i)
        mask[row, :lens[i]] = 1
    t_norm = shard.global_max_len(mask.shape[1])
    hp = dict(DEFAULT_HPARAMS, num_res_neighbours=6, num_res_mpnn_layers=2, padding_len=64)
    sd = O.state_dict_from_numpy(synth.closed_form_state_dict(state_dict_shapes(hp)))
    cfg = O.OracleConfig(**{k: v for k, v in hp.items() if k in O.OracleConfig.__dataclass_fields__})
    # the oracle pads to T_norm explicitly (what the kernels do in closed form)
    cpad = np.zeros((len(mine), t_norm, 7, 3), np.float32); cpad[:, :coords.shape[1]] = coords
    mpad = np.zeros((len(mine), t_norm), np.float32); mpad[:, :mask.shape[1]] = mask
    logits, _ = O.forward(torch.from_numpy(cpad), torch.from_numpy(mpad), sd, cfg)
    el, units = shard.reduce_job(1.0 + rank, float(sum(my_lens)))
    correct = torch.tensor([1.0 * n for n in my_lens]) * 0.5
    micro, macro = shard.reduce_recovery(correct, torch.tensor([float(n) for n in my_lens]))
    ret[rank] = dict(mine=mine, t_norm=t_norm, elapsed=el, units=units, micro=micro, macro=macro,
                     logits={i: logits[row, :lens[i]].numpy() for row, i in enumerate(mine)})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_reproduces_single_rank():
    import sys
    sys.path.insert(0, os.path.join(REPO, "rna-mpnn_amd"))
    from rnampnn.utils import shard, synth
    from rnampnn.model._schema import DEFAULT_HPARAMS, state_dict_shapes
    from oracle import rnampnn_oracle as O
    lens = [31, 12, 27, 40, 9, 22]
    world = 2
    shards = shard.balanced_shards(lens, world)
    assert sorted(sum(shards, [])) == list(range(len(lens)))               # whole RNAs, each exactly once
    loads = [sum(lens[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(lens)
    assert shard.strided_shard(7, 1, 3) == [1, 4]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), lens, ret), nprocs=world, join=True)
    assert ret[0]["t_norm"] == ret[1]["t_norm"] == 40
    assert ret[0]["elapsed"] == ret[1]["elapsed"] == 2.0                   # MAX over ranks
    assert ret[0]["units"] == float(sum(lens))                             # SUM over ranks
    assert abs(ret[0]["micro"] - 0.5) < 1e-12 and abs(ret[1]["macro"] - 0.5) < 1e-12
    # single-rank reference on the whole padded batch
    coords, mask, _ = synth.synth_batch(lens, first_index=0)
    hp = dict(DEFAULT_HPARAMS, num_res_neighbours=6, num_res_mpnn_layers=2, padding_len=64)
    sd = O.state_dict_from_numpy(synth.closed_form_state_dict(state_dict_shapes(hp)))
    cfg = O.OracleConfig(**{k: v for k, v in hp.items() if k in O.OracleConfig.__dataclass_fields__})
    full, _ = O.forward(torch.from_numpy(coords), torch.from_numpy(mask), sd, cfg)
    for r in range(world):
        for i, lg in ret[r]["logits"].items():
            assert np.abs(lg - full[i, :lens[i]].numpy()).max() < 1e-5


def _grad_worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.join(REPO, "rna-mpnn_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rnampnn.model.rnampnn import RNAMPNN
    m = RNAMPNN(precision="f32", num_res_mpnn_layers=1)
    n = sum(p.numel() for p in m.parameters())
    m.flat_grad = torch.full((n,), float(rank + 1))            # stands in for the HIP backward's flat buffer
    off = 0
    for p in m.parameters():                                   # grads are views of the flat buffer
        p.grad = m.flat_grad[off:off + p.numel()].view(p.shape)
        off += p.numel()
    m.allreduce_gradients()
    ret[rank] = (float(m.flat_grad.min()), float(m.flat_grad.max()), float(next(m.parameters()).grad.mean()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks():
    """One flat all-reduce averages every parameter gradient (views of the flat buffer follow)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_grad_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    for r in (0, 1):
        assert ret[r] == (1.5, 1.5, 1.5)


def _chunk_worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.join(REPO, "rna-mpnn_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rnampnn.utils import shard
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(10007, generator=g)
    whole = flat.clone()
    dist.all_reduce(whole); whole /= world
    chunks = [(7000, 10007), (3000, 7000), (0, 3000)]          # backward order: tail of the model first (RNAMPNN.grad_chunks)
    shard.allreduce_mean_chunks(flat, chunks)
    bad = None
    try:
        shard.allreduce_mean_chunks(flat.clone(), [(7000, 10007), (0, 3000)])
    except ValueError as exc:
        bad = str(exc)
    ret[rank] = (bool(torch.equal(flat, whole)), bad)
    dist.barrier()
    dist.destroy_process_group()


def test_chunked_gradient_allreduce_equals_one_flat_allreduce():
    """The three chunk collectives of the overlapped exchange (bench.py train leg, rnampnn.utils.train.Trainer) give exactly the
    average one flat all-reduce gives; ranges that do not tile the buffer are refused."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_chunk_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    for r in (0, 1):
        assert ret[r][0] is True and ret[r][1] and "tile" in ret[r][1]
