"""Child process of tests/test_model_gpu.py::test_two_rank_graphed_step_matches_manual_data_parallel: one data-parallel rank
driving GraphedTrainStep (hipGraph + bucketed, event-ordered gradient all-reduce + fused AGC/AdamW) on the real model.
Two ranks share this box's one GPU, hence the gloo backend there (RCCL refuses duplicate devices).  With WORLD_SIZE=1 and
backend 'nccl' the same chain runs over RCCL itself (1-rank communicator, exchange forced on): graph replay -> in-graph event
nodes -> communication stream -> RCCL collective -> fused optimizer.  The code path is the product's in both cases.
argv: out_path steps bucket_mb [backend [exchange [payload]]]"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import weights as OW                                            # noqa: E402  (seeded weights / inputs only)
from segmentation_factory_amd import SegmentationModel, criterion_lowres    # noqa: E402
from segmentation_factory_amd.graph import GraphedTrainStep                 # noqa: E402
from segmentation_factory_amd.optim import FusedAGCAdamW, param_groups_weight_decay   # noqa: E402


def main():
    out_path, steps, bucket_mb = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    backend = sys.argv[4] if len(sys.argv) > 4 else 'gloo'
    exchange = sys.argv[5] if len(sys.argv) > 5 else 'all_reduce'
    payload = sys.argv[6] if len(sys.argv) > 6 else 'fp32'
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group(backend, init_method='env://', rank=rank, world_size=world)
    backbone, head, nc, per_rank, H, W, seed = 'MiT-B0', 'SegFormerHead', 19, 2, 64, 64, 17
    sd = OW.make_state_dict(backbone, head, nc, seed + rank)        # DIFFERENT initial weights per rank: rank 0's must win (broadcast)
    x, y = OW.synthetic_batch(per_rank * world, H, W, nc, seed)
    x, y = x[rank * per_rank:(rank + 1) * per_rank].cuda(), y[rank * per_rank:(rank + 1) * per_rank].cuda()
    model = SegmentationModel(backbone, num_classes=nc, seg_head=head, compute_dtype=torch.float32)
    model.load_state_dict(sd)
    model = model.cuda().train()
    for mod in model.backbone.modules():
        if hasattr(mod, 'drop_prob'):
            mod.drop_prob = 0.0
    model.decode_head.dropout.p = 0.0
    opt = FusedAGCAdamW(param_groups_weight_decay(model, 0.025), lr=1e-3)

    def loss_fn(m, img, lbl):
        return criterion_lowres(m.forward_lowres(img), lbl, (H, W), None, num_classes=nc, dice=True, ignore_index=255)
    gs = GraphedTrainStep(model, opt, loss_fn, (x, y), clip_grad=0.02, clip_mode='agc', warmup=1, bucket_mb=bucket_mb,
                          exchange=exchange, payload=payload, force_exchange=True)
    losses = [gs.step(x, y).item() for _ in range(steps)]
    torch.cuda.synchronize()
    extra = {}
    if os.environ.get('DP_MODE') == 'evalsync':
        # collective C2 on the graph path: after training every rank holds its OWN BatchNorm statistics; engine.evaluate must hand
        # everyone rank 0's before its forwards (train_gpu.py:233-236 broadcast_buffers=True), then C4 / C5 sum the matrices
        from types import SimpleNamespace
        from segmentation_factory_amd.engine import evaluate
        mine = torch.cat([b.detach().float().reshape(-1) for b in model.buffers()])
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        extra['buffers_differed_before'] = bool((both[0] != both[-1]).any().item())
        args = SimpleNamespace(nb_classes=nc, ignore_label=255, hip_graph=True)
        confmat, metric = evaluate(args, model, [(x, y), (x.flip(3).contiguous(), y.flip(2).contiguous())], torch.device('cuda'), 1)
        after = torch.cat([b.detach().float().reshape(-1) for b in model.buffers()])
        both = [torch.empty_like(after) for _ in range(world)]
        dist.all_gather(both, after)
        extra.update(buffers_equal_after=bool(all(torch.equal(both[0], t) for t in both)), hist=metric.hist.cpu(), mat=confmat.mat.cpu(),
                     eval_graphs=len(model.__dict__.get('_graphed_eval', {}).get('graphs', {})))      # (same shape twice: eager, then captured)
    if rank == 0:
        torch.save({'losses': losses, 'state': {k: v.detach().cpu() for k, v in model.state_dict().items()},
                    'n_buckets': len(gs.buckets), 'events': sum(e is not None for e in gs.events), 'ranges': gs.ranges,
                    'backend': dist.get_backend(), 'exchanging': gs.exchanging, **extra}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
