#!/usr/bin/env python3
"""train_gpu.py -- the reference's training CLI (train_gpu.py:33-375) on the MI355X path.

Same flags, same epoch loop (train_one_epoch -> evaluate -> best-mIoU checkpoint), same checkpoint dict keys
(model_state / optimizer_state / scheduler_state / best_mIoU / F1_Score / Acc / scaler), same results / args / model
text files, same auto-resume from the first *.pth in --save_weights_dir.  Differences, all deliberate:

  * the model, loss, metrics and optimizer come from segmentation_factory_amd (HIP kernels); the optimizer is the fused
    AGC + AdamW kernel (timm create_optimizer semantics, "parity unpinned" -- DESIGN.md);
  * --nb_classes / --backbone / --heads accept a superset of the reference's choices (the
    reference CLI cannot express the ADE20K-150 / ConvNeXt configurations of BASELINE.json);
  * data: the reference's `datasets` package (PIL / torchvision pipelines, out of scope here) is used unchanged when it is
    importable (put the reference checkout on PYTHONPATH); `--dataset synthetic` runs on generated tensors with the same
    tensor contract (fp32 [3,S,S] image, int64 [S,S] label, ignore 255);
  * the train step and the eval forward are replayed as hipGraphs BY DEFAULT (segmentation_factory_amd/graph.py; --no-hip-graph
    for per-kernel launches, --hip-graph to make a capture failure an error instead of a fallback); data parallelism is then a
    bucketed RCCL exchange of the flat gradient buffer released by in-graph events (--grad-exchange, --grad-payload) instead of
    DistributedDataParallel hooks; evaluate() runs in fp32 as the reference does (--eval-dtype bf16 for the production type);
  * --finetune defaults to '' (the reference's default path makes the run fail unless that file exists, quirk Q10).

Launch: `python train_gpu.py ...` or `python -m torch.distributed.run --nproc-per-node N train_gpu.py ...` (RCCL).
"""
import argparse
import datetime
import json
import os
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, DistributedSampler, RandomSampler

from segmentation_factory_amd import SegmentationModel, evaluate, train_one_epoch, utils
from segmentation_factory_amd.build_models import backbone_registry, head_dict
from segmentation_factory_amd.optim import NativeScaler, create_optimizer, FusedAGCAdamW
from segmentation_factory_amd.scheduler import create_scheduler


def get_args_parser():
    parser = argparse.ArgumentParser('Segmentation Models training and evaluation script', add_help=False)
    # Dataset parameters (train_gpu.py:37-66)
    parser.add_argument("--data_root", type=str, default='/mnt/d/CityScapesDataset', help="path to CityScapes Dataset")
    parser.add_argument("--dataset", type=str, default='cityscapes',
                        choices=['cityscapes', 'voc', 'cocostuff', 'ade', 'kvasir', 'synapse', 'synthetic'])
    parser.add_argument("--image_size", type=int, default=1024, help="input size")
    parser.add_argument("--ignore_label", type=int, default=255, help="the dataset ignore_label")
    parser.add_argument("--ignore_index", type=int, default=255, help="the dataset ignore_index")
    parser.add_argument("--dice", type=bool, default=True, help="Calculate Dice Loss")
    parser.add_argument('--data_len', default=5000, type=int, help='count of your entire data_set')
    parser.add_argument('--nb_classes', default=19, type=int, help='number classes of your dataset (including background); '
                        'the reference admits 19, 21, 172, 151, 9, 2 -- any value <= 192 works here')
    parser.add_argument("--Kvasir_path", type=str, default='/mnt/d/MedicalSeg/Kvasir-SEG/')
    parser.add_argument("--ClinicDB_path", type=str, default='/mnt/d/MedicalSeg/CVC-ClinicDB/')
    parser.add_argument("--synapse_train_base_dir", type=str, default='/mnt/f/Synapse/Synapse/train_npz')
    parser.add_argument("--synapse_val_base_dir", type=str, default='/mnt/f/Synapse/Synapse/test_vol_h5')
    parser.add_argument("--synapse_list_dir", type=str, default='./lists/lists_Synapse')
    parser.add_argument('--batch-size', default=4, type=int)
    parser.add_argument("--val_batch_size", type=int, default=1)
    parser.add_argument('--epochs', default=5, type=int)
    parser.add_argument("--train_print_freq", type=int, default=100)
    parser.add_argument("--val_print_freq", type=int, default=100)
    # Model parameters (train_gpu.py:76-90)
    parser.add_argument('--backbone', default='MiT-B2', type=str, metavar='MODEL',
                        help='Feature extractor: MiT-B0..B5, ConvNeXt, convnextv2_{atto,femto,nano,tiny,base,large,huge}, convnext_pico, '
                             'or any name registered with segmentation_factory_amd.register_backbone')
    parser.add_argument('--pretrained_backbone', default='', type=str, metavar='MODEL')
    parser.add_argument('--heads', default='SegFormerHead', type=str, metavar='MODEL', help='SegFormerHead | UPerHead | registered head')
    # Optimizer parameters (train_gpu.py:92-106)
    parser.add_argument('--opt', default='adamw', type=str, metavar='OPTIMIZER')
    parser.add_argument('--opt-eps', default=1e-8, type=float, metavar='EPSILON')
    parser.add_argument('--opt-betas', default=None, type=float, nargs='+', metavar='BETA')
    parser.add_argument('--clip-grad', type=float, default=0.02, metavar='NORM')
    parser.add_argument('--clip-mode', type=str, default='agc')
    parser.add_argument('--momentum', type=float, default=0.9, metavar='M')
    parser.add_argument('--weight-decay', type=float, default=0.025)
    # Learning rate schedule parameters (train_gpu.py:109-146)
    parser.add_argument('--sched', default='cosine', type=str, metavar='SCHEDULER')
    parser.add_argument('--lr', type=float, default=1e-3, metavar='LR')
    parser.add_argument('--lr-ep', action='store_true', default=False, help='using the epoch-based scheduler')
    parser.add_argument('--lr-noise', type=float, nargs='+', default=None)
    parser.add_argument('--lr-noise-pct', type=float, default=0.67)
    parser.add_argument('--lr-noise-std', type=float, default=1.0)
    parser.add_argument('--lr-cycle-mul', type=float, default=1.0)
    parser.add_argument('--lr-cycle-decay', type=float, default=1.0)
    parser.add_argument('--lr-cycle-limit', type=int, default=1)
    parser.add_argument('--lr-k-decay', type=float, default=1.0)
    parser.add_argument('--warmup-lr', type=float, default=2e-4)
    parser.add_argument('--min-lr', type=float, default=1e-4)
    parser.add_argument('--decay-milestones', default=[30, 60], type=int, nargs='+')
    parser.add_argument('--decay-epochs', type=float, default=30)
    parser.add_argument('--warmup-epochs', type=int, default=5)
    parser.add_argument('--cooldown-epochs', type=int, default=10)
    parser.add_argument('--patience-epochs', type=int, default=10)
    parser.add_argument('--decay-rate', '--dr', type=float, default=0.1)
    # Finetuning params (train_gpu.py:149-155)
    parser.add_argument('--finetune', default='', help='finetune from checkpoint (reference default: a local SegFormer-B2 file)')
    parser.add_argument('--encoder_pretrain_weights', type=str, default='')
    parser.add_argument('--freeze_layers', type=bool, default=True, help='freeze layers')
    parser.add_argument('--set_bn_eval', action='store_true', default=False)
    parser.add_argument('--save_weights_dir', default='./output', help='path where to save, empty for no saving')
    parser.add_argument('--writer_output', default='./', help='path where to save SummaryWriter, empty for no saving')
    parser.add_argument('--device', default='cuda')
    parser.add_argument('--seed', default=0, type=int)
    parser.add_argument('--resume', default='', help='resume from checkpoint')
    parser.add_argument('--eval', action='store_true', help='Perform evaluation only')
    parser.add_argument('--dist-eval', action='store_true', default=False)
    parser.add_argument('--num_workers', default=0, type=int)
    parser.add_argument('--pin-mem', action='store_true')
    parser.add_argument('--no-pin-mem', action='store_false', dest='pin_mem')
    parser.set_defaults(pin_mem=True)
    # distributed (train_gpu.py:177-183)
    parser.add_argument('--world_size', default=1, type=int)
    parser.add_argument('--local_rank', default=0, type=int)
    parser.add_argument('--dist_url', default='env://')
    parser.add_argument('--save_freq', default=1, type=int)
    # MI355X-path extras
    parser.add_argument('--compute-dtype', default='bf16', choices=['bf16', 'fp32'], help='activation storage (fp32 = exact-parity mode)')
    parser.add_argument('--hip-graph', dest='hip_graph', action='store_const', const=True, default=None,
                        help='REQUIRE the replayed-hipGraph train step / eval forward (a capture failure is an error).  Without either flag '
                             'the graph is used whenever the model offers forward_lowres and the optimizer is the fused AGC/AdamW one, and '
                             'a capture failure falls back to eager launches with a printed reason')
    parser.add_argument('--no-hip-graph', dest='hip_graph', action='store_const', const=False,
                        help='per-kernel (eager) launches; under several ranks the model is wrapped in DistributedDataParallel as in the reference')
    parser.add_argument('--eval-dtype', default='fp32', choices=['fp32', 'bf16'],
                        help="precision of evaluate()'s forward: fp32 as the reference (engine.py:86-88 switches autocast off), or the "
                             'bf16 production storage type')
    parser.add_argument('--grad-exchange', default='all_reduce', choices=['all_reduce', 'rs_ag'],
                        help='--hip-graph data parallelism: one all-reduce per gradient bucket, or in-place reduce-scatter + all-gather')
    parser.add_argument('--grad-payload', default='fp32', choices=['fp32', 'bf16'],
                        help='--hip-graph data parallelism: exchange gradients as fp32 (reference arithmetic) or rounded to bf16')
    parser.add_argument('--device-input', action='store_true',
                        help='training batches from the device-side input pipeline (segmentation_factory_amd/transforms.py): the '
                             'decoded uint8 training set is uploaded once and the transform stack of datasets/build_datasets.py:14-22 '
                             'runs as HIP kernels')
    return parser


class SyntheticSegDataset(Dataset):
    """Generated (image, label) pairs honouring the reference's tensor contract (datasets/*.py: fp32 CHW image, int64 HW
    label, ignore 255): the label is a coarse block pattern that is a function of the image, so a model can fit it."""

    def __init__(self, n, size, num_classes, ignore=255, seed=0):
        self.n, self.size, self.nc, self.ignore, self.seed = n, size, num_classes, ignore, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        rng = np.random.default_rng(self.seed * 100003 + i)
        s, blk = self.size, max(self.size // 8, 1)
        coarse = rng.integers(0, self.nc, (s // blk + 1, s // blk + 1))
        lbl = np.kron(coarse, np.ones((blk, blk), dtype=np.int64))[:s, :s].astype(np.int64)
        img = rng.standard_normal((3, s, s), dtype=np.float32) * 0.1
        img += (lbl[None].astype(np.float32) / max(self.nc - 1, 1) - 0.5) * np.array([1.0, -1.0, 0.5], dtype=np.float32)[:, None, None]
        lbl[:2] = self.ignore
        return torch.from_numpy(img), torch.from_numpy(lbl)


def build_dataset(args):
    if args.dataset == 'synthetic':
        n = max(int(args.data_len), 1)
        return (SyntheticSegDataset(n, args.image_size, args.nb_classes, args.ignore_label, args.seed),
                SyntheticSegDataset(max(n // 4, 1), args.image_size, args.nb_classes, args.ignore_label, args.seed + 1))
    try:
        from datasets import build_dataset as ref_build_dataset      # the reference's datasets package, used unchanged
    except Exception as e:
        raise SystemExit(f"--dataset {args.dataset} needs the reference's `datasets` package on PYTHONPATH ({e}); "
                         f"use --dataset synthetic for generated data") from e
    return ref_build_dataset(args)


def _decode_only(image, label):
    """Stands where the reference's train_transform stands (datasets/*.py `self.transform(image, label)`): hands back the decoded
    PIL pair as uint8 tensors, untouched, so the dataset classes of the reference do the file handling and decoding."""
    return torch.from_numpy(np.array(image, dtype=np.uint8)), torch.from_numpy(np.array(label, dtype=np.uint8))


def build_device_loader(args, train_set):
    """--device-input: every training sample is decoded ONCE into HBM; each batch is then two kernel launches."""
    from segmentation_factory_amd.transforms import DeviceBatchLoader, DeviceDataset, DeviceTrainTransform
    ds = DeviceDataset(args.device)
    if args.dataset == 'synthetic':                 # uint8 photographs-to-be: block-pattern labels, colours that encode them
        for i in range(len(train_set)):
            rng = np.random.default_rng(args.seed * 100003 + i)
            s = args.image_size + 32 + 8 * (i % 5)
            blk = max(args.image_size // 8, 1)
            coarse = rng.integers(0, args.nb_classes, (s // blk + 1, s // blk + 1))
            lbl = np.kron(coarse, np.ones((blk, blk), dtype=np.int64))[:s, :s]
            img = (lbl[..., None] * np.array([255, 127, 63]) // max(args.nb_classes - 1, 1) + rng.integers(0, 32, (s, s, 3))) % 256
            ds.add(img.astype(np.uint8), lbl.astype(np.uint8))
    else:                                           # the reference's dataset class decodes; its own label mapping has been applied
        train_set.transform = _decode_only
        for i in range(len(train_set)):
            img, lbl = train_set[i]
            # the reference's dataset classes differ in what they return here: ADE20K / VOC hand the decoded PIL pair through the
            # transform (uint8 tensors from _decode_only), Cityscapes returns encode_target(target) = id_to_train_id[np.array(target)],
            # a numpy int64 array (datasets/cityscapes.py:131,159)
            lbl = torch.as_tensor(np.asarray(lbl))
            if lbl.dtype != torch.uint8:
                assert int(lbl.min()) >= 0 and int(lbl.max()) <= 255, 'label values must fit uint8 (255 = ignore)'
                lbl = lbl.to(torch.uint8)
            ds.add(torch.as_tensor(np.asarray(img)), lbl)
    tf = DeviceTrainTransform(args.image_size, device=args.device)
    # seed 0 = the DistributedSampler of train_gpu.py:212-214 (constructed without a seed)
    return DeviceBatchLoader(ds, args.batch_size, tf, shuffle=True, seed=0, rank=utils.get_rank(), world=utils.get_world_size())


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass


def main(args):
    print(args)
    utils.init_distributed_mode(args)
    writer = None
    if args.local_rank == 0:
        try:
            from torch.utils.tensorboard import SummaryWriter
            writer = SummaryWriter(os.path.join(args.writer_output, 'runs'))
        except Exception:
            writer = _NullWriter()
    seed = args.seed + utils.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    best_mIoU = best_F1 = best_acc = 0.0
    device = args.device
    results_file = "results{}.txt".format(datetime.datetime.now().strftime("%Y%m%d-%H%M%S"))

    train_set, valid_set = build_dataset(args)
    if args.distributed:
        sampler_train = DistributedSampler(train_set, num_replicas=utils.get_world_size(), rank=utils.get_rank(), shuffle=True)
        sampler_val = DistributedSampler(valid_set)
    else:
        sampler_train = RandomSampler(train_set)
        sampler_val = torch.utils.data.SequentialSampler(valid_set)
    trainloader = DataLoader(train_set, batch_size=args.batch_size, num_workers=args.num_workers, drop_last=True,
                             pin_memory=args.pin_mem, sampler=sampler_train)
    if args.device_input:
        trainloader = build_device_loader(args, train_set)
    valloader = DataLoader(valid_set, batch_size=args.val_batch_size, num_workers=args.num_workers, drop_last=True,
                           pin_memory=args.pin_mem, sampler=sampler_val)

    dtype = torch.bfloat16 if args.compute_dtype == 'bf16' else torch.float32
    model = SegmentationModel(args.backbone, pretrained_backbone=args.pretrained_backbone, num_classes=args.nb_classes,
                              seg_head=args.heads, compute_dtype=dtype, args=args).to(device)
    model_without_ddp = model
    if args.finetune:       # train_gpu.py:238-260
        checkpoint_model = utils.load_model(args.finetune, model)
        for k in list(checkpoint_model.keys()):
            if 'linear_pred' in k:
                print(f"Removing key {k} from pretrained checkpoint")
                del checkpoint_model[k]
        print(model.load_state_dict(checkpoint_model, strict=False))
        if args.freeze_layers:
            for name, para in model.named_parameters():
                para.requires_grad_('linear_pred' in name)
                if 'linear_pred' in name:
                    print('training {}'.format(name))
    if args.distributed and args.hip_graph is False:
        # wrapped AFTER the finetune load / freeze (the reducer must see the final requires_grad flags), and with
        # find_unused_parameters=True as the reference does (train_gpu.py:233-236: e.g. FPNHead.output_convs[0] is never used)
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[args.gpu] if device == 'cuda' else None,
                                                          find_unused_parameters=True)
        model_without_ddp = model.module

    n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print('\n********ESTABLISH ARCHITECTURE********')
    print(f'Model: {model_without_ddp}\nNumber of parameters: {n_parameters}')
    print('**************************************\n')

    if args.finetune:      # train_gpu.py:269: torch.optim.AdamW(model.parameters(), lr=2e-4, weight_decay): ONE group, biases decay too
        optimizer = FusedAGCAdamW(model_without_ddp.parameters(), lr=2e-4, weight_decay=args.weight_decay)
    else:
        optimizer = create_optimizer(args, model_without_ddp)
    loss_scaler = NativeScaler()
    lr_scheduler, _ = create_scheduler(args, optimizer)

    output_dir = Path(args.save_weights_dir)
    if args.save_weights_dir and utils.is_main_process():
        with (output_dir / "model.txt").open("a") as f:
            f.write(str(model_without_ddp))
        with (output_dir / "args.txt").open("a") as f:
            f.write(json.dumps({k: v for k, v in args.__dict__.items()}, indent=2, default=str) + "\n")

    checkpoint_name = utils.get_pth_file(args.save_weights_dir) if args.save_weights_dir else None
    if checkpoint_name:                                  # auto-resume (train_gpu.py:281-307: any *.pth in the save dir wins)
        args.resume = os.path.join(f'{args.save_weights_dir}/', checkpoint_name)
    elif args.resume and not os.path.isfile(args.resume):    # quirk Q15: the reference dies with a TypeError here
        raise SystemExit(f'--resume {args.resume}: no such file (and no *.pth in --save_weights_dir to auto-resume from)')
    if args.resume:                                      # an explicit --resume is honoured when the save dir holds no checkpoint
        print("Loading local checkpoint at {}".format(args.resume))
        checkpoint = torch.load(args.resume, map_location='cpu', weights_only=False)
        print(model_without_ddp.load_state_dict(checkpoint['model_state']))
        if not args.eval:
            optimizer.load_state_dict(checkpoint['optimizer_state'])
            lr_scheduler.load_state_dict(checkpoint['scheduler_state'])
            best_mIoU, best_F1, best_acc = checkpoint['best_mIoU'], checkpoint['F1_Score'], checkpoint['Acc']
            print(f'Now max mIOU is {best_mIoU}\n')
            print(f'Now max F1-score is {best_F1}\n')
            print(f'Now max Accuracy is {best_acc}\n')
            if 'scaler' in checkpoint:
                loss_scaler.load_state_dict(checkpoint['scaler'])

    def summarise(confmat, metric):
        mean_iou = round(confmat.compute()[2].mean().item() * 100, 2)       # NaN if a class is absent (quirk Q6)
        _, mean_f1 = metric.compute_f1()
        _, mean_acc = metric.compute_pixel_acc()
        return mean_iou, mean_f1, mean_acc

    if args.eval:
        print(f"Evaluating model: {args.backbone}_{args.heads}")
        mean_iou, mean_f1, mean_acc = summarise(*evaluate(args, model, valloader, device, args.val_print_freq))
        print(f"**val_meanF1: {mean_f1}\n**val_meanACC: {mean_acc}\n**val_mIOU: {mean_iou}")
        return

    print(f"Start training for {args.epochs} epochs")
    for epoch in range(args.epochs):
        if args.device_input:
            trainloader.set_epoch(epoch)
        elif args.distributed:
            trainloader.sampler.set_epoch(epoch)
        mean_loss, lr = train_one_epoch(model, optimizer, trainloader, epoch, device, args.train_print_freq, args.clip_grad,
                                        args.clip_mode, loss_scaler, writer, args)
        confmat, metric = evaluate(args, model, valloader, device, args.val_print_freq, writer)
        mean_iou, mean_f1, mean_acc = summarise(confmat, metric)
        print(f"**Val_meanF1: {mean_f1}\n**Val_meanACC: {mean_acc}\n**Val_mIOU: {mean_iou}")
        lr_scheduler.step(epoch)
        val_info = f'{str(confmat)}\nval_meanF1: {mean_f1}\nval_meanACC: {mean_acc}'
        print(val_info)
        if utils.is_main_process():
            with open(results_file, "a") as f:
                f.write(f"[epoch: {epoch}]\ntrain_loss: {mean_loss:.4f}\nlr: {lr:.6f}\n" + val_info + "\n\n")
        if mean_iou > best_mIoU:
            print(f'Increasing mIoU: from {best_mIoU} to {mean_iou}!\n')
            best_mIoU = mean_iou
            print(f'Max mIOU: {best_mIoU}\n')
            if utils.is_main_process() and args.save_weights_dir:
                torch.save({"model_state": model_without_ddp.state_dict(), "optimizer_state": optimizer.state_dict(),
                            "scheduler_state": lr_scheduler.state_dict(), "best_mIoU": mean_iou, "F1_Score": mean_f1,
                            "Acc": mean_acc, "scaler": loss_scaler.state_dict()},
                           f'{args.save_weights_dir}/{args.backbone}_{args.heads}_best_model.pth')
                print('******************Save Checkpoint******************')
                print(f'Save weights to {args.save_weights_dir}/{args.backbone}_{args.heads}_best_model.pth\n')
        else:
            print('*********No improving mIOU, No saving checkpoint*********')


if __name__ == '__main__':
    parser = argparse.ArgumentParser('Segmentation Models training and evaluation script', parents=[get_args_parser()])
    args = parser.parse_args()
    if args.save_weights_dir:
        Path(args.save_weights_dir).mkdir(parents=True, exist_ok=True)
    main(args)
