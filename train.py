#!/usr/bin/env python3
"""Zero-TIG self-supervised training on MI355X -- same flags and loop semantics as the reference train.py:15-27, 116-152,
one process per GPU.  Single GPU: `python train.py --lowlight_images_path DATA`.  One node, N GPUs (frame-clip data parallel,
one RCCL all-reduce of the flat 370 KB gradient bucket per step):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 train.py ...
"""
import argparse
import glob
import logging
import os
import sys
import time

import numpy as np
import torch
import torch.utils.data
from PIL import Image

from dataloader.create_data import CreateDataset
from model.model import Network
from utils import utils

optim = __import__("importlib").import_module("zero-tig_amd.optim")

parser = argparse.ArgumentParser("ZERO-TIG")
parser.add_argument("--batch_size", type=int, default=1, help="batch size (frames per GPU per step; the recurrent cache needs 1)")
parser.add_argument("--cuda", default=True, type=bool, help="kept for CLI compatibility")
parser.add_argument("--gpu", type=str, default="0", help="gpu device id (single-process runs)")
parser.add_argument("--seed", type=int, default=2, help="random seed")
parser.add_argument("--epochs", type=int, default=5, help="epochs")
parser.add_argument("--lr", type=float, default=0.0001, help="learning rate")
parser.add_argument("--save", type=str, default="./EXP/", help="experiment directory")
parser.add_argument("--model_pretrain", type=str, help="checkpoint to start from")
parser.add_argument("--lowlight_images_path", type=str, default="", help="input data folder")
parser.add_argument("--of_scale", type=int, default=3, help="downscale factor for optical flow")
parser.add_argument("--dataset", type=str, default="RLV", help="dataset name")
parser.add_argument("--num_workers", type=int, default=-1, help="dataloader (decode) workers; -1: host cores - 2, at most 12")
parser.add_argument("--host_ingest", action="store_true", help="resize + ToTensor on the host like the reference (default: decode only, the rest on the device)")
parser.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"], help="bf16 throughput mode / fp32 parity mode")
parser.add_argument("--graph", type=int, default=1, help="1: replay the steady-state step from a captured hipGraph (zero-tig_amd/optim.py:TrainStep); 0: eager launches")
parser.add_argument("--resume", type=str, default=None, help="resume file written every epoch (model + Adam moments + step + loop position)")
parser.add_argument("--reference_eval_quirk", action="store_true", help="stay in eval() after the first epoch like the reference (train.py:138)")


def save_images(tensor):
    """predict.py:57-61: clip(x * 255, 0, 255).astype(uint8), HWC -- quantised and interleaved on the device (6 MB instead of
    25 MB per 1080p frame over PCIe)."""
    return utils.quantize_u8(tensor).cpu().numpy()


def main():
    args = parser.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", args.gpu.split(",")[0]))
    if not torch.cuda.is_available():
        logging.info("no gpu device available")
        sys.exit(1)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    args.save = args.save + "/" + "Train-{}".format(time.strftime("%Y%m%d-%H%M%S"))
    if rank == 0:
        utils.create_exp_dir(args.save, scripts_to_save=glob.glob("*.py"))
    model_path = args.save + "/model_epochs/"
    os.makedirs(model_path, exist_ok=True)
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format="%(asctime)s %(message)s", datefmt="%m/%d %I:%M:%S %p")
    if rank == 0:
        fh = logging.FileHandler(os.path.join(args.save, "log.txt"))
        logging.getLogger().addHandler(fh)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    logging.info("args = %s", args)

    model = Network(args, precision=args.precision)
    if rank == 0:
        utils.save(model, os.path.join(args.save, "initial_weights.pt"))
    model.enhance.in_conv.apply(model.enhance_weights_init)
    model.enhance.conv.apply(model.enhance_weights_init)
    model.enhance.out_conv.apply(model.enhance_weights_init)
    try:
        base = torch.load(args.model_pretrain)
        md = model.state_dict()
        md.update({k: v for k, v in base.items() if k in md})
        model.load_state_dict(md)
        logging.info("Loaded pre-trained model from %s." % args.model_pretrain)
    except Exception:
        logging.info("Model is initialized without pre-trained model.")
    model = model.cuda(dev)
    if world > 1:       # identical start on every rank (21 MB incl. the frozen RAFT)
        import torch.distributed as dist
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
    optimizer = optim.ClipAdam(model, lr=args.lr, betas=(0.9, 0.999), weight_decay=3e-4, max_norm=5.0)
    logging.info("model size = %f", utils.count_parameters_in_MB(model))

    args.device_ingest = not args.host_ingest      # loaders deliver the decoded uint8 frame; resize + ToTensor run on the GPU
    workers = utils.loader_workers(args.num_workers)
    train_set = CreateDataset(args, task="train")
    test_set = CreateDataset(args, task="test")
    # data parallelism at clip granularity: each rank walks a contiguous share of the (temporally ordered) frame list
    n = len(train_set)
    per = (n + world - 1) // world
    idx = list(range(rank * per, min(n, (rank + 1) * per)))
    steps_per_epoch = per if world == 1 else min(len(range(r * per, min(n, (r + 1) * per))) for r in range(world))
    train_queue = torch.utils.data.DataLoader(torch.utils.data.Subset(train_set, idx), batch_size=1, **utils.loader_kwargs(workers))
    test_queue = torch.utils.data.DataLoader(test_set, batch_size=1, **utils.loader_kwargs(min(workers, 4)))

    stepper = optim.TrainStep(model, optimizer, use_graph=bool(args.graph))
    total_step, first_epoch = 0, 0
    if args.resume:
        first_epoch, total_step = utils.load_checkpoint(model, optimizer, args.resume)
        logging.info("resumed from %s: epoch %d, step %d", args.resume, first_epoch, total_step)
    model.train()
    for epoch in range(first_epoch, args.epochs):
        losses = []
        t_epoch = time.perf_counter()
        # pinned frames reach HBM on a copy stream, one frame ahead of the step that consumes them
        for it, (inp, img_name, img_path, last_img_path) in enumerate(optim.FramePrefetcher(train_queue, dev)):
            if it >= steps_per_epoch:
                break
            new_seq = it == 0 or utils.sequential_judgment(img_path[0], last_img_path[0])
            total_step += 1
            # zero_grad + _loss + backward + (all-reduce) + clip_grad_norm_(5) + Adam (train.py:126-131); pinned frame -> HBM inside
            loss = stepper(inp, is_new_seq=new_seq)
            losses.append(loss.item())
            logging.info("train-epoch %03d %03d %f", epoch, it, losses[-1])
        logging.info("train-epoch %03d %f", epoch, np.average(losses))
        torch.cuda.synchronize(dev)
        dt_epoch = time.perf_counter() - t_epoch
        logging.info("train-epoch %03d throughput: %d frames in %.2f s = %.1f frames/s per rank, files -> decode (%d workers) -> %s -> step",
                     epoch, len(losses), dt_epoch, len(losses) / max(dt_epoch, 1e-9), workers,
                     "PCIe (uint8) -> device resize + ToTensor" if args.device_ingest else "host resize + ToTensor -> PCIe (fp32)")
        if rank == 0:
            utils.save(model, os.path.join(model_path, "weights_%d.pt" % epoch))        # the reference's plain state_dict (train.py:135)
        utils.save_checkpoint(model, optimizer, os.path.join(model_path, "resume.pt"), epoch=epoch + 1, step=total_step)
        # every rank switches mode together: under DP the gradients that get all-reduced must come from the same BN mode
        model.eval()
        if rank == 0:
            with torch.no_grad():
                for it, (inp, img_name, img_path, last_img_path) in enumerate(test_queue):
                    model.is_new_seq = it == 0 or utils.sequential_judgment(img_path[0], last_img_path[0])
                    outs = model(utils.ingest_frame(inp, dev))
                    H2, H3 = outs[6], outs[13]
                    name = "%s_%s" % (os.path.basename(os.path.split(img_path[0])[0]), img_name[0])
                    os.makedirs(args.save + "/result/denoise/", exist_ok=True)
                    os.makedirs(args.save + "/result/enhance/", exist_ok=True)
                    Image.fromarray(save_images(H3)).save(args.save + "/result/denoise/" + name + "_denoise_" + str(epoch) + ".png", "PNG")
                    Image.fromarray(save_images(H2)).save(args.save + "/result/enhance/" + name + "_enhance_" + str(epoch) + ".png", "PNG")
        # NOTE: the reference stays in eval() from here on (train.py:138, SURVEY A-14); this loop returns to train mode.
        if not args.reference_eval_quirk:
            model.train()


if __name__ == "__main__":
    main()
