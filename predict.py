#!/usr/bin/env python3
"""Zero-TIG inference on MI355X -- same flags and outputs as the reference predict.py:23-36, 76-104."""
import argparse
import logging
import os
import sys

import numpy as np
import torch
import torch.utils.data
from PIL import Image

from dataloader.create_data import CreateDataset
from model.model import Finetunemodel
from utils import utils
from utils.utils import sequential_judgment

parser = argparse.ArgumentParser("ZERO-TIG")
parser.add_argument("--lowlight_images_path", type=str, default="./data")
parser.add_argument("--save", type=str, default="./results/")
parser.add_argument("--model_pretrain", type=str, default=r"./weights/BVI-RLV.pt")
parser.add_argument("--gpu", type=int, default=0)
parser.add_argument("--seed", type=int, default=2)
parser.add_argument("--of_scale", type=int, default=3)
parser.add_argument("--dataset", type=str, default="RLV")
parser.add_argument("--num_workers", type=int, default=-1, help="decode workers; -1: host cores - 2, at most 12")


def save_images(tensor):
    """predict.py:57-61: clip(x * 255, 0, 255).astype(uint8), HWC -- quantised and interleaved on the device (6 MB instead of
    25 MB per 1080p frame over PCIe)."""
    return utils.quantize_u8(tensor).cpu().numpy()


def main():
    args = parser.parse_args()
    os.makedirs(args.save, exist_ok=True)
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format="%(asctime)s %(message)s")
    dev = torch.device("cuda", args.gpu)
    args.device_ingest = True                      # loaders decode only; resize + ToTensor (multi_read_data.py:127-132) on the GPU
    test_set = CreateDataset(args, task="test")
    queue = torch.utils.data.DataLoader(test_set, batch_size=1, **utils.loader_kwargs(utils.loader_workers(args.num_workers)))
    print("Total image number: ", len(test_set))
    model = Finetunemodel(args).to(dev)
    model.eval()
    for p in model.parameters():
        p.requires_grad = False
    with torch.no_grad():
        for i, (inp, img_name, img_path, last_img_path) in enumerate(queue):
            model.is_new_seq = i == 0 or sequential_judgment(img_path[0], last_img_path[0])
            enhance, output, illum = model(utils.ingest_frame(inp, dev))
            if "RLV" == args.dataset:
                parts = img_path[0].split(os.sep)
                save_dir = os.path.join(args.save, parts[-3], parts[-2])
            else:
                save_dir = os.path.join(args.save, os.path.basename(os.path.split(img_path[0])[0]))
            os.makedirs(save_dir, exist_ok=True)
            name = img_name[0].split("/")[-1].split(".")[0]
            Image.fromarray(save_images(output)).save(save_dir + "/" + name + "_denoise.png", "PNG")
            Image.fromarray(save_images(enhance)).save(save_dir + "/" + name + "_enhance.png", "PNG")


if __name__ == "__main__":
    main()
