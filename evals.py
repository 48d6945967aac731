#!/usr/bin/env python3
"""Inference + PSNR against ground truth -- the PSNR leg of the reference evals.py (flags evals.py:26-39, loop 107-170, metric
83-85, summary 184-192) with the metric computed on the device as an exact integer reduction.  SSIM (skimage), LPIPS (lpips /
VGG weights) and histogram matching (skimage) are third-party and absent here: their fields are written as null.
Ground truth: `<...>/input/<scene>/low_light_*/N.png` -> `<...>/gt/<scene>/normal_light_*/N.png` (evals.py:122)."""
import argparse
import json
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

parser = argparse.ArgumentParser("ZERO-IG")
parser.add_argument("--lowlight_images_path", type=str, default="./lowlight_dataset")
parser.add_argument("--save", type=str, default="./results/BVI-RLV")
parser.add_argument("--model_pretrain", type=str, default=r"./weights_1.pt")
parser.add_argument("--gpu", type=int, default=0)
parser.add_argument("--seed", type=int, default=2)
parser.add_argument("--of_scale", type=int, default=3)
parser.add_argument("--dataset", type=str, default="RLV")
parser.add_argument("--gain", type=int, default=100, help="kept for CLI compatibility (unused upstream as well)")
parser.add_argument("--save_images", type=int, default=20, help="write the first N result pairs (evals.py:162)")


def main():
    args = parser.parse_args()
    os.makedirs(args.save, exist_ok=True)
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format="%(asctime)s %(message)s", datefmt="%m/%d %I:%M:%S %p")
    logging.getLogger().addHandler(logging.FileHandler(os.path.join(args.save, "log.txt")))
    dev = torch.device("cuda", args.gpu)
    args.device_ingest = True                      # loaders decode only; resize + ToTensor (multi_read_data.py:127-132) on the GPU
    test_set = CreateDataset(args, task="test")
    queue = torch.utils.data.DataLoader(test_set, batch_size=1, **utils.loader_kwargs(utils.loader_workers(-1)))
    logging.info("Total image number: %d; model path = %s", len(test_set), args.model_pretrain)
    model = Finetunemodel(args).to(dev)
    model.eval()
    total, n = 0.0, 0
    with torch.no_grad():
        for i, (inp, img_name, img_path, last_img_path) in enumerate(queue):
            model.is_new_seq = i == 0 or utils.sequential_judgment(img_path[0], last_img_path[0])
            enhance, output, illum = model(utils.ingest_frame(inp, dev))   # Finetunemodel.forward updates the recurrent cache itself
            gt_path = img_path[0].replace("input", "gt").replace("low_light_", "normal_light_")
            gt = torch.from_numpy(np.asarray(Image.open(gt_path).convert("RGB"), dtype=np.uint8).copy())
            gt_t = utils.ingest_frame(gt, dev)                     # same resize to 1920 x 1080 + ToTensor as the inputs, on the device
            psnr = utils.psnr(output, gt_t)                        # evals.py:83-85, exact integer sum on the device
            total, n = total + psnr, n + 1
            logging.info("NUM: %d, PSNR: %.3f, Total PSNR: %.3f", n, psnr, total / n)
            if i < args.save_images:
                parts = img_path[0].split(os.sep)
                save_dir = os.path.join(args.save, parts[-3] + "/" + parts[-2])
                os.makedirs(save_dir, exist_ok=True)
                name = img_name[0].split("/")[-1].split(".")[0]
                Image.fromarray(utils.quantize_u8(output).cpu().numpy()).save(save_dir + "/" + name + "_denoise.png", "PNG")
                Image.fromarray(utils.quantize_u8(enhance).cpu().numpy()).save(save_dir + "/" + name + "_enhance.png", "PNG")
    with open(os.path.join(args.save, "Metrics.json"), "w") as fh:
        json.dump({"Total_PSNR": total / max(n, 1), "Total_SSIM": None, "Total_LPIPS": None, "Total_PSNR_HM": None, "Total_SSIM_HM": None,
                   "Total_LPIPS_HM": None, "images": n}, fh)


if __name__ == "__main__":
    main()
