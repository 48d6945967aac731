#!/usr/bin/env python3
"""Multi-dataset train -> evaluate orchestration (reference run_pipeline.py:63-171, same flags) for the MI355X scripts: per dataset
folder, fine-tune with train.py (one process per GPU when --gpus > 1, started through torch.distributed.run), then evals.py with
the last epoch's weights, and log the Metrics.json summary."""
import argparse
import glob
import json
import logging
import os
import subprocess
import sys

DATASET_TYPES = {"lowlight_dataset": "lowlight_dataset", "RLV": "RLV", "BVI-RLV": "RLV", "DID_1080": "DID", "SDSD-indoor": "SDSD",
                 "SDSD-outdoor": "SDSD", "3_SDSD": "SDSD"}


def run(cmd, log):
    log.info("Executing command: %s", " ".join(cmd))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, bufsize=1)
    for line in p.stdout:
        log.info(line.rstrip())
    return p.wait() == 0


def main():
    ap = argparse.ArgumentParser(description="Zero-TIG training + evaluation pipeline")
    ap.add_argument("--datasets", nargs="+", required=True)
    ap.add_argument("--base_data_dir", type=str, default="./data/")
    ap.add_argument("--weights_dir", type=str, default="./weights/")
    ap.add_argument("--pretrain_weights_file", type=str, default="BVI-RLV.pt")
    ap.add_argument("--base_exp_dir", type=str, default="./PIPELINE_EXP")
    ap.add_argument("--num_workers", type=int, default=0)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of this node used for training (frame-clip data parallel over RCCL)")
    a = ap.parse_args()
    os.makedirs(a.base_exp_dir, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s",
                        handlers=[logging.FileHandler(os.path.join(a.base_exp_dir, "pipeline_log.txt"), mode="w"), logging.StreamHandler(sys.stdout)])
    log = logging.getLogger()
    here = os.path.dirname(os.path.abspath(__file__))
    ok_all = True
    for name in a.datasets:
        log.info("========== PROCESSING DATASET: %s ==========", name)
        data = os.path.join(a.base_data_dir, name)
        if not os.path.isdir(data):
            log.error("Dataset directory not found: %s. Skipping.", data)
            ok_all = False
            continue
        train_dir = os.path.join(a.base_exp_dir, name, "training")
        eval_dir = os.path.join(a.base_exp_dir, name, "evaluation")
        os.makedirs(train_dir, exist_ok=True)
        os.makedirs(eval_dir, exist_ok=True)
        dtype = DATASET_TYPES.get(name, name)
        launcher = [sys.executable] if a.gpus <= 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
                                                         "--master-addr", "127.0.0.1"]
        train = launcher + [os.path.join(here, "train.py"), "--dataset", dtype, "--lowlight_images_path", data, "--model_pretrain",
                            os.path.join(a.weights_dir, a.pretrain_weights_file), "--save", train_dir, "--epochs", str(a.epochs),
                            "--num_workers", str(a.num_workers)]
        if not run(train, log):
            log.error("Training failed for %s. Skipping to next dataset.", name)
            ok_all = False
            continue
        runs = glob.glob(os.path.join(train_dir, "Train-*"))
        weights = os.path.join(max(runs, key=os.path.getctime), "model_epochs", "weights_%d.pt" % (a.epochs - 1)) if runs else ""
        if not os.path.exists(weights):
            log.error("Final weights file not found at %s. Skipping.", weights)
            ok_all = False
            continue
        if not run([sys.executable, os.path.join(here, "evals.py"), "--dataset", dtype, "--lowlight_images_path", data, "--model_pretrain", weights,
                    "--save", eval_dir], log):
            log.error("Evaluation failed for %s.", name)
            ok_all = False
            continue
        mj = os.path.join(eval_dir, "Metrics.json")
        if os.path.exists(mj):
            log.info("--- FINAL PERFORMANCE for %s --- %s", name, json.load(open(mj)))
        log.info("========== FINISHED DATASET: %s ==========", name)
    log.info("Pipeline has completed for all datasets.")
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
