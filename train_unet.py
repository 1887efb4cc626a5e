#!/usr/bin/env python3
"""Train the U-Net segmentation network on the MI355X engine.

Same flags and defaults as /root/reference/train_unet.py:27-79 (--name --samples --d --epochs --lr
--batch_size --nrot --nclasses --split), same paths (data/<name>/matrices, saved_models/unet/<name>/
unet_weights_<name>.best.hdf5, output/unet/<name>).  Added: --channels (the reference hard-codes 4,
train_unet.py:84) and --synthetic N (Gaussian-blob grids of SURVEY 8(d); no dataset exists offline).

  python3 train_unet.py --name heusler --samples 5000 --epochs 100
  python3 train_unet.py --name demo --synthetic 64 --channels 1 --epochs 2 --batch_size 8

Data parallel (new; the reference is single-process): launch one process per GPU,
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_unet.py ...
--batch_size is then the PER-GPU batch, the id lists are dealt round-robin over the ranks, gradients are
averaged with RCCL inside the engine; --sync_bn 1 exchanges the BatchNorm batch statistics as well.  Under the
launcher spell --d as --dim: torch.distributed.run's own parser rejects `--d` as an ambiguous abbreviation.
"""
import argparse
import json
import os
import sys

import numpy as np

from icsg3d_amd.dataparallel import from_env, shard_ids
from icsg3d_amd.unet.data import SyntheticUnetGenerator, UnetDataGenerator
from icsg3d_amd.unet.get_weights import get_weights
from icsg3d_amd.unet.unet import AtomUnet
from icsg3d_amd.utils import data_split

if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--name", type=str, help="Name of data folder")
    p.add_argument("--samples", type=int, default=20000, help="Total number of training and validation samples")
    p.add_argument("--d", "--dim", dest="d", type=int, default=32, help="Dimension of density matrices (number of voxels)")
    p.add_argument("--epochs", type=int, default=50)
    p.add_argument("--lr", type=float, default=3e-6)
    p.add_argument("--batch_size", type=int, default=10)
    p.add_argument("--nrot", type=int, default=10)
    p.add_argument("--nclasses", type=int, default=95)
    p.add_argument("--split", type=float, default=0.8)
    p.add_argument("--channels", type=int, default=4, help="input channels (density + 3 coordinate grids)")
    p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic grids instead of data/<name>")
    p.add_argument("--sync_bn", type=int, default=0, help="data parallel: BatchNorm statistics over all ranks")
    p.add_argument("--gpus", type=int, default=0,
                   help="data parallel over N GPUs of this node: without torch.distributed.run the script starts its N "
                        "ranks itself (icsg3d_amd/launcher.py); 0 = whatever the environment says (default)")
    a = p.parse_args()

    if a.gpus:                                   # before anything touches the GPU; exits with the ranks' status
        from icsg3d_amd.launcher import ensure_ranks
        ensure_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:])
    dp = from_env()                              # (dist, rank, world, local_rank) under torch.distributed.run
    rank, world = (dp[1], dp[2]) if dp else (0, 1)
    mode, d = a.name, a.d
    path = os.path.join("data", mode, "matrices")
    input_shape = (d, d, d, a.channels)
    weights_dir = os.path.join("saved_models", "unet", mode)
    os.makedirs(weights_dir, exist_ok=True)
    os.makedirs(os.path.join("output/unet", mode), exist_ok=True)
    weights = os.path.join(weights_dir, "unet_weights_" + mode + ".best.hdf5")

    if a.synthetic:
        n_train = int(a.synthetic * a.split)
        n_val = a.synthetic - n_train
        if world > 1:                            # equal whole batches per rank, a different stream per rank
            n_train, n_val = (n // (world * a.batch_size) * a.batch_size for n in (n_train, n_val))
        training_generator = SyntheticUnetGenerator(n_train, a.batch_size, (d, d, d), a.channels, a.nclasses,
                                                    seed=rank * 10 ** 7)
        validation_generator = SyntheticUnetGenerator(n_val, a.batch_size, (d, d, d), a.channels,
                                                      a.nclasses, seed=10 ** 6 + rank * 10 ** 7)
        class_weights = get_weights()
    else:
        training_ids, validation_ids = data_split(path, a.samples, frac=a.split, n_rot=a.nrot)
        all_training_ids, all_validation_ids = training_ids, validation_ids
        if world > 1:
            training_ids = shard_ids(training_ids, rank, world, a.batch_size)
            validation_ids = shard_ids(validation_ids, rank, world, a.batch_size)
        if rank == 0:                            # provenance: which files this run trained / validated on, in order
            with open(os.path.join("output", "unet", mode, "split_ids.json"), "w") as f:
                json.dump({"train": all_training_ids, "val": all_validation_ids}, f)
        training_generator = UnetDataGenerator(training_ids, data_path=path, batch_size=a.batch_size, dim=(d, d, d),
                                               n_channels=a.channels, shuffle=True)
        validation_generator = UnetDataGenerator(validation_ids, data_path=path, batch_size=a.batch_size, dim=(d, d, d),
                                                 n_channels=a.channels, shuffle=True)
        try:
            class_weights = np.load(weights_dir + "/class_weights.npy")
        except Exception:
            class_weights = get_weights(path, all_training_ids, a.nclasses)
            class_weights[0] = 0.0
            if rank == 0:
                np.save(weights_dir + "/class_weights.npy", class_weights)

    unet = AtomUnet(num_classes=a.nclasses, class_weights=class_weights, input_shape=input_shape, weights=weights,
                    lr=a.lr, max_batch=a.batch_size)
    if dp:
        unet.enable_data_parallel(dp[0], rank, world, sync_bn=bool(a.sync_bn), force=world == 1)
    unet.train_generator(training_generator, validation_generator, epochs=a.epochs,
                         output_dir=os.path.join("output", "unet", mode))
    if rank == 0:
        unet.save_(weights, os.path.splitext(weights)[0] + ".h5")
    if dp:
        dp[0].barrier()
        dp[0].destroy_process_group()
