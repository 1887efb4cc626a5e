#!/usr/bin/env python3
"""Train the Cond-DFC-VAE on the MI355X engine.

Same flags and defaults as /root/reference/train_vae.py:29-83 (--name --samples --epochs --batch_size
--ncond --nrot --cond --split --d), same paths; the perceptual model is the U-Net checkpoint
saved_models/unet/<name>/unet_weights_<name>.best.h5 written by train_unet.py.  Unlike the reference
 --d reaches the model.  Added: --channels, --synthetic N, and data parallelism: under
`python3 -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train_vae.py ...` every rank
trains on its share of the ids at the PER-GPU --batch_size (train_unet.py explains the flags).
"""
import argparse
import json
import os
import sys

from icsg3d_amd.dataparallel import from_env, shard_ids
from icsg3d_amd.utils import data_split
from icsg3d_amd.vae.data import SyntheticVAEGenerator, VAEDataGenerator
from icsg3d_amd.vae.lattice_vae import LatticeDFCVAE

if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--name", type=str, help="Name of data folder")
    p.add_argument("--samples", type=int, default=40000)
    p.add_argument("--epochs", type=int, default=50)
    p.add_argument("--batch_size", type=int, default=20)
    p.add_argument("--ncond", type=int, default=10)
    p.add_argument("--nrot", type=int, default=10)
    p.add_argument("--cond", type=str, default="formation_energy_per_atom")
    p.add_argument("--split", type=float, default=0.8)
    p.add_argument("--d", "--dim", dest="d", type=int, default=32)
    p.add_argument("--channels", type=int, default=4)
    p.add_argument("--synthetic", type=int, default=0)
    p.add_argument("--sync_bn", type=int, default=0)
    p.add_argument("--gpus", type=int, default=0,
                   help="data parallel over N GPUs of this node: without torch.distributed.run the script starts its N "
                        "ranks itself (icsg3d_amd/launcher.py); 0 = whatever the environment says (default)")
    a = p.parse_args()

    if a.gpus:                                   # before anything touches the GPU; exits with the ranks' status
        from icsg3d_amd.launcher import ensure_ranks
        ensure_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:])
    dp = from_env()
    rank, world = (dp[1], dp[2]) if dp else (0, 1)
    mode, d, bs = a.name, a.d, a.batch_size
    path = os.path.join("data", mode, "matrices")
    csv_path = os.path.join("data", mode, mode + ".csv")
    input_shape = (d, d, d, a.channels)
    weights_dir = os.path.join("saved_models", "vae", mode)
    os.makedirs(weights_dir, exist_ok=True)
    os.makedirs(os.path.join("output", "vae", mode), exist_ok=True)
    weights = os.path.join(weights_dir, "vae_weights_" + mode + ".best.hdf5")
    perceptual_model = os.path.join("saved_models", "unet", mode, "unet_weights_" + mode + ".best.h5")

    if a.synthetic:
        n_train = int(a.synthetic * a.split) // (bs * world) * bs
        n_val = (a.synthetic - int(a.synthetic * a.split)) // (bs * world) * bs
        training_generator = SyntheticVAEGenerator(n_train, bs, (d, d, d), a.channels, a.ncond, seed=rank * 10 ** 7)
        validation_generator = SyntheticVAEGenerator(n_val, bs, (d, d, d), a.channels, a.ncond,
                                                     seed=10 ** 6 + rank * 10 ** 7)
    else:
        training_ids, validation_ids = data_split(path, a.samples, frac=a.split, n_rot=a.nrot)
        if len(training_ids) % bs != 0:          # ids must be a multiple of the batch size (train_vae.py:108-111)
            training_ids = training_ids[:-1 * int(len(training_ids) % bs)]
        if len(validation_ids) % bs != 0:
            validation_ids = validation_ids[:-1 * int(len(validation_ids) % bs)]
        if rank == 0:                            # provenance: which files this run trained / validated on, in order
            with open(os.path.join("output", "vae", mode, "split_ids.json"), "w") as f:
                json.dump({"train": training_ids, "val": validation_ids}, f)
        if world > 1:
            training_ids = shard_ids(training_ids, rank, world, bs)
            validation_ids = shard_ids(validation_ids, rank, world, bs)
        print(len(training_ids), len(validation_ids))
        kw = dict(data_path=path, property_csv=csv_path, batch_size=bs, dim=(d, d, d), n_channels=a.channels,
                  shuffle=True, n_bins=a.ncond, target=a.cond)
        training_generator = VAEDataGenerator(training_ids, **kw)
        validation_generator = VAEDataGenerator(validation_ids, **kw)

    lattice_vae = LatticeDFCVAE(input_shape=input_shape, perceptual_model=perceptual_model, cond_shape=a.ncond,
                                output_dir=os.path.join("output", "vae", mode))
    if dp:
        lattice_vae.enable_data_parallel(dp[0], rank, world, sync_bn=bool(a.sync_bn), force=world == 1)
    lattice_vae.train(training_generator, validation_generator, epochs=a.epochs, weights=weights)
    if dp:
        dp[0].barrier()
        dp[0].destroy_process_group()
