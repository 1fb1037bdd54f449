#!/usr/bin/env python
"""Command-line trainer / evaluator for the MI355X DRN path: the counterpart of the reference's main.py.

    python train.py --root /data/DRN --config data/default_config.yaml --stage 1 --snapshot-pref runs/s1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --stage 3 --resume runs/s2/...
    python train.py --stage 3 --resume <ckpt> --evaluate

`--root` holds the reference's layout (data/dataset/Charades/*.txt|json, the props file and feature_root named by the
config's `Charades` section, data/glove_weights).  One process per GPU; each rank reads its own shard of the training
set (DistributedSampler) and the gradients are averaged over RCCL (drn_amd.dist)."""
import argparse
import functools
import json
import os

import torch
import yaml
from torch.utils.data import DataLoader

from drn_amd import dist as ddist
from drn_amd import trainer as T
from drn_amd.data import CharadesSTA, ShardSampler, collate_data
from drn_amd.model import mainModel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default=".")
    ap.add_argument("--config", default="data/default_config.yaml")
    ap.add_argument("--dataset", default="Charades")
    ap.add_argument("--feature-type", default="C3D")
    ap.add_argument("--stage", type=int, default=1, choices=[1, 2, 3])
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--n-epoch", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=None, help="clips per GPU")
    ap.add_argument("--resume", default="")
    ap.add_argument("--evaluate", action="store_true")
    ap.add_argument("--snapshot-pref", default="runs/drn")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="launch every kernel of a step from Python instead of replaying it as a hipGraph (stages 1 and 3)")
    args = ap.parse_args()

    rank, local, world = ddist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = yaml.safe_load(open(os.path.join(args.root, args.config)))[args.dataset]
    cfg.update(feature_type=args.feature_type, is_first_stage=args.stage == 1, is_second_stage=args.stage == 2,
               is_third_stage=args.stage == 3)
    word2id = json.load(open(os.path.join(args.root, "data", "dataset", "Charades", "Charades_word2id.json")))
    id2word = {i: w for w, i in word2id.items()}
    train_set = CharadesSTA(cfg, "train", args.root)
    test_set = CharadesSTA(cfg, "test", args.root)
    sampler = torch.utils.data.distributed.DistributedSampler(train_set, world, rank, shuffle=True) if world > 1 else None
    bs = args.batch_size or cfg.get("batch_size", 32)
    # the workers hand the features over in the compute dtype (bf16: half the bytes over PCIe, same bits after the step's cast)
    collate = functools.partial(collate_data, feature_dtype=torch.bfloat16 if args.dtype == "bf16" else None)
    train_loader = DataLoader(train_set, batch_size=bs, shuffle=sampler is None, sampler=sampler, collate_fn=collate,
                              num_workers=args.workers, pin_memory=True, drop_last=world > 1)
    # evaluation is sharded too: every rank scores its share of the test queries (Recall@k with temporal NMS on the device)
    # and the ranks merge their counts (drn_amd.trainer.Trainer.evaluate) -- replaces main.py:275-366 on one process
    test_loader = DataLoader(test_set, batch_size=cfg.get("test_batch_size", 16), shuffle=False, collate_fn=collate,
                             sampler=ShardSampler(test_set, world, rank) if world > 1 else None,
                             num_workers=args.workers, pin_memory=True)

    model = mainModel(len(word2id), argparse.Namespace(**cfg),
                      compute_dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    T.init_glove(model, os.path.join(args.root, cfg.get("glove_weights", "data/glove_weights")))
    model = model.to(dev)
    start_epoch = 0
    if args.resume:
        start_epoch, picked = T.load_checkpoint(model, args.resume, map_location=dev)
        if rank == 0:
            print("resumed %d tensors from %s (epoch %d)" % (len(picked), args.resume, start_epoch))
    tr = T.Trainer(model, args.stage, lr=args.lr or cfg.get("lr", 1e-3), clip_gradient=cfg.get("clip_gradient", 0.5), world_size=world,
                   graph=args.graph)
    if args.evaluate:
        _, topks, accs, _ = tr.evaluate(test_loader, id2word, with_results=False)      # collective: all ranks, merged numbers
        if rank == 0:
            for k, a in zip(topks, accs):
                print("R@{}: {:.1f}".format(k, a * 100))
        return
    n_epoch = args.n_epoch or cfg.get("n_epoch", 50)
    first = 0 if args.stage > 1 else start_epoch                     # the same epoch range on every rank
    hist = tr.fit(train_loader, test_loader, n_epoch=n_epoch, eval_freq=cfg.get("eval_freq", 1),
                  snapshot_pref=args.snapshot_pref, dataset=args.dataset, id2word=id2word, start_epoch=first, rank=rank)
    if rank == 0:
        print(json.dumps(hist[-1] if hist else {}))


if __name__ == "__main__":
    main()
