"""Data layer of the DRN trainer (reference: dataset.py:13-224) -- SURVEY row 8f-3.  Same on-disk formats, same
8-tuple per sample and per batch, so main.py-style loops run unchanged:

  data/dataset/Charades/Charades_sta_{split}.txt   "VID start end##sentence."                     (dataset.py:64,88-92)
  Charades_fps_dict.json / Charades_duration.json / Charades_word2id.json                        (dataset.py:65-67)
  props file (config `props_file_path`): blocks "#", "<vid>", "<num_frames>", then "<start> <end>" frame pairs (75-85)
  {feature_root}/{vid}.pt: (n_segments, feature_dim) tensor of sliding-window features            (dataset.py:118)

A proposal's feature is the element-wise max over the window features it covers: window size / overlap from the config
entry of `feature_type`, interval = int(window * (1 - overlap)), first index floor(start / interval), every interval up to
the end frame when the proposal is longer than one window, indices clamped to the last stored segment (128-154).
Ground truth is normalised by the video DURATION (seconds), proposals by the frame count (123,160-162).

Tokenisation: the reference uses nltk.word_tokenize (absent here).  When nltk is importable it is used; otherwise a
regular expression that agrees with it on the Charades-STA sentences (lower-case words, full stops removed beforehand,
an occasional comma or apostrophe).  Pass `tokenizer=` to override."""
import json
import os
import re
from itertools import groupby

import numpy as np
import torch
from torch.utils.data import Dataset


def default_tokenizer():
    try:
        import nltk
        return nltk.word_tokenize
    except Exception:
        pat = re.compile(r"[A-Za-z0-9]+|'[A-Za-z]+|[^\sA-Za-z0-9]")
        return lambda s: pat.findall(s)


class CharadesSTA(Dataset):
    def __init__(self, dataset_configs, split="train", root=".", tokenizer=None):
        cfg = dataset_configs if isinstance(dataset_configs, dict) else vars(dataset_configs)
        base = os.path.join(root, "data", "dataset", "Charades")
        self.lang_data = list(open(os.path.join(base, "Charades_sta_%s.txt" % split), "r"))
        self.fps_info = json.load(open(os.path.join(base, "Charades_fps_dict.json"), "r"))
        self.duration_info = json.load(open(os.path.join(base, "Charades_duration.json"), "r"))
        self.word2id = json.load(open(os.path.join(base, "Charades_word2id.json"), "r"))
        ft = cfg[cfg["feature_type"]]
        self.ft_root = ft["feature_root"] if os.path.isabs(ft["feature_root"]) else os.path.join(root, ft["feature_root"])
        self.ft_window_size, self.ft_overlap = ft["ft_window_size"], ft["ft_overlap"]
        self.tokenize = tokenizer or default_tokenizer()
        props_path = cfg["props_file_path"]
        self._load_props(props_path if os.path.isabs(props_path) else os.path.join(root, props_path))
        self._load_queries()

    def _load_props(self, path):
        lines = list(open(path, "r"))
        groups = groupby(lines, lambda x: x.startswith("#"))
        blocks = [[x.strip() for x in g] for k, g in groups if not k]
        self.props = {}
        for blk in blocks:                                         # dataset.py:78-85
            vid = blk[0].split()[-1]
            num_frames = int(blk[1])
            pairs = [(float(x.split()[0]), float(x.split()[1])) for x in blk[2:]]
            # CharadesInstance (dataset.py:13-17): the end frame is truncated to an int and capped at the frame count
            self.props[vid] = (num_frames, [(s, min(int(e), num_frames)) for s, e in pairs])

    def _load_queries(self):
        self.samples = []
        for item in self.lang_data:                                # dataset.py:88-104
            first, sentence = item.strip().split("##")
            sentence = sentence.replace(".", "")
            vid, start, end = first.split()
            tokens = [self.word2id[w] for w in self.tokenize(sentence)]
            duration = float(self.duration_info[vid])
            self.samples.append({"vid": vid, "tokens": tokens, "gt_start_time": float(start),
                                 "gt_end_time": min(float(end), duration), "fps": float(self.fps_info[vid])})

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        smp = self.samples[index]
        vid = smp["vid"]
        num_frames, proposals = self.props[vid]
        duration = float(self.duration_info[vid])
        feats = torch.load(os.path.join(self.ft_root, "%s.pt" % vid))
        interval = int(self.ft_window_size * (1 - self.ft_overlap))
        last = len(feats) - 1
        props, rows = [], []
        for start, end in proposals:                               # dataset.py:119-154
            props.append((start / num_frames, end / num_frames))
            first = (int(start) // interval) * interval
            if end - start <= self.ft_window_size:
                idx = [first // interval]
            else:
                idx = [x // interval for x in range(first, end, interval)]
            idx = sorted(min(last, x) for x in idx)
            rows.append(torch.max(feats[idx, :], dim=0)[0])
        props_fts = torch.stack(rows)
        props_s_e = torch.from_numpy(np.array(props))
        gt = (smp["gt_start_time"] / duration, smp["gt_end_time"] / duration)
        tokens = torch.from_numpy(np.array(smp["tokens"]))
        return vid, props_s_e, props_fts, gt, tokens, len(smp["tokens"]), len(proposals), num_frames


class ShardSampler(torch.utils.data.Sampler):
    """Evaluation shard of one rank: indices rank, rank + world, ... in dataset order -- every sample exactly once over the
    ranks (torch's DistributedSampler pads the last round with repeats, which would count some queries twice in Recall@k)."""

    def __init__(self, dataset, world_size, rank):
        self.n, self.world, self.rank = len(dataset), int(world_size), int(rank)

    def __iter__(self):
        return iter(range(self.rank, self.n, self.world))

    def __len__(self):
        return (self.n - self.rank + self.world - 1) // self.world


def collate_data(batch, feature_dtype=None):
    """dataset.py:180-224: sort by query length (descending, stable), zero-pad proposals and tokens.
    feature_dtype (torch.bfloat16 for a bf16 model; train.py passes it): the padded feature tensor is built in the model's
    compute dtype here, in the DataLoader workers -- the same round-to-nearest-even the step's first kernel would apply to the
    fp32 values, so every result stays bit-identical while the host -> device copy of the features (134 MB per step at
    T = 256, D = 4096: longer than the step itself) halves."""
    data = sorted(batch, key=lambda x: x[5], reverse=True)
    bs = len(batch)
    ft_dim = batch[0][2].size(-1)
    max_props = max(x[6] for x in batch)
    max_len = max(x[5] for x in batch)
    props_features = torch.zeros(bs, max_props, ft_dim, dtype=feature_dtype or torch.float32)
    props_s_e = torch.zeros(bs, max_props, 2, dtype=torch.double)
    query_tokens = torch.zeros(bs, max_len)
    names, gts, qlens, nprops, nframes = [], [], [], [], []
    for i, smp in enumerate(data):
        names.append(smp[0]); gts.append(smp[3]); qlens.append(smp[5]); nprops.append(smp[6]); nframes.append(smp[7])
        query_tokens[i, :smp[5]] = smp[4]
        props_features[i, :smp[6], :] = smp[2]
        props_s_e[i, :smp[6], :] = smp[1]
    return (names, props_s_e, props_features, torch.from_numpy(np.array(gts)).double(), query_tokens.long(),
            torch.LongTensor(np.array(qlens)), torch.from_numpy(np.array(nprops)), torch.from_numpy(np.array(nframes)))
