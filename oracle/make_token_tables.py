"""TEST / DATA INFRASTRUCTURE — tokenises the class lists of the BASELINE configurations with the REAL reference
tokenizer (clip/clip.py:187-223 `tokenize`, clip/simple_tokenizer.py:62-132) exactly as
MultitaskVLPromptLearner.__init__ does (trainers/mvlpt.py:292-305) and stores the INTEGER tables as data:

    python oracle/make_token_tables.py      # build container only (needs /root/reference) -> mvlpt_amd/data/class_prompts.npz

Why: CLIP's BPE merge table is not shipped, so on the GPU box no class list can be tokenised; bench.py and the GPU tests
then use real token ids, real name lengths / EOT positions and the reference's CUT_CONTEXTLEN lengths through
`mvlpt_amd.class_prompts.load_class_prompts` + `PretokenizedPrompts` instead of the hash-based SyntheticTokenizer.

Class lists (name strings are read from the reference's own tables, never hard-coded here):
  caltech101   100 classes: trainers/vision_benchmark/datasets/prompts.py `caltech101_classes` minus the two folders CoOp's
               reader ignores (datasets/caltech101.py:10 IGNORED = BACKGROUND_Google, Faces_easy) — BASELINE configs[0], [1]
  imagenet1k   scripts/classnames.txt (1000) — configs[2]
  coop11       the 11-dataset CoOp multitask set of scripts/mvlpt/main_mt_coopdata_cut.sh:21, 2191 classes: ImageNet,
               Caltech101, Food101, StanfordCars, OxfordPets, OxfordFlowers, FGVCAircraft, SUN397, DTD, EuroSAT
               (datasets/eurosat.py:10-21 NEW_CNAMES), UCF101 — configs[3].  The CoOp readers take most names from dataset
               folders that are not available offline; the ELEVATER tables of the same datasets stand in for them (same
               class counts; flagged approximation)
  elevater20   the 20 ELEVATER datasets listed in scripts/mvlpt/main_mt_coopdata_cut.sh:15, 1151 classes, names from
               prompts.py `class_map` — configs[4]
For every list and n_ctx in {0, 4, 16}: prompts "X ... X name." (or "a photo of a  name." for n_ctx = 0, :201, :295).
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "mvlpt_amd", "data", "class_prompts.npz")
ELEVATER20 = ['hateful-memes', 'cifar-10', 'mnist', 'oxford-flower-102', 'oxford-iiit-pets', 'resisc45_clip', 'country211',
              'food-101', 'stanford-cars', 'fgvc-aircraft-2013b-variants102', 'caltech-101', 'dtd', 'voc-2007-classification',
              'cifar-100', 'patch-camelyon', 'rendered-sst2', 'gtsrb', 'eurosat_clip', 'fer-2013', 'kitti-distance']


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def class_lists():
    R = ref_shim.REFERENCE_ROOT
    P = _load(os.path.join(R, "trainers", "vision_benchmark", "datasets", "prompts.py"), "ref_prompts")
    first = lambda c: c[0] if isinstance(c, (list, tuple)) else c      # fer-2013 lists synonyms per class
    cm = {k: [first(c) for c in v] for k, v in P.class_map.items()}
    with open(os.path.join(R, "scripts", "classnames.txt")) as f:
        imagenet = [" ".join(line.strip().split(" ")[1:]) for line in f if line.strip()]
    caltech = [c for c in cm["caltech-101"] if c not in ("background", "off-center face")]
    src = open(os.path.join(R, "datasets", "eurosat.py")).read()
    ns = {}
    exec(src[src.index("NEW_CNAMES"):src.index("}", src.index("NEW_CNAMES")) + 1], ns)     # the dict literal only
    eurosat = list(ns["NEW_CNAMES"].values())
    coop11 = [("ImageNet", imagenet), ("Caltech101", caltech), ("Food101", cm["food-101"]), ("StanfordCars", cm["stanford-cars"]),
              ("OxfordPets", cm["oxford-iiit-pets"]), ("OxfordFlowers", cm["oxford-flower-102"]),
              ("FGVCAircraft", cm["fgvc-aircraft-2013b-variants102"]), ("SUN397", cm["sun397"]), ("DescribableTextures", cm["dtd"]),
              ("EuroSAT", eurosat), ("UCF101", cm["ucf101"])]
    elev = [(k, cm[k]) for k in ELEVATER20]
    return {"caltech101": [("Caltech101", caltech)], "imagenet1k": [("ImageNet", imagenet)], "coop11": coop11, "elevater20": elev}


def main():
    ref_shim.install()
    from clip import clip as refclip
    from clip.simple_tokenizer import SimpleTokenizer
    tok = SimpleTokenizer()
    out = {}
    for lname, tasks in class_lists().items():
        names = [n.replace("_", " ") for _, cl in tasks for n in cl]                      # trainers/mvlpt.py:292
        out[f"{lname}/task_counts"] = np.array([len(cl) for _, cl in tasks], dtype=np.int32)
        out[f"{lname}/task_names"] = np.array([t for t, _ in tasks])
        out[f"{lname}/name_lens"] = np.array([len(tok.encode(n)) for n in names], dtype=np.int16)   # :293
        for n_ctx in (0, 4, 16):
            prefix = " ".join(["X"] * n_ctx) if n_ctx else "a photo of a "                 # :201, :227
            prompts = [prefix + " " + n + "." for n in names]                             # :295
            cut = max(len(tok.encode(p)) + 2 for p in prompts)                            # :297-300
            ids = np.concatenate([refclip.tokenize(p).numpy() for p in prompts]).astype(np.int64)   # [C,77]
            assert ids.max() == 49407 and not ids[:, cut:].any()
            out[f"{lname}/ids_nctx{n_ctx}"] = ids[:, :cut].astype(np.uint16)
            print(f"{lname:11s} n_ctx={n_ctx:2d}: {len(names)} classes, CUT_CONTEXTLEN length {cut}")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
