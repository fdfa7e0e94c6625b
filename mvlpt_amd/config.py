"""Minimal yacs-like config carrying the hot-path keys of the reference (train.py:105-169 `extend_cfg`,
configs/trainers/MVLPT/vit_b16.yaml) with the same names and defaults, so reference scripts' ``KEY VAL``
overrides map one-to-one.  Dassl/yacs are not available offline; this is only the key tree."""
from __future__ import annotations

from typing import Any, Iterable


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self) -> "CfgNode":
        return CfgNode({k: (v.clone() if isinstance(v, CfgNode) else v) for k, v in self.items()})

    def merge_from_list(self, opts: Iterable[Any]) -> None:
        """`KEY VAL KEY VAL …` overrides, as train.py:187."""
        opts = list(opts)
        assert len(opts) % 2 == 0, "override list must be KEY VAL pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"unknown config key {key}")
            old = node[parts[-1]]
            if isinstance(val, str) and not isinstance(old, str):
                if isinstance(old, bool):
                    val = val.lower() in ("1", "true", "yes")
                elif isinstance(old, (list, tuple)):
                    val = type(old)(int(x) for x in val.strip("()[]").split(","))
                else:
                    val = type(old)(val)
            node[parts[-1]] = val


def get_cfg_default() -> CfgNode:
    CN = CfgNode
    cfg = CN()
    cfg.SEED = 1
    cfg.OUTPUT_DIR = "./output"
    cfg.MODEL = CN(BACKBONE=CN(NAME="ViT-B/16"), INIT_WEIGHTS="")
    cfg.INPUT = CN(SIZE=(224, 224))
    cfg.DATALOADER = CN(TRAIN_X=CN(BATCH_SIZE=32), TEST=CN(BATCH_SIZE=100), NUM_WORKERS=8)
    # configs/trainers/MVLPT/vit_b16.yaml:15-22 + Dassl optimizer defaults
    cfg.OPTIM = CN(NAME="sgd", LR=0.002, MAX_EPOCH=200, MOMENTUM=0.9, WEIGHT_DECAY=5e-4, SGD_DAMPNING=0.0,
                   SGD_NESTEROV=False, LR_SCHEDULER="cosine", WARMUP_EPOCH=1, WARMUP_TYPE="constant",
                   WARMUP_CONS_LR=1e-5, WARMUP_MIN_LR=1e-5)
    cfg.TRAIN = CN(PRINT_FREQ=5, CHECKPOINT_FREQ=0)
    cfg.TEST = CN(FINAL_MODEL="last_step", SPLIT="test", NO_TEST=False)   # Dassl's defaults; a run without evaluation data sets NO_TEST True
    cfg.TRAINER = CN(NAME="MVLPT", CUT_CONTEXTLEN=False, ACT_CKPT=1)
    cfg.TRAINER.MVLPT = CN(
        PREC="fp16", PROJECT_METHOD="transformer", PROJECT_DIM=128,
        VPT=CN(N_CTX=0, CSC=False, CTX_INIT="", DROPOUT=0.0, PROJECT=-1, DEEP=True),
        COOP=CN(N_CTX=0, CSC=False, CTX_INIT="", CLASS_TOKEN_POSITION="middle"),
        COCOOP=CN(N_CTX=0, CTX_INIT="", PREC="fp16"),
        # not in the reference: MFMA input type of the MI355X towers
        COMPUTE_DTYPE="fp16",
        # not in the reference: "split_grad" (default: hi+lo operand pairs (GEMMs and attention) in towers that carry a
        # gradient: prompt gradients within 1e-3 of the fp32 CPU path) | "fast" (single 16-bit operands, ~4e-3)
        GRAD_PRECISION="split_grad",
        # not in the reference: compute the NEXT batch's image features underneath the current text-tower backward when
        # the method has no visual prompts (TrainerX.run_epoch reads the loader one batch ahead)
        STEP_PIPELINING=True,
    )
    cfg.DATASET = CN(NAME="synthetic", COOP=True, MULTITASK=False, MULTITASK_LABEL_PERTASK=False,
                     MULTITASK_EVALKEY="average", NUM_SHOTS=16)
    return cfg
