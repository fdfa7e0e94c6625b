"""Train-step driver with the reference's trainer surface (trainers/mvlpt.py:827-1125), MI355X-native.

`MVLPT` keeps the method names / signatures the reference overrides on Dassl's ``TrainerX``
(check_cfg, build_model, build_data_loader, forward_backward, parse_batch_train/test, model_inference,
load_model) and the attributes it uses (self.model, self.optim, self.sched, self.device, batch_idx, …).
Dassl itself is third-party, not vendored by the reference and unavailable offline, so `TrainerX` below is a
minimal stand-in of the parts the hot path touches — epochs, SGD + cosine LR with
constant warm-up, model registry, checkpoints in Dassl's dict format.

Multi-GPU: one process per GPU (torch.distributed, RCCL); every rank holds the frozen CLIP and a slice of the
batch; the only exchange per step is ONE all-reduce of the flat prompt-gradient buffer (mvlpt_amd/distributed.py)
— this replaces nn.DataParallel (trainers/mvlpt.py:877-880), which re-broadcasts the frozen weights every step.
"""
from __future__ import annotations

import math
import os
import os.path as osp
import time
from collections import OrderedDict
from typing import Dict, Iterable, Optional

import torch

from . import distributed as dist_utils
from .config import CfgNode
from .model import CustomCLIP, FrozenCLIP
from .weights import ARCHS, make_state_dict


# ------------------------------------------------------------------------------------------------ Dassl stand-ins
def build_optimizer(model: torch.nn.Module, optim_cfg) -> torch.optim.Optimizer:
    """Dassl build_optimizer("sgd") defaults: over model.parameters()."""
    params = [p for p in model.parameters() if p.requires_grad]
    name = optim_cfg.NAME.lower()
    if name == "sgd":
        return torch.optim.SGD(params, lr=optim_cfg.LR, momentum=optim_cfg.MOMENTUM, weight_decay=optim_cfg.WEIGHT_DECAY,
                               dampening=optim_cfg.SGD_DAMPNING, nesterov=optim_cfg.SGD_NESTEROV)
    if name == "adam":
        return torch.optim.Adam(params, lr=optim_cfg.LR, weight_decay=optim_cfg.WEIGHT_DECAY)
    raise ValueError(f"unsupported optimizer {optim_cfg.NAME}")


def load_pretrained_weights(model: torch.nn.Module, weight_path: str) -> None:
    """Dassl `load_pretrained_weights` (recalled): keys that are missing in the model or whose shape
    differs are dropped (a checkpoint trained with another class count still warm-starts the prompts); the class
    dependent buffers token_prefix / token_suffix always come from THIS model's tokenisation."""
    ck = torch.load(weight_path, map_location="cpu")
    sd = ck.get("state_dict", ck)
    own = model.state_dict()
    keep, dropped = {}, []
    for k, v in sd.items():
        k = k[7:] if k.startswith("module.") else k
        if k in ("token_prefix", "token_suffix") or k not in own or own[k].shape != v.shape:
            dropped.append(k)
            continue
        keep[k] = v
    if not keep:
        raise RuntimeError(f'The pretrained weights "{weight_path}" cannot be loaded, check the key names')
    own.update(keep)
    model.load_state_dict(own)
    if dropped:
        print(f"load_pretrained_weights: layers discarded due to unmatched keys or size: {dropped}")


class _ConstantWarmupCosine:
    """CosineAnnealingLR(T_max = MAX_EPOCH) behind a constant warm-up (WARMUP_EPOCH at WARMUP_CONS_LR);
    stepped once per epoch (trainers/mvlpt.py:948-949)."""

    def __init__(self, optim, optim_cfg):
        if optim_cfg.WARMUP_EPOCH > 0 and optim_cfg.WARMUP_TYPE not in ("constant", "linear"):
            raise ValueError(f"unsupported OPTIM.WARMUP_TYPE {optim_cfg.WARMUP_TYPE!r} (constant | linear)")
        if optim_cfg.LR_SCHEDULER != "cosine":      # every reference config uses cosine (configs/trainers/MVLPT/*.yaml)
            raise ValueError(f"unsupported OPTIM.LR_SCHEDULER {optim_cfg.LR_SCHEDULER!r} (only cosine)")
        self.optim, self.cfg = optim, optim_cfg
        self.base_lrs = [g["lr"] for g in optim.param_groups]
        self.last_epoch = 0
        self._apply()

    def _lr(self, base):
        # Dassl's warm-up wrappers only start stepping their successor once the warm-up is over, so the successor
        # (CosineAnnealingLR, T_max = MAX_EPOCH) sees epoch e - WARMUP_EPOCH: the first epoch after the warm-up runs
        # at the full base LR (recalled from Dassl's _BaseWarmupScheduler.step).
        c = self.cfg
        W = c.WARMUP_EPOCH
        if self.last_epoch < W:
            if c.WARMUP_TYPE == "constant":
                return c.WARMUP_CONS_LR
            if self.last_epoch == 0:                      # LinearWarmupScheduler
                return c.WARMUP_MIN_LR
            return base * self.last_epoch / W
        return 0.5 * base * (1 + math.cos(math.pi * (self.last_epoch - W) / c.MAX_EPOCH))

    def _apply(self):
        for g, b in zip(self.optim.param_groups, self.base_lrs):
            g["lr"] = self._lr(b)

    def step(self):
        self.last_epoch += 1
        self._apply()

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lrs": self.base_lrs}

    def load_state_dict(self, sd):
        self.last_epoch, self.base_lrs = sd["last_epoch"], sd["base_lrs"]
        self._apply()


def build_lr_scheduler(optim, optim_cfg):
    return _ConstantWarmupCosine(optim, optim_cfg)


# Enqueue order of MVLPT.forward_backward: the next batch's image tower BEFORE (1) or behind (0, default) this step's own forward.
# Measured (NOTES_experiments.md, round 4): enqueued early the image stream never waits for a step's logits, but the step gets no
# shorter — 14.18-14.21 vs 14.11-14.12 ms in the same run: what bounds the step is the work of the two towers, not that bubble.
_PREFETCH_EARLY = os.environ.get("MVLPT_PREFETCH_EARLY", "0") != "0"


class TrainerX:
    """The slice of dassl.engine.TrainerX the reference relies on."""

    def __init__(self, cfg: CfgNode):
        self._models, self._optims, self._scheds = OrderedDict(), OrderedDict(), OrderedDict()
        self._flat_grads = {}
        self.cfg = cfg
        self.rank, self.world_size, self.local_rank = dist_utils.rank_info()
        self.device = torch.device(f"cuda:{self.local_rank}") if torch.cuda.is_available() else torch.device("cpu")
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)      # the engine's workspaces and streams live on THIS rank's GPU
        dist_utils.init_process_group()             # no-op for one process or when the launcher already did it
        self.start_epoch = self.epoch = 0
        self.max_epoch = cfg.OPTIM.MAX_EPOCH
        self.output_dir = cfg.OUTPUT_DIR
        self.batch_idx, self.num_batches = 0, 0
        self.next_batch, self.batch_hook = None, None
        self.check_cfg(cfg)
        self.build_data_loader()
        self.build_model()

    # -- registry / bookkeeping
    def register_model(self, name="model", model=None, optim=None, sched=None):
        self._models[name], self._optims[name], self._scheds[name] = model, optim, sched

    def get_model_names(self, names=None):
        return list(self._models.keys()) if names is None else list(names)

    def set_model_mode(self, mode="train", names=None):
        for n in self.get_model_names(names):
            self._models[n].train(mode == "train")

    def update_lr(self, names=None):
        for n in self.get_model_names(names):
            if self._scheds[n] is not None:
                self._scheds[n].step()

    def model_zero_grad(self, names=None):
        for n in self.get_model_names(names):
            flat = getattr(self, "_flat_grads", {}).get(n)
            if flat is not None:
                flat.zero_()                      # one launch: every .grad is a view of this buffer
            elif self._optims[n] is not None:
                self._optims[n].zero_grad(set_to_none=False)

    def model_backward(self, loss):
        loss.backward()

    def model_update(self, names=None):
        for n in self.get_model_names(names):
            if self._optims[n] is not None:
                self._optims[n].step()

    def model_backward_and_update(self, loss, names=None):
        """Dassl order: zero_grad -> backward -> step.  The finite-loss check of Dassl's `detect_anomaly` is done
        once per PRINT_FREQ on the host copy of the loss instead of forcing a device sync every step."""
        self.model_zero_grad(names)
        self.model_backward(loss)
        self.sync_gradients(names)
        self.model_update(names)

    def sync_gradients(self, names=None):
        """The one exchange of the data-parallel step (replaces nn.DataParallel's gather, trainers/mvlpt.py:877-880): a single
        in-place all-reduce (mean) of the flat gradient buffer the parameters' .grad alias."""
        if self.world_size > 1:
            for n in self.get_model_names(names):
                flat = getattr(self, "_flat_grads", {}).get(n)
                if flat is not None:
                    flat.all_reduce_mean_(self.world_size)
                else:
                    dist_utils.all_reduce_gradients(self._models[n].parameters(), self.world_size)

    def flatten_gradients(self, name):
        """Give the registered model's parameters one persistent flat gradient buffer (distributed.FlatGradients)."""
        self._flat_grads[name] = dist_utils.FlatGradients(self._models[name].parameters())
        return self._flat_grads[name]

    # -- checkpoints (Dassl format: state_dict, epoch, optimizer, scheduler, val_result)
    def save_model(self, epoch, directory, is_best=False, val_result=None, model_name=""):
        if self.rank != 0:
            return
        for n in self.get_model_names():
            d = osp.join(directory, n)
            os.makedirs(d, exist_ok=True)
            ckpt = {"state_dict": self._models[n].state_dict(), "epoch": epoch + 1,
                    "optimizer": None if self._optims[n] is None else self._optims[n].state_dict(),
                    "scheduler": None if self._scheds[n] is None else self._scheds[n].state_dict(),
                    "val_result": val_result}
            fname = model_name or f"model.pth.tar-{epoch + 1}"
            torch.save(ckpt, osp.join(d, fname))
            if is_best:
                torch.save(ckpt, osp.join(d, "model-best.pth.tar"))

    # -- loop
    def train(self):
        """Dassl's generic loop: before_train / per epoch run_epoch + after_epoch / after_train."""
        self.best_result = -math.inf
        for self.epoch in range(self.start_epoch, self.max_epoch):
            self.run_epoch()
            self.after_epoch()
        self.after_train()

    def run_epoch(self):
        """One pass over train_loader_x: the plain Dassl loop — `for batch_idx, batch in
        enumerate(loader): forward_backward(batch)`.  The one-batch look-ahead that lets `forward_backward` start the next
        step's image tower underneath this step's backward does NOT live here: it comes from the loader itself
        (`LookAheadLoader`, installed by `MVLPT.build_data_loader`), so an unmodified Dassl `run_epoch` gets it too.
        `self.batch_hook(batch_idx)` (optional) runs before every step: bench.py uses it to place its timers."""
        self.set_model_mode("train")
        self.num_batches = len(self.train_loader_x)
        t0 = time.time()
        summary = None
        self.batch_idx = -1
        try:
            for self.batch_idx, batch in enumerate(self.train_loader_x):
                if self.batch_hook is not None and self.batch_hook(self.batch_idx) is False:
                    break
                summary = self.forward_backward(batch)
                if (self.batch_idx + 1) % self.cfg.TRAIN.PRINT_FREQ == 0 and self.rank == 0:
                    vals = {k: (float(v) if torch.is_tensor(v) else v) for k, v in summary.items()}
                    if not math.isfinite(vals["loss"]):
                        raise FloatingPointError("Loss is infinite or NaN!")
                    print(f"epoch [{self.epoch + 1}/{self.max_epoch}] batch [{self.batch_idx + 1}/{self.num_batches}] "
                          f"time {time.time() - t0:.2f}s " + " ".join(f"{k} {v:.4f}" for k, v in vals.items()))
        finally:
            self.end_of_epoch_loop()
        return summary

    def after_epoch(self):
        """Dassl's after_epoch (recalled from upstream): with TEST.FINAL_MODEL == "best_val" validate every
        epoch and keep the best checkpoint as `model-best.pth.tar` — the file `load_model(directory, epoch=None)` opens
        (trainers/mvlpt.py:1098-1104) —, and write `model.pth.tar-<epoch>` at the last epoch and every CHECKPOINT_FREQ epochs."""
        last_epoch = (self.epoch + 1) == self.max_epoch
        freq = self.cfg.TRAIN.CHECKPOINT_FREQ
        if not self.cfg.TEST.NO_TEST and self.cfg.TEST.FINAL_MODEL == "best_val":
            curr = self.test(split="val")
            if curr > getattr(self, "best_result", -math.inf):
                self.best_result = curr
                self.save_model(self.epoch, self.output_dir, val_result=curr, model_name="model-best.pth.tar")
        if last_epoch or (freq > 0 and (self.epoch + 1) % freq == 0):
            self.save_model(self.epoch, self.output_dir)
        # rank 0 writes the checkpoints; nobody goes on (to after_train's load_model of model-best.pth.tar in particular)
        # before the files are complete
        dist_utils.barrier()

    def after_train(self):
        """Dassl's after_train: final test, on the best-validation checkpoint when that is the model selection rule."""
        if self.cfg.TEST.NO_TEST:
            return None
        if self.cfg.TEST.FINAL_MODEL == "best_val":
            self.load_model(self.output_dir)
        return self.test()

    def end_of_epoch_loop(self):
        """The loop left the loader (exhausted, hook break, exception): nothing of the look-ahead may survive it."""
        self.next_batch = None

    # hooks the concrete trainer provides
    def check_cfg(self, cfg):
        pass

    def build_data_loader(self):
        raise NotImplementedError

    def build_model(self):
        raise NotImplementedError

    def forward_backward(self, batch):
        raise NotImplementedError


# ------------------------------------------------------------------------------------------------ look-ahead
class LookAheadLoader:
    """Wraps a train loader so that ANY loop over it — Dassl's own `run_epoch` included — runs one batch ahead: while the
    consumer works on batch i, `owner.next_batch` holds batch i+1 (None on the last one).  `MVLPT.forward_backward(batch)`
    (the reference's one-argument signature, trainers/mvlpt.py:910) reads it to upload batch i+1 and to start its image
    tower underneath the backward of step i.  Length, attributes (`dataset`, `batch_size`, ...) pass through."""

    def __init__(self, loader, owner):
        self.loader, self.owner = loader, owner
        self.hits = 0                      # batches that were handed out with a successor already fetched

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):
        return getattr(self.loader, name)

    def __getitem__(self, i):             # list-like loaders (resident synthetic batches)
        return self.loader[i]

    def __iter__(self):
        it = iter(self.loader)
        nxt = next(it, None)
        try:
            while nxt is not None:
                batch, nxt = nxt, next(it, None)
                self.owner.next_batch = nxt
                self.hits += nxt is not None
                yield batch
        finally:
            self.owner.next_batch = None


# ------------------------------------------------------------------------------------------------ synthetic data
class SyntheticDataManager:
    """Shape-contract stand-in for MVLPTCOOPDataManager (trainers/mvlpt.py:585-671): N(0,1) images, uniform labels,
    optional task ids with labels drawn inside the task's class range."""

    def __init__(self, cfg, num_classes: int, steps_per_epoch: int, task_class_counts=None, device="cpu", seed=1234,
                 soft_labels=False, elevater=False, metric_names=None, book=None):
        """`book` (mvlpt_amd.class_prompts.MultitaskBook): real per-dataset class lists — task names, class names and the
        label offsets come from it (trainers/mvlpt.py:585-645); only the images and the label draws stay synthetic."""
        if book is not None:
            assert num_classes == book.num_classes
            task_class_counts = list(book.num_classes_list)
        self.num_classes = self._num_classes = num_classes
        self.classnames = list(book.classnames) if book is not None else [f"class {i}" for i in range(num_classes)]
        self.lab2cname = dict(book.lab2cname) if book is not None else {i: n for i, n in enumerate(self.classnames)}
        self.dataset = self
        self.num_source_domains = 1
        self._task_names, self._labelmap = [], {}
        self.task_class_counts = task_class_counts
        self._task_class_idx, self._id2task = {}, {}
        if book is not None:
            self._task_names, self._labelmap = list(book._task_names), dict(book._labelmap)
            self._id2task, self._task_class_idx = dict(book._id2task), dict(book._task_class_idx)
        elif task_class_counts:
            self._task_names = [f"task{i}" for i in range(len(task_class_counts))]
            self._labelmap = {n: list(range(c)) for n, c in zip(self._task_names, task_class_counts)}
            self._id2task = dict(enumerate(self._task_names))                       # trainers/mvlpt.py:780-790
            lo = 0
            for n, c in zip(self._task_names, task_class_counts):
                self._task_class_idx[n] = (lo, lo + c)
                lo += c
        if elevater:                                                                # trainers/mvlpt.py:746-747, 782-783
            from .metrics import get_metric
            if task_class_counts:
                names = list(metric_names or ["accuracy"] * len(task_class_counts))
                self._metric_name = dict(zip(self._task_names, names))
                self._metric = {t: get_metric(n) for t, n in self._metric_name.items()}
            else:
                self._metric_name = metric_names or "accuracy"
                self._metric = get_metric(self._metric_name)
        B, R = cfg.DATALOADER.TRAIN_X.BATCH_SIZE, cfg.INPUT.SIZE[0]
        g = torch.Generator().manual_seed(seed)
        batches = []
        for _ in range(steps_per_epoch):
            img = torch.randn(B, 3, R, R, generator=g)
            dom = torch.zeros(B, dtype=torch.long)
            if task_class_counts:
                dom = torch.randint(0, len(task_class_counts), (B,), generator=g)
                starts = torch.tensor([0] + list(torch.tensor(task_class_counts).cumsum(0)[:-1]))
                lab = starts[dom] + (torch.rand(B, generator=g) * torch.tensor(task_class_counts)[dom]).long()
            else:
                lab = torch.randint(0, num_classes, (B,), generator=g)
            # ELEVATER targets: one-hot rows for multitask / multi-label metrics, plain ids for single-task accuracy
            if soft_labels or (elevater and (task_class_counts or self._metric_name != "accuracy")):
                lab = torch.nn.functional.one_hot(lab, num_classes).float()
            if elevater:      # ELEVATER loaders yield (img, target, index string, task id)  (trainers/mvlpt.py:954-957)
                batches.append((img.to(device), lab.to(device), [str(i) for i in range(B)], dom))
            else:
                batches.append({"img": img.to(device), "label": lab.to(device), "domain": dom})
        self.train_loader_x = batches
        self.train_loader_u = self.val_loader = None
        self.test_loader = batches[:1]


# ------------------------------------------------------------------------------------------------ the trainer
class MVLPT(TrainerX):
    """trainers/mvlpt.py:827-1125 on the HIP engine."""

    def __init__(self, cfg, dm=None, clip_state_dict=None):
        self._dm_arg, self._sd_arg = dm, clip_state_dict
        super().__init__(cfg)

    def check_cfg(self, cfg):
        """:835-836 accepts fp16 | fp32 | amp.  Mapping onto the MI355X engine ("Precision modes"):
          fp16 : fp16 MFMA operands, fp32 accumulation / LayerNorm / softmax / residual stream; towers that carry a
                 gradient run with split (hi + lo) operands so that prompt gradients match the fp32 CPU path to 1e-3
          amp  : same engine mode (fp32 master prompts + 16-bit compute + internal gradient scaling IS this engine's
                 fp16 mode; there is no GradScaler object, self.scaler stays None)
          fp32 : every tower in the split-operand mode, forward-only towers included (~22-bit products)."""
        assert cfg.TRAINER.MVLPT.PREC in ["fp16", "fp32", "amp"]        # :835-836

    def build_data_loader(self):
        self.multi_task = self.cfg.DATASET.MULTITASK
        self.multi_task_label_pertask = self.cfg.DATASET.MULTITASK_LABEL_PERTASK
        dm = self._dm_arg
        if dm is None:
            raise ValueError("pass a data manager (e.g. SyntheticDataManager); dataset readers are out of scope")
        # the look-ahead lives in the loader (see LookAheadLoader): a plain `for batch in self.train_loader_x` pipelines
        self.train_loader_x = LookAheadLoader(dm.train_loader_x, self) if self.cfg.TRAINER.MVLPT.STEP_PIPELINING else dm.train_loader_x
        self.train_loader_u = dm.train_loader_u
        self.val_loader, self.test_loader = dm.val_loader, dm.test_loader
        self.num_classes, self.num_source_domains, self.lab2cname = dm.num_classes, dm.num_source_domains, dm.lab2cname
        self.dm = dm

    def build_model(self):
        cfg = self.cfg
        classnames = self.dm.dataset.classnames if cfg.DATASET.COOP else list(self.dm.lab2cname.values())
        # token ids produced elsewhere (mvlpt_amd.class_prompts: reference-tokenizer tables for the BASELINE class lists)
        pretok = getattr(self.dm, "pretokenized", None)
        sd = self._sd_arg
        if sd is None:
            # no network: synthetic frozen weights of the named architecture (clip/clip.py:57 would download)
            sd = make_state_dict(ARCHS[cfg.MODEL.BACKBONE.NAME], seed=cfg.SEED)
        # PREC (see check_cfg): fp16 / amp -> split operands only where gradients flow; fp32 -> in every tower;
        # GRAD_PRECISION = "fast" (not in the reference) drops the split operands altogether (gradients within ~4e-3)
        prec = "split_all" if cfg.TRAINER.MVLPT.PREC == "fp32" else cfg.TRAINER.MVLPT.GRAD_PRECISION
        clip_model = FrozenCLIP(sd, compute_dtype=cfg.TRAINER.MVLPT.COMPUTE_DTYPE, device=self.device, precision=prec)
        self.model = CustomCLIP(cfg, classnames, clip_model, dm=self.dm, pretokenized=pretok)
        for name, param in self.model.named_parameters():               # :855-858 (the towers hold no nn.Parameters)
            if "prompt_learner" not in name:
                param.requires_grad_(False)
        if cfg.MODEL.INIT_WEIGHTS:
            load_pretrained_weights(self.model.prompt_learner, cfg.MODEL.INIT_WEIGHTS)    # :864-865
        self.model.to(self.device)
        if self.world_size > 1:
            dist_utils.broadcast_parameters(self.model.prompt_learner)   # identical prompts on every rank
        self.optim = build_optimizer(self.model.prompt_learner, cfg.OPTIM)   # NOTE: only the prompt learner (:869)
        self.sched = build_lr_scheduler(self.optim, cfg.OPTIM)
        self.register_model("prompt_learner", self.model.prompt_learner, self.optim, self.sched)
        self.flatten_gradients("prompt_learner")      # zero_grad = 1 launch; N > 1: one in-place all-reduce, nothing else
        self.scaler = None    # the HIP backward scales its 16-bit activation gradients internally

    def forward_backward(self, batch):
        """trainers/mvlpt.py:910-951.  When the loop has read one batch ahead (`self.next_batch`, TrainerX.run_epoch) and
        the method has no visual prompts, the image features of the following step are computed underneath this step's
        text-tower backward (model.prefetch_image_features)."""
        next_batch = self.next_batch if self.cfg.TRAINER.MVLPT.STEP_PIPELINING else None
        ahead = getattr(self, "_parsed_ahead", None)
        if ahead is not None and ahead[0] is batch:
            image, label, tasks_ = ahead[1]          # parsed (and uploaded) during the previous step
        else:
            image, label, tasks_ = self.parse_batch_train(batch)
        self._parsed_ahead = None
        if len(label.shape) > 1 and label.shape[-1] > 1:                # :914-916
            label = label.float()
            label = label / label.sum(dim=-1, keepdim=True)
        early = next_batch is not None and _PREFETCH_EARLY
        if early:
            # batch i+1: its H2D copy and its image tower are enqueued BEFORE step i's own forward, so the image stream
            # goes from one batch straight into the next while this step's text tower, head and optimizer run beside it
            parsed = self.parse_batch_train(next_batch)
            self._parsed_ahead = (next_batch, parsed)
            self.model.prefetch_image_features(parsed[0])
        output = self.model(image, task=tasks_)
        loss = self.model.cross_entropy(output, label)                  # F.cross_entropy (:931) as a HIP kernel
        if next_batch is not None and not early:                        # the H2D copy of batch i+1 and its image tower overlap this step's backward
            parsed = self.parse_batch_train(next_batch)
            self._parsed_ahead = (next_batch, parsed)
            self.model.prefetch_image_features(parsed[0])
        self.model_backward_and_update(loss)
        # device tensors: no .item() sync inside the step (the reference syncs twice per step, :941-942)
        loss_summary = {"loss": loss.detach(), "acc": self.model.last_ncorrect[0] * (100.0 / output.shape[0])}
        if tasks_ is not None:
            loss_summary["num_tasks"] = len(set(tasks_.tolist()))
        if (self.batch_idx + 1) == self.num_batches:
            self.update_lr()
        return loss_summary

    def end_of_epoch_loop(self):
        super().end_of_epoch_loop()
        self._parsed_ahead = None
        if getattr(self, "model", None) is not None:
            self.model.drop_prefetch()

    def after_epoch(self):
        super().after_epoch()             # validation / checkpoints (Dassl)
        self.model.engine.trim()          # workspace blocks outgrown during the epoch (a larger eval batch, more classes)

    def parse_batch_train(self, batch):
        if self.cfg.DATASET.COOP:
            inp_key, lab_key, task_key = "img", "label", "domain"
        else:
            inp_key, lab_key, task_key = 0, 1, 3
        tasks = batch[task_key] if self.multi_task else None
        return batch[inp_key].to(self.device), batch[lab_key].to(self.device), tasks

    parse_batch_test = parse_batch_train

    @torch.no_grad()
    def model_inference(self, input, task=None):
        return self.model(input, task=task)

    @torch.no_grad()
    def test(self, split=None):
        """Generic testing pipeline (trainers/mvlpt.py:989-1088).  CoOp data (`cfg.DATASET.COOP`): top-1 accuracy in
        percent, overall and — multitask — per task on the task's own class range (:1035-1039).  ELEVATER data: the
        dataset's own metric (accuracy / mean-per-class / 11point_mAP / roc_auc, `dm._metric[_name]`) on the collected
        logits, per task on the task's column slice (:1048-1060) or overall (:1076-1080).  Multitask runs return the
        average over tasks or the task named by DATASET.MULTITASK_EVALKEY (:1068-1075).  Logits stay on the device
        until the loader is exhausted (the reference copies every batch to the host, :1027-1028)."""
        self.set_model_mode("eval")
        split = split or self.cfg.TEST.SPLIT
        loader = self.val_loader if (split == "val" and self.val_loader is not None) else self.test_loader
        if loader is None:
            raise RuntimeError(f"test(split={split!r}): the data manager has no val/test loader; set TEST.NO_TEST True for "
                               "runs without evaluation data")
        coop = self.cfg.DATASET.COOP
        correct = torch.zeros((), device=self.device)
        total = 0
        per_task = {}
        y_pred, y_true, y_task = [], [], []
        for batch in loader:
            input, label, tasks_ = self.parse_batch_test(batch)
            output = self.model_inference(input, task=tasks_)
            if not coop:
                y_pred.append(output.float())
                y_true.append(label)
                if tasks_ is not None:
                    y_task.append(torch.as_tensor(tasks_).cpu())
                continue
            if label.dim() > 1:
                label = label.argmax(dim=1)
            correct += (output.argmax(dim=1) == label).sum()
            total += label.shape[0]
            if tasks_ is not None and self._task_ranges() is not None:
                ranges = self._task_ranges()
                for t_id in set(tasks_.tolist()):
                    sel = (tasks_ == t_id).to(output.device)
                    lo, hi = ranges[t_id]
                    hit = (output[sel][:, lo:hi].argmax(dim=1) + lo == label[sel]).sum()
                    acc = per_task.setdefault(t_id, [torch.zeros((), device=self.device), 0])
                    acc[0] += hit
                    acc[1] += int(sel.sum())
        results_overall = {}
        if coop:
            results = {"accuracy": 100.0 * float(correct) / max(total, 1)}
            results_overall = {self.dm._task_names[t]: 100.0 * float(c) / max(n, 1) for t, (c, n) in per_task.items()}
        else:
            import numpy as np
            pred = torch.cat(y_pred).cpu().numpy()
            true = torch.cat(y_true).cpu().numpy()
            if y_task:
                task_ids = torch.cat(y_task).numpy()
                for t_id in sorted(set(task_ids.tolist())):
                    task = self.dm._id2task[t_id]
                    lo, hi = self.dm._task_class_idx[task]
                    yt, yp = true[task_ids == t_id][:, lo:hi], pred[task_ids == t_id][:, lo:hi]
                    if self.dm._metric_name[task] == "accuracy":
                        yt = np.argmax(yt, axis=-1)                                      # :1057-1058
                    results_overall[task] = self.dm._metric[task](yt, yp)
            else:
                results = {self.dm._metric_name: self.dm._metric(true, pred)}            # :1076-1080
        if self.multi_task and results_overall:
            key = self.cfg.DATASET.MULTITASK_EVALKEY
            if key == "average":
                results = {"average": sum(results_overall.values()) / len(results_overall)}
            else:
                assert key in results_overall
                results = {key: results_overall[key]}
        self.last_task_results = results_overall
        self.last_results = results
        return list(results.values())[0]

    def _task_ranges(self):
        """[(class_start, class_end)] per task id (trainers/mvlpt.py:785-790), or None for single-task data."""
        idx = getattr(self.dm, "_task_class_idx", None)
        if not idx:
            return None
        return [idx[name] for name in self.dm._task_names]

    def load_model(self, directory, epoch=None):
        if not directory:
            print("Note that load_model() is skipped as no pretrained model is given")
            return
        model_file = "model-best.pth.tar" if epoch is None else "model.pth.tar-" + str(epoch)
        for name in self.get_model_names():
            path = osp.join(directory, name, model_file)
            if not osp.exists(path):
                raise FileNotFoundError('Model not found at "{}"'.format(path))
            ck = torch.load(path, map_location="cpu")
            sd = {k.replace("upt_proj", "mvlpt_proj"): v for k, v in ck["state_dict"].items()}   # :1112
            sd.pop("token_prefix", None)
            sd.pop("token_suffix", None)
            self._models[name].load_state_dict(sd, strict=False)
