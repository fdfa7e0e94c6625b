"""How long the HOST needs to enqueue one training step (no device sync inside the loop) vs the device step time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd.config import get_cfg_default
from mvlpt_amd.trainer import MVLPT, SyntheticDataManager
from mvlpt_amd.weights import ARCHS, make_state_dict
arch = ARCHS["ViT-B/16"]
cfg = get_cfg_default(); cfg.MODEL.BACKBONE.NAME = "ViT-B/16"; cfg.INPUT.SIZE = (224, 224)
cfg.DATALOADER.TRAIN_X.BATCH_SIZE = 256; cfg.TRAINER.MVLPT.COOP.N_CTX = 16
dm = SyntheticDataManager(cfg, 100, 4, device="cuda", seed=1)
tr = MVLPT(cfg, dm=dm, clip_state_dict=make_state_dict(arch, seed=1)); tr.num_batches = 10 ** 9
def step(i, pipe=True):
    tr.batch_idx = i
    return tr.forward_backward(dm.train_loader_x[i % 4], next_batch=dm.train_loader_x[(i + 1) % 4] if pipe else None)
for pipe in (True, False):
    for i in range(5): step(i, pipe)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): step(i, pipe)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"pipelining={pipe}: host enqueue {t_host/20*1e3:.2f} ms/step, wall {t_all/20*1e3:.2f} ms/step")
