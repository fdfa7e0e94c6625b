"""Host wall time of the phases of MVLPT.forward_backward in the pipelined headline loop (no device sync inside): which host call
blocks?  Usage: python tools/host_phase_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd.config import get_cfg_default
from mvlpt_amd.trainer import MVLPT, SyntheticDataManager
from mvlpt_amd.weights import ARCHS, make_state_dict
from mvlpt_amd import class_prompts as CP
arch = ARCHS["ViT-B/16"]
cfg = get_cfg_default(); cfg.MODEL.BACKBONE.NAME = "ViT-B/16"; cfg.INPUT.SIZE = (224, 224)
cfg.DATALOADER.TRAIN_X.BATCH_SIZE = 256; cfg.TRAINER.MVLPT.COOP.N_CTX = 16; cfg.TRAINER.MVLPT.COOP.CLASS_TOKEN_POSITION = "middle"
dm = SyntheticDataManager(cfg, 100, 4, device="cuda", seed=1)
dm.pretokenized, _ = CP.load_class_prompts("caltech101", 16)
tr = MVLPT(cfg, dm=dm, clip_state_dict=make_state_dict(arch, seed=1)); tr.num_batches = 10 ** 9
batches = dm.train_loader_x
T = {}
def tick(name, t0):
    T.setdefault(name, []).append(time.perf_counter() - t0)
model = tr.model
def step(i):
    tr.batch_idx = i
    batch, nxt = batches[i % 4], batches[(i + 1) % 4]
    t0 = time.perf_counter(); image, label, tasks_ = tr.parse_batch_train(batch); tick("parse", t0)
    t0 = time.perf_counter(); out = model(image, task=tasks_); tick("model() [text_fwd on side, pick up prefetched image, logits]", t0)
    t0 = time.perf_counter(); loss = model.cross_entropy(out, label); tick("cross_entropy", t0)
    t0 = time.perf_counter(); model.prefetch_image_features(nxt["img"]); tick("prefetch_image_features(next)", t0)
    t0 = time.perf_counter(); tr.model_zero_grad(); tick("zero_grad", t0)
    t0 = time.perf_counter(); loss.backward(); tick("loss.backward()", t0)
    t0 = time.perf_counter(); tr.sync_gradients(); tr.model_update(); tick("sync + SGD", t0)
for i in range(6): step(i)
torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter()
for i in range(30): step(i)
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"host {host / 30 * 1e3:.2f} ms/step, wall {wall / 30 * 1e3:.2f} ms/step")
for k, v in T.items():
    v = sorted(v)
    print(f"  {k:70s} median {v[len(v) // 2] * 1e3:7.3f} ms   max {v[-1] * 1e3:7.3f}")
