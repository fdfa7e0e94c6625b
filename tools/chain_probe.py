"""GPU timeline of the headline step WITHOUT a profiler: torch.cuda events recorded on the stream every engine call runs on
(text_fwd, image_fwd, logits_fwd, cross_entropy, logits_bwd, text_bwd) and around the optimizer, over steady-state steps of the
trainer's own forward_backward.  Answers: which chain sets the step period, and how long each link takes under contention.
Usage (GPU box): python tools/chain_probe.py [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd.config import get_cfg_default
from mvlpt_amd.trainer import MVLPT, SyntheticDataManager
from mvlpt_amd.weights import ARCHS, make_state_dict
from mvlpt_amd import class_prompts as CP

arch = ARCHS["ViT-B/16"]
cfg = get_cfg_default(); cfg.MODEL.BACKBONE.NAME = "ViT-B/16"; cfg.INPUT.SIZE = (224, 224)
cfg.DATALOADER.TRAIN_X.BATCH_SIZE = 256; cfg.TRAINER.MVLPT.COOP.N_CTX = 16; cfg.TRAINER.MVLPT.COOP.CLASS_TOKEN_POSITION = "middle"
cfg.TRAIN.PRINT_FREQ = 10 ** 9
dm = SyntheticDataManager(cfg, 100, 4, device="cuda", seed=1)
dm.pretokenized, _ = CP.load_class_prompts("caltech101", 16)
tr = MVLPT(cfg, dm=dm, clip_state_dict=make_state_dict(arch, seed=1)); tr.num_batches = 10 ** 9
eng = tr.model.engine
REC = []          # (step, name, start_event, end_event)
cur = {"step": -1, "on": False}

def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        if not cur["on"]:
            return f(*a, **k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = f(*a, **k)
        e.record()
        REC.append((cur["step"], label or name, s, e))
        return r
    setattr(obj, name, g)

for n in ("text_fwd", "image_fwd", "logits_fwd", "cross_entropy", "logits_bwd", "text_bwd"):
    wrap(eng, n)
wrap(tr, "model_zero_grad", "zero_grad")
wrap(tr, "model_update", "optimizer")
batches = dm.train_loader_x
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
base = torch.cuda.Event(enable_timing=True)
for i in range(8 + N):
    tr.batch_idx = i
    tr.next_batch = batches[(i + 1) % 4]
    if i == 8:
        torch.cuda.synchronize(); cur["on"] = True; base.record()
    cur["step"] = i
    tr.forward_backward(batches[i % 4])
torch.cuda.synchronize()
rows = [(st, nm, base.elapsed_time(s), base.elapsed_time(e)) for st, nm, s, e in REC]
# step period from consecutive logits_fwd ends
le = [e for st, nm, s, e in rows if nm == "logits_fwd"]
per = [(b - a) for a, b in zip(le, le[1:])]
print(f"step period (logits_fwd end to end): mean {sum(per) / len(per):.3f} ms  ({', '.join(f'{p:.2f}' for p in per)})")
for st in sorted(set(r[0] for r in rows))[2:5]:
    t0 = [e for s_, nm, s, e in rows if s_ == st and nm == "logits_fwd"][0]
    print(f"-- step {st} (times relative to the end of its logits_fwd)")
    for s_, nm, s, e in sorted([r for r in rows if r[0] == st], key=lambda r: r[2]):
        print(f"   {nm:14s} start {s - t0:+8.3f}  end {e - t0:+8.3f}   ({e - s:6.3f} ms)")
import collections
agg = collections.defaultdict(list)
for st, nm, s, e in rows:
    agg[nm].append(e - s)
print("mean duration per call (ms): " + ", ".join(f"{k} {sum(v) / len(v):.3f}" for k, v in agg.items()))
