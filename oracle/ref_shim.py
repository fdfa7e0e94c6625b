"""TEST INFRASTRUCTURE — never imported by the product path.

Import shim that lets the *real* reference classes
(`/root/reference/trainers/mvlpt.py` CustomCLIP, `/root/reference/clip/model.py`
CLIP) be imported in the BUILD container, where Dassl / torchvision / ftfy /
yacs are not installed.  It only pre-seeds ``sys.modules`` with
attribute-permissive stubs for third-party packages the hot path never calls;
no reference source is copied or modified.

Runs only where ``/root/reference`` exists (this container).  The GPU box never
has it: nothing under ``tests -m gpu``, ``bench.py`` or ``smoke()`` imports this
module.  Used by ``oracle/make_golden.py`` to generate ``tests/golden/*.npz``.
"""
import importlib
import os
import sys
import types
from types import SimpleNamespace

REFERENCE_ROOT = os.environ.get("MVLPT_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "trainers", "mvlpt.py"))


class _Permissive(types.ModuleType):
    """Module whose every missing attribute is a harmless placeholder."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        val = type(name, (), {})  # an empty class: usable as base class / decorator target
        setattr(self, name, val)
        return val


def _stub(name: str) -> types.ModuleType:
    mod = sys.modules.get(name)
    if mod is None:
        mod = _Permissive(name)
        mod.__path__ = []  # behave like a package
        sys.modules[name] = mod
    return mod


class _Registry:
    def register(self, *a, **k):
        def deco(cls):
            return cls
        return deco


def install() -> None:
    """Seed sys.modules so `import trainers.mvlpt` works; idempotent."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for name in [
        "dassl", "dassl.engine", "dassl.metrics", "dassl.utils", "dassl.optim",
        "dassl.data", "dassl.data.data_manager", "dassl.data.datasets",
        "dassl.data.samplers", "dassl.data.transforms",
        "torchvision", "torchvision.transforms",
        "trainers.vision_benchmark.evaluation", "trainers.vision_benchmark.datasets",
    ]:
        _stub(name)
    eng = sys.modules["dassl.engine"]
    eng.TRAINER_REGISTRY = _Registry()
    eng.TrainerX = type("TrainerX", (), {})
    sys.modules["dassl.data"].DataManager = type("DataManager", (), {})
    sys.modules["dassl.data.transforms"].INTERPOLATION_MODES = {}
    sys.modules["torchvision.transforms"].InterpolationMode = SimpleNamespace(BICUBIC=3)
    if "ftfy" not in sys.modules:
        ftfy = types.ModuleType("ftfy")
        ftfy.fix_text = lambda s: s  # exact for ASCII class names (clip/simple_tokenizer.py:51)
        sys.modules["ftfy"] = ftfy
    if "tabulate" not in sys.modules:
        try:
            importlib.import_module("tabulate")
        except Exception:
            _stub("tabulate")


def load_reference():
    """Return (trainers.mvlpt module, clip.model module)."""
    install()
    mv = importlib.import_module("trainers.mvlpt")
    cm = importlib.import_module("clip.model")
    return mv, cm


def make_cfg(*, coop_n_ctx=0, vpt_n_ctx=0, vpt_deep=True, csc=False, class_token_position="middle",
             cut_contextlen=False, input_size=224, project_method="transformer", project_dim=128,
             label_pertask=False, prec="fp32", vpt_project=-1, vpt_dropout=0.0, coop_ctx_init="", vpt_ctx_init=""):
    """SimpleNamespace tree with exactly the keys the hot path reads
    (train.py:105-169, trainers/mvlpt.py:139-325,517-538)."""
    ns = SimpleNamespace
    return ns(
        TRAINER=ns(
            MVLPT=ns(
                PREC=prec, PROJECT_METHOD=project_method, PROJECT_DIM=project_dim,
                VPT=ns(N_CTX=vpt_n_ctx, CSC=False, CTX_INIT=vpt_ctx_init, DROPOUT=vpt_dropout, PROJECT=vpt_project, DEEP=vpt_deep),
                COOP=ns(N_CTX=coop_n_ctx, CSC=csc, CTX_INIT=coop_ctx_init, CLASS_TOKEN_POSITION=class_token_position),
                COCOOP=ns(N_CTX=0, CTX_INIT="", PREC="fp16"),
            ),
            CUT_CONTEXTLEN=cut_contextlen, ACT_CKPT=1,
        ),
        INPUT=ns(SIZE=(input_size, input_size)),
        DATASET=ns(MULTITASK_LABEL_PERTASK=label_pertask, COOP=True, MULTITASK=label_pertask),
        MODEL=ns(BACKBONE=ns(NAME="synthetic")),
    )


def make_dm(task_class_counts):
    """Data-manager stand-in for `CustomCLIP.__init__` per-task tables (trainers/mvlpt.py:527-538)."""
    names = [f"task{i}" for i in range(len(task_class_counts))]
    return SimpleNamespace(
        _num_classes=int(sum(task_class_counts)),
        _task_names=names,
        _labelmap={n: list(range(c)) for n, c in zip(names, task_class_counts)},
    )
