"""Host-side mirror of the reference's prompted-CLIP model API (trainers/mvlpt.py:138-583), MI355X-native.

Same class names, constructor arguments, attribute names and — most importantly — the same
``prompt_learner.state_dict()`` keys (``ctx``, ``vpt_embeddings``, ``vpt_embeddings_deep``, ``token_prefix``,
``token_suffix``, ``mvlpt_proj.resblocks.0.*``, ``mvlpt_proj_ctx_{coop,vpt}_{pre,post}.*``), so checkpoints
interoperate and ``self.model(image, task=…)`` / ``self.model.prompt_learner`` work under the trainer exactly
as in the reference.  The towers do NOT run on PyTorch ops: `CustomCLIP.forward` is one
``torch.autograd.Function`` whose forward/backward call libmvlpt_hip.so (hand-written gfx950 kernels).  Only the
tiny UPT projection (forward_mvlpt_proj, ≤52 tokens × width 128, trainable weights) stays on torch autograd.

Deviations (documented in DESIGN.md): prompt parameters are kept in fp32 even when PREC == "fp16" (fp32 master
copies; the reference keeps fp16 parameters and has no loss scaling); CoCoOp (COCOOP.N_CTX != 0) is out of scope.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from functools import reduce
from operator import mul
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from .engine import Engine
from .weights import ClipArch, _randn, arch_from_state_dict

# compute units of the text tower's partition in the two-tower pipeline (CustomCLIP.set_cu_partition); 0 = no partition.
# Overridden by the environment variable MVLPT_TEXT_CUS.
DEFAULT_TEXT_CUS = 0

SOT_TOKEN, EOT_TOKEN = 49406, 49407     # clip/simple_tokenizer.py: <|startoftext|>, <|endoftext|>
X_TOKEN = 343                            # "X" placeholder word (trainers/mvlpt.py:227), from tests/golden/tokens.npz


# ------------------------------------------------------------------------------------------------ tokenisation
_DEFAULT_TOKENIZER = None


def default_tokenizer():
    """The real thing: own byte-level BPE over the shipped merge table (mvlpt_amd/tokenizer.py), ids bit-exact against the
    reference tokenizer (tests/test_tokenizer.py)."""
    global _DEFAULT_TOKENIZER
    if _DEFAULT_TOKENIZER is None:
        from .tokenizer import BPETokenizer
        _DEFAULT_TOKENIZER = BPETokenizer()
    return _DEFAULT_TOKENIZER


class SyntheticTokenizer:
    """Hash-based stand-in for clip.simple_tokenizer, kept for tests that want token ids independent of any table (rounds 1-2
    used it wherever the BPE merge table was missing; the table ships now and `FrozenCLIP` defaults to the real tokenizer).
    One pseudo-token per whitespace-separated word (stable hash), which reproduces the
    *structure* the hot path depends on: [SOT, prefix words…, name words…, '.', EOT, 0…] and name_lens."""

    def encode(self, text: str) -> List[int]:
        out = []
        for w in text.replace(".", " .").split():
            if w == "X":
                out.append(X_TOKEN)
            elif w == ".":
                out.append(269)
            else:
                h = 0
                for ch in w.lower():
                    h = (h * 131 + ord(ch)) % 40000
                out.append(1000 + h)
        return out

    def tokenize(self, texts, context_length: int = 77) -> torch.Tensor:
        """Same contract as clip.tokenize (clip/clip.py:187-223)."""
        if isinstance(texts, str):
            texts = [texts]
        res = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [SOT_TOKEN] + self.encode(t) + [EOT_TOKEN]
            if len(ids) > context_length:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            res[i, :len(ids)] = torch.tensor(ids)
        return res


class PretokenizedPrompts:
    """Token ids produced elsewhere (e.g. by the reference tokenizer, stored in a fixture)."""

    def __init__(self, tokenized_prompts: torch.Tensor, name_lens: Sequence[int]):
        self.tokenized_prompts = tokenized_prompts.long()
        self.name_lens = [int(x) for x in name_lens]


def build_prompt_layout(name_lens: Sequence[int], n_ctx: int, L: int, position: str) -> torch.Tensor:
    """Source-row table [C, L] (int32) equivalent to the torch.cat's of forward_coop (trainers/mvlpt.py:439-515):
    0 = token_prefix, e > 0 = token_suffix[e-1], e < 0 = ctx[-e-1]."""
    rows = []
    for nl in name_lens:
        ctx = [-(j + 1) for j in range(n_ctx)]
        fixed = list(range(L - n_ctx))
        if n_ctx == 0:
            row = fixed
        elif position == "end":
            row = fixed[:1] + ctx + fixed[1:]
        elif position == "middle":
            half = n_ctx // 2
            row = fixed[:1] + ctx[:half] + fixed[1:1 + nl] + ctx[half:] + fixed[1 + nl:]
        elif position == "front":
            row = fixed[:1] + fixed[1:1 + nl] + ctx + fixed[1 + nl:]
        else:
            raise ValueError(position)
        if len(row) != L:
            raise ValueError("layout length mismatch")
        rows.append(row)
    return torch.tensor(rows, dtype=torch.int32)


# ------------------------------------------------------------------------------------------------ frozen CLIP
class FrozenCLIP:
    """What the reference passes around as ``clip_model``: here the frozen weights live packed inside the
    HIP engine; this object only carries what the prompt learner needs at construction time."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], compute_dtype: str = "fp16", device=None,
                 tokenizer=None, arch: Optional[ClipArch] = None, token_seed: int = 0, precision: str = "split_grad"):
        self.arch = arch or arch_from_state_dict(state_dict)
        self.engine = Engine.from_state_dict(state_dict, compute_dtype, device, self.arch)
        self.engine.set_precision(precision)       # "fast" | "split_grad" | "split_all" (include/mvlpt_hip.h MVLPT_PREC_*)
        self.device = self.engine.device
        self.context_length = self.arch.context_length
        self.logit_scale = state_dict["logit_scale"].detach().float().to(self.device)     # frozen, clip/model.py:291
        self.tokenizer = tokenizer or default_tokenizer()
        self._token_embedding = state_dict.get("token_embedding.weight")
        self._token_seed = token_seed
        self.dtype = torch.float32   # dtype of the prompt parameters (see module docstring)

    def token_embedding(self, ids: torch.Tensor) -> torch.Tensor:
        """clip_model.token_embedding(tokenized_prompts) (trainers/mvlpt.py:306-307); init-time only, CPU."""
        if self._token_embedding is None:
            self._token_embedding = _randn("token_embedding.weight", self._token_seed,
                                           (self.arch.vocab_size, self.arch.transformer_width), 0.02)
        return self._token_embedding.float().cpu()[ids.cpu()]


# ------------------------------------------------------------------------------------------------ UPT projection
class _ProjAttention(nn.Module):
    """Parameter container with nn.MultiheadAttention's names (in_proj_weight, in_proj_bias, out_proj.*)."""

    def __init__(self, d: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _ProjBlock(nn.Module):
    """clip.model.ResidualAttentionBlock(width, heads=1) as the reference USES it in forward_mvlpt_proj
    (trainers/mvlpt.py:403-407): a batch-first [1,T,D] tensor goes into a seq-first block, so attention sees
    sequence length 1 and the softmax over the single key is 1:
        x + out_proj(v_proj(ln_1(x)));  then  + c_proj(QuickGELU(c_fc(ln_2(.))))."""

    def __init__(self, d: int):
        super().__init__()
        self.attn = _ProjAttention(d)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d, 4 * d)), ("gelu", nn.Identity()),
                                              ("c_proj", nn.Linear(4 * d, d))]))
        self.ln_2 = nn.LayerNorm(d)

    def forward(self, x):
        d = x.shape[-1]
        wv, bv = self.attn.in_proj_weight[2 * d:], self.attn.in_proj_bias[2 * d:]
        x = x + self.attn.out_proj(nn.functional.linear(self.ln_1(x), wv, bv))
        u = self.mlp.c_fc(self.ln_2(x))
        return x + self.mlp.c_proj(u * torch.sigmoid(1.702 * u))


class _ProjTransformer(nn.Module):
    def __init__(self, width: int):
        super().__init__()
        self.width, self.layers = width, 1
        self.resblocks = nn.Sequential(_ProjBlock(width))

    def forward(self, x):
        return self.resblocks(x)


# ------------------------------------------------------------------------------------------------ prompt learner
class MultitaskVLPromptLearner(nn.Module):
    """trainers/mvlpt.py:138-515.  ``clip_model`` is a :class:`FrozenCLIP`; ``classnames`` a list of str, or a
    :class:`PretokenizedPrompts` (then ``classnames`` only gives n_cls)."""

    def __init__(self, cfg, classnames, clip_model: FrozenCLIP, pretokenized: Optional[PretokenizedPrompts] = None):
        super().__init__()
        n_cls = len(classnames)
        T = cfg.TRAINER.MVLPT
        coop_n_ctx, cocoop_n_ctx, vpt_n_ctx = T.COOP.N_CTX, T.COCOOP.N_CTX, T.VPT.N_CTX
        if cocoop_n_ctx != 0:
            raise NotImplementedError("CoCoOp (COCOOP.N_CTX != 0) is outside the MI355X hot path (SURVEY §2.1 #5)")
        arch = clip_model.arch
        dtype = clip_model.dtype
        coop_ctx_dim, vpt_ctx_dim = arch.transformer_width, arch.vision_width
        clip_imsize, cfg_imsize = arch.image_resolution, cfg.INPUT.SIZE[0]
        assert cfg_imsize == clip_imsize, f"cfg_imsize ({cfg_imsize}) must equal to clip_imsize ({clip_imsize})"

        self.vpt_dropout = nn.Dropout(T.VPT.DROPOUT)                                    # :165 (applied per image: CustomCLIP.forward)
        self.vpt_deep = T.VPT.DEEP
        self.vpt_embeddings = None
        self.vpt_embeddings_deep = None
        prompt_prefix = ""
        if vpt_n_ctx != 0:
            if T.VPT.PROJECT > -1:                                                      # :170-175
                vpt_dim = T.VPT.PROJECT
                self.vpt_proj = nn.Linear(vpt_dim, vpt_ctx_dim, dtype=dtype)
                nn.init.kaiming_normal_(self.vpt_proj.weight, a=0, mode="fan_out")
            else:
                vpt_dim = vpt_ctx_dim
                self.vpt_proj = nn.Identity()
            if T.VPT.CTX_INIT:
                raise ValueError("CTX initiation scheme is not supported")            # :180-182
            ps = arch.vision_patch_size
            val = math.sqrt(6. / float(3 * reduce(mul, (ps, ps), 1) + vpt_dim))         # :186
            self.vpt_embeddings = nn.Parameter(torch.zeros(1, vpt_n_ctx, vpt_dim, dtype=dtype))
            nn.init.uniform_(self.vpt_embeddings.data, -val, val)
            if self.vpt_deep:
                self.vision_layers = arch.vision_layers
                self.vpt_embeddings_deep = nn.Parameter(torch.zeros(arch.vision_layers - 1, vpt_n_ctx, vpt_dim, dtype=dtype))
                nn.init.uniform_(self.vpt_embeddings_deep.data, -val, val)
            prompt_prefix = "a photo of a "                                             # :201

        self.ctx = None
        if coop_n_ctx != 0:
            if T.COOP.CTX_INIT:
                init = T.COOP.CTX_INIT.replace("_", " ")
                coop_n_ctx = len(init.split(" "))
                ids = clip_model.tokenizer.tokenize(init)
                ctx_vectors = clip_model.token_embedding(ids)[0, 1:1 + coop_n_ctx, :].to(dtype)   # :208-216
                prompt_prefix = init
            else:
                shape = (n_cls, coop_n_ctx, coop_ctx_dim) if T.COOP.CSC else (coop_n_ctx, coop_ctx_dim)
                ctx_vectors = torch.empty(*shape, dtype=dtype)
                nn.init.normal_(ctx_vectors, std=0.02)                                  # :220-226
                prompt_prefix = " ".join(["X"] * coop_n_ctx)
            self.ctx = nn.Parameter(ctx_vectors)

        self.mvlpt_proj = nn.Identity()
        if vpt_n_ctx != 0 and coop_n_ctx != 0:
            self.mvlpt_proj_ctx_dim = T.PROJECT_DIM
            if T.PROJECT_METHOD == "identity":
                self.mvlpt_proj = nn.Identity()
            else:
                self.mvlpt_proj_ctx_vpt_pre, self.mvlpt_proj_ctx_vpt_post = nn.Identity(), nn.Identity()
                self.mvlpt_proj_ctx_coop_pre, self.mvlpt_proj_ctx_coop_post = nn.Identity(), nn.Identity()
                D = self.mvlpt_proj_ctx_dim
                if coop_ctx_dim != D:
                    self.mvlpt_proj_ctx_coop_pre = nn.Linear(coop_ctx_dim, D, dtype=dtype)
                    self.mvlpt_proj_ctx_coop_post = nn.Linear(D, coop_ctx_dim, dtype=dtype)
                if vpt_ctx_dim != D:
                    self.mvlpt_proj_ctx_vpt_pre = nn.Linear(vpt_ctx_dim, D, dtype=dtype)
                    self.mvlpt_proj_ctx_vpt_post = nn.Linear(D, vpt_ctx_dim, dtype=dtype)
                if T.PROJECT_METHOD == "transformer":
                    self.mvlpt_proj = _ProjTransformer(D)
                else:
                    # 'mlp' crashes in the reference too (nn.GeLU, trainers/mvlpt.py:253)
                    raise AttributeError("module 'torch.nn' has no attribute 'GeLU'")
        self.cocoop_ctx = None

        # ---- tokenisation + frozen token embeddings (init time, CPU) : trainers/mvlpt.py:292-316 ----
        if pretokenized is not None:
            tokenized_prompts, name_lens = pretokenized.tokenized_prompts, pretokenized.name_lens
        else:
            tok = clip_model.tokenizer
            names = [n.replace("_", " ") for n in classnames]
            name_lens = [len(tok.encode(n)) for n in names]
            prompts = [prompt_prefix + " " + n + "." for n in names]
            if cfg.TRAINER.CUT_CONTEXTLEN:
                max_length = min(clip_model.context_length, max(len(tok.encode(p)) + 2 for p in prompts))
            else:
                max_length = clip_model.context_length
            tokenized_prompts = torch.cat([tok.tokenize(p, context_length=max_length) for p in prompts])
        with torch.no_grad():
            embedding = clip_model.token_embedding(tokenized_prompts).to(dtype)
        self.register_buffer("token_prefix", embedding[:, :1, :].contiguous())                       # SOS
        self.register_buffer("token_suffix", embedding[:, 1 + coop_n_ctx:, :].contiguous())         # CLS, EOS

        self.n_cls, self.vpt_n_ctx, self.coop_n_ctx, self.cocoop_n_ctx = n_cls, vpt_n_ctx, coop_n_ctx, 0
        self.tokenized_prompts = tokenized_prompts
        self.name_lens = list(name_lens)
        self.class_token_position = T.COOP.CLASS_TOKEN_POSITION
        L = tokenized_prompts.shape[1]
        # integer tables consumed by the HIP text tower (bit-exact indexing)
        self.register_buffer("layout", build_prompt_layout(self.name_lens, coop_n_ctx, L, self.class_token_position),
                             persistent=False)
        self.register_buffer("eot", tokenized_prompts.argmax(dim=-1).to(torch.int32), persistent=False)   # :128
        self.max_eot = int(self.eot.max())

    # trainers/mvlpt.py:376-414
    def forward_mvlpt_proj(self, dtype=torch.float):
        if self.coop_n_ctx == 0 or isinstance(self.mvlpt_proj, nn.Identity) or self.vpt_n_ctx == 0:
            return self.ctx, self.vpt_embeddings, self.vpt_embeddings_deep
        vpt_emb = self.vpt_embeddings
        if self.vpt_deep:
            vpt_emb = torch.cat([vpt_emb, self.vpt_embeddings_deep], dim=0)
        vpt_ctx_dim = vpt_emb.shape[-1]
        vpt_emb = vpt_emb.reshape(1, -1, vpt_ctx_dim)
        coop_emb = self.ctx
        coop_ctx_dim = self.ctx.shape[-1]
        if coop_emb.dim() == 2:
            coop_emb = coop_emb.unsqueeze(0)
        coop_emb = coop_emb.reshape(1, -1, coop_ctx_dim)
        n = coop_emb.shape[1]
        coop_emb = self.mvlpt_proj_ctx_coop_pre(coop_emb)
        vpt_emb = self.mvlpt_proj_ctx_vpt_pre(vpt_emb)
        e = self.mvlpt_proj(torch.cat([coop_emb, vpt_emb], dim=1).float()).type(dtype)
        coop_emb, vpt_emb = e[:, :n, :], e[:, n:, :]
        coop_emb = self.mvlpt_proj_ctx_coop_post(coop_emb).reshape(-1, self.coop_n_ctx, coop_ctx_dim).squeeze(0)
        vpt_emb = self.mvlpt_proj_ctx_vpt_post(vpt_emb).reshape(-1, self.vpt_n_ctx, vpt_ctx_dim)
        vpt_emb_deep = None if vpt_emb.shape[0] == 1 else vpt_emb[1:, :, :]
        return coop_emb, vpt_emb[0, :, :].unsqueeze(0), vpt_emb_deep


# ------------------------------------------------------------------------------------------------ class sharding
def class_shard_bounds(n_cls: int, world: int):
    """Balanced contiguous split of the classes over the ranks: the first n_cls % world ranks own one class more."""
    base, rem = divmod(n_cls, world)
    return [(r * base + min(r, rem), (r + 1) * base + min(r + 1, rem)) for r in range(world)]


class ClassShard:
    """Class-sharded text tower (collective 2): rank r encodes classes bounds[r] only.  Every rank contributes
    a slot of `cmax` rows (the largest shard) to ONE all_gather_into_tensor of the features and receives its slot of ONE
    reduce_scatter_tensor (sum) of their gradients, both on persistent buffers.  Shards of equal size need no other launch
    (the gathered buffer IS the [n_cls, e] feature matrix); ragged shards (n_cls % world != 0) add one index_select /
    index_copy with index tensors built once.  Per step N > 1 adds: 1 copy + 1 all-gather (+1), 1 reduce-scatter (+1)."""

    def __init__(self, rank: int, bounds, device, dtype=torch.float32):
        self.rank, self.bounds, self.world = rank, bounds, len(bounds)
        self.cmax = max(hi - lo for lo, hi in bounds)
        self.lo, self.hi = bounds[rank]
        self.even = all(hi - lo == self.cmax for lo, hi in bounds)
        self.device, self.dtype = device, dtype
        self._e = None
        # slot row of every class (class c of rank r sits at r * cmax + (c - lo_r))
        self.slot_of_class = torch.cat([torch.arange(hi - lo) + r * self.cmax for r, (lo, hi) in enumerate(bounds)]).to(device)

    def _buffers(self, e: int):
        if self._e != e:
            z = lambda rows: torch.zeros(rows, e, device=self.device, dtype=self.dtype)
            self.gin, self.gout, self.sin, self.sout = z(self.cmax), z(self.world * self.cmax), z(self.world * self.cmax), z(self.cmax)
            self._e = e
        return self.gin, self.gout, self.sin, self.sout

    def gather(self, loc: torch.Tensor) -> torch.Tensor:
        """per-rank text features [c_r, e] -> [n_cls, e] on every rank (RCCL over xGMI: <= 4.5 MB)."""
        import torch.distributed as dist
        gin, gout, _, _ = self._buffers(loc.shape[1])
        gin[:loc.shape[0]].copy_(loc)                     # pad rows (ragged shards) stay zero
        if dist.get_backend() == "gloo":                  # CPU tests / one-GPU debug mode: host-staged, disjoint slots summed
            h = torch.zeros(gout.shape, dtype=gout.dtype)
            h[self.rank * self.cmax:(self.rank + 1) * self.cmax] = gin.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            gout.copy_(h)
        else:
            dist.all_gather_into_tensor(gout, gin)
        return gout if self.even else gout.index_select(0, self.slot_of_class)

    def scatter_grads(self, dtxt: torch.Tensor) -> torch.Tensor:
        """reduce-scatter (sum over ranks) of d txt [n_cls, e] back to the class owners -> [hi - lo, e]."""
        import torch.distributed as dist
        _, _, sin, sout = self._buffers(dtxt.shape[1])
        if self.even:
            src = dtxt.contiguous()
        else:
            sin.index_copy_(0, self.slot_of_class, dtxt)  # pad rows were zeroed at allocation and are never written
            src = sin
        if dist.get_backend() == "gloo":                  # gloo has no reduce_scatter: host-staged all-reduce
            h = src.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            sout.copy_(h[self.rank * self.cmax:(self.rank + 1) * self.cmax])
        else:
            dist.reduce_scatter_tensor(sout, src, op=dist.ReduceOp.SUM)
        return sout[:self.hi - self.lo]


# ------------------------------------------------------------------------------------------------ autograd bridge
def _text_inputs(model, pl):
    """token_suffix / layout handed to the text tower.  With `trim_text_to_eot` (off by default) only the first
    max(eot)+1 positions are evaluated: the text transformer is causal (clip/model.py:324-330) and the feature is read
    at the EOT position (trainers/mvlpt.py:128), so later (padding) positions can influence neither the logits nor
    any gradient — the same observation the reference's CUT_CONTEXTLEN option is built on (trainers/mvlpt.py:297-305)."""
    if not model.trim_text_to_eot:
        return pl.token_suffix, pl.layout
    L_eff = pl.max_eot + 1
    return pl.token_suffix[:, :L_eff - 1 - pl.coop_n_ctx], pl.layout[:, :L_eff]


class _PromptedClipFn(torch.autograd.Function):
    """CustomCLIP.forward as ONE autograd node: forward and backward are libmvlpt_hip.so calls."""

    @staticmethod
    def forward(fctx, model: "CustomCLIP", image, task_lo, task_hi, coop_emb, vpt_emb, vpt_deep_emb, grad_on=True):
        eng = model.engine
        pl = model.prompt_learner
        # (grad mode is off inside Function.forward: ask autograd which inputs need a gradient)
        # `grad_on` = torch.is_grad_enabled() at the call site (needs_input_grad ignores torch.no_grad())
        need_txt = grad_on and bool(fctx.needs_input_grad[4])
        need_img = grad_on and bool(fctx.needs_input_grad[5] or fctx.needs_input_grad[6])
        # inference: the text features depend only on the prompt parameters -> computed once per parameter version
        # instead of once per batch as the reference does (trainers/mvlpt.py:546-548 under test(), :989-1088)
        ver = None
        if not pl.training and not need_txt and coop_emb is not None:   # Dassl sets the mode on the registered prompt_learner
            ver = tuple((id(p), p._version) for p in model.prompt_learner.parameters())
            if model._eval_text_cache is not None and model._eval_text_cache[0] == ver:
                model._const_text_features, cached_eval = model._eval_text_cache[1], True
            else:
                cached_eval = False
        else:
            cached_eval = False
        run_text = not ((coop_emb is None or cached_eval) and model._const_text_features is not None)
        suffix, layout = _text_inputs(model, pl)
        # features computed ahead of time (prefetch_image_features), keyed by the tensor they belong to
        pre = model._prefetched.pop((image.data_ptr(), tuple(image.shape), image._version), None) if vpt_emb is None else None
        if pre is not None:
            def image_fwd(*_a, **_k):
                torch.cuda.current_stream().wait_event(pre[1])
                pre[0].record_stream(torch.cuda.current_stream())
                return pre[0]
        else:
            # prefetches for OTHER tensors are (or were) writing the image-tower workspace on their own stream: this forward
            # runs the tower itself, behind them (and they stay available to the forwards they were made for)
            for other in model._prefetched.values():
                torch.cuda.current_stream().wait_event(other[1])
            image_fwd = eng.image_fwd
        shard = model._class_shard if (run_text and coop_emb is not None) else None
        side = model._side_stream if (run_text and model.overlap_towers and shard is None) else None
        if shard is not None:
            # Class-sharded text tower (ClassShard): this rank encodes classes [lo, hi) only and the features are
            # all-gathered; the backward reduce-scatters d(txt) back to the owners.  Tower and collective run on the second
            # stream underneath the image tower, exactly like the replicated text tower below.
            lo, hi = shard.lo, shard.hi
            ctx_loc = coop_emb if coop_emb.dim() == 2 else coop_emb[lo:hi]

            def text_side():
                loc = eng.text_fwd(pl.token_prefix[lo:hi], suffix[lo:hi], ctx_loc, layout[lo:hi], pl.eot[lo:hi],
                                   save_for_bwd=need_txt)
                return shard.gather(loc)
            side_s = model._side_stream if model.overlap_towers else None
            if side_s is not None:
                main = torch.cuda.current_stream()
                side_s.wait_stream(main)
                with torch.cuda.stream(side_s):
                    txt = text_side()
                img = image_fwd(image, vpt_emb, vpt_deep_emb, save_for_bwd=need_img)
                main.wait_stream(side_s)
                txt.record_stream(main)
            else:
                img = image_fwd(image, vpt_emb, vpt_deep_emb, save_for_bwd=need_img)
                txt = text_side()
        elif side is not None:
            # The two towers are independent until the logits: the text tower (few, small launches that cannot
            # fill 256 CUs) runs on a second HIP stream underneath the image tower's large GEMMs.
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                txt = eng.text_fwd(pl.token_prefix, suffix, coop_emb, layout, pl.eot, save_for_bwd=need_txt)
            img = image_fwd(image, vpt_emb, vpt_deep_emb, save_for_bwd=need_img)
            main.wait_stream(side)
            txt.record_stream(main)
        else:
            img = image_fwd(image, vpt_emb, vpt_deep_emb, save_for_bwd=need_img)
            if run_text:
                txt = eng.text_fwd(pl.token_prefix, suffix, coop_emb, layout, pl.eot, save_for_bwd=need_txt)
        if not run_text:
            txt = model._const_text_features
        elif coop_emb is None:
            model._const_text_features = txt     # no text context: features are constants
        if ver is not None:
            model._eval_text_cache = (ver, txt.clone() if shard is not None else txt)   # (a gathered txt lives in a reused buffer)
        if coop_emb is not None:
            model._const_text_features = None    # only the no-context case may persist across training steps
        logits = eng.logits_fwd(img, txt, model.logit_scale_exp, task_lo, task_hi)
        # the engine keeps ONE set of saved activations per tower: stamp this forward so that a backward that arrives
        # after another forward has overwritten them (gradient accumulation, test() between forward and backward) is
        # refused instead of silently using the wrong activations
        model._fwd_generation += 1
        fctx.generation = model._fwd_generation
        fctx.model, fctx.need_img, fctx.need_txt = model, need_img, need_txt
        fctx.shard, fctx.ctx_shape = shard, (None if coop_emb is None else coop_emb.shape)
        fctx.has_deep = vpt_deep_emb is not None
        fctx.vpt_shape = None if vpt_emb is None else vpt_emb.shape
        return logits

    @staticmethod
    def backward(fctx, dlogits):
        eng = fctx.model.engine
        if fctx.generation != fctx.model._fwd_generation:
            raise RuntimeError("backward of a stale forward: the engine holds the saved activations of the most recent "
                               "CustomCLIP.forward only (call backward before the next forward)")
        dimg, dtxt = eng.logits_bwd(dlogits.contiguous(), fctx.need_img, fctx.need_txt)
        dctx = dvpt = ddeep = None
        if fctx.need_txt and fctx.shard is not None:
            lo, hi = fctx.shard.lo, fctx.shard.hi
            # sum over ranks of d(local loss)/d(txt) for the classes this rank owns; the trainer's gradient
            # all-reduce (mean over ranks) then yields d(global mean loss)/d(ctx) summed over all class shards
            own = fctx.shard.scatter_grads(dtxt)
            dloc = eng.text_bwd(own)
            if len(fctx.ctx_shape) == 3:                      # class-specific contexts: only the owned rows are non-zero
                dctx = torch.zeros(fctx.ctx_shape, device=dloc.device, dtype=dloc.dtype)
                dctx[lo:hi] = dloc
            else:
                dctx = dloc
        elif fctx.need_txt:
            part = fctx.model._text_partition
            if part is not None and not fctx.need_img:
                # CU partition (set_cu_partition): the backward of the text tower stays on the text stream's compute units,
                # next to the image tower of the following batch on the prefetch stream's
                main = torch.cuda.current_stream()
                part.wait_stream(main)
                with torch.cuda.stream(part):
                    dctx = eng.text_bwd(dtxt)
                dtxt.record_stream(part)
                main.wait_stream(part)
                dctx.record_stream(main)
            else:
                dctx = eng.text_bwd(dtxt)
        if fctx.need_img:
            dvpt, ddeep = eng.image_bwd(dimg)
            dvpt = dvpt.view(fctx.vpt_shape)
        return None, None, None, None, dctx, dvpt, ddeep, None


class _CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy (mean) with the gradient produced by the same fused HIP kernel."""

    @staticmethod
    def forward(fctx, model, logits, label):
        loss, dl, nc = model.engine.cross_entropy(logits, label, need_grad=bool(fctx.needs_input_grad[1]))
        if dl is not None:
            fctx.save_for_backward(dl)
        model.last_ncorrect = nc
        return loss.squeeze(0)

    @staticmethod
    def backward(fctx, g):
        (dl,) = fctx.saved_tensors
        return None, dl * g, None


class CustomCLIP(nn.Module):
    """trainers/mvlpt.py:517-583."""

    def __init__(self, cfg, classnames, clip_model: FrozenCLIP, dm=None, pretokenized: Optional[PretokenizedPrompts] = None):
        super().__init__()
        self.prompt_learner = MultitaskVLPromptLearner(cfg, classnames, clip_model, pretokenized)
        self.tokenized_prompts = self.prompt_learner.tokenized_prompts
        self.clip_model = clip_model
        self.engine = clip_model.engine
        self.logit_scale = clip_model.logit_scale
        self.logit_scale_exp = float(clip_model.logit_scale.exp())
        self.dtype = clip_model.dtype
        self._const_text_features = None
        self.last_ncorrect = None
        self.overlap_towers = True
        self._class_shard = None
        self._eval_text_cache = None
        self.trim_text_to_eot = False
        self._prefetch_stream = None
        self._prefetched = {}                     # (data_ptr, shape, version) of an image tensor -> (features, event, the tensor:
                                                  # holding it pins its storage, so the pointer cannot be handed to another batch)
        self._fwd_generation = 0
        # MVLPT_TEXT_PRIORITY=1 (experiment): the text tower's stream above, the image prefetch stream below the default priority
        self._prio = os.environ.get("MVLPT_TEXT_PRIORITY", "0") != "0"
        self._side_stream = torch.cuda.Stream(device=clip_model.device, priority=-1 if self._prio else 0) if torch.cuda.is_available() else None
        self._text_partition = None
        self.text_cus = 0
        if self._side_stream is not None:
            self.set_cu_partition(int(os.environ.get("MVLPT_TEXT_CUS", DEFAULT_TEXT_CUS)))
        self.multi_task_label_pertask = cfg.DATASET.MULTITASK_LABEL_PERTASK
        if self.multi_task_label_pertask:
            # indexed by task id; sized num_classes as in the reference (:529-537)
            start = torch.arange(dm._num_classes)
            end = torch.arange(dm._num_classes)
            s = 0
            for i, task in enumerate(dm._task_names):
                start[i] = s
                s += len(dm._labelmap[task])
                end[i] = s
            self.class_index_pertask_start, self.class_index_pertask_end = start, end

    def enable_class_sharding(self, rank: int, world: int) -> None:
        """Shard the text tower over classes across `world` data-parallel ranks (many-class configs: the text tower
        costs C * L tokens per step independent of the batch, so replicating it caps weak scaling)."""
        if world <= 1:
            self._class_shard = None
            return
        C = self.prompt_learner.n_cls
        if world > C:
            raise ValueError(f"class sharding needs at least one class per rank ({C} classes, {world} ranks)")
        self._class_shard = ClassShard(rank, class_shard_bounds(C, world), self.clip_model.device)

    def set_cu_partition(self, text_cus: int) -> None:
        """Run the two towers on disjoint compute units (methods without visual prompts, i.e. the cross-step pipeline of
        prefetch_image_features): the text tower forward AND backward on a stream that owns logical CUs [0, text_cus), the
        image tower of the following batch on a stream that owns the rest (include/mvlpt_hip.h: mvlpt_stream_create_cus).
        Without it the image tower's persistent GEMM workgroups hold every CU for a whole launch and the text tower's short
        kernels only run in their tails (measured: each stretches ~3x, only 1.7 of 4.4 ms hide).  0 switches it off."""
        from .engine import device_cus, partition_stream
        dev = self.clip_model.device
        pl = self.prompt_learner
        if text_cus <= 0 or pl.vpt_embeddings is not None or pl.coop_n_ctx == 0:
            if self._text_partition is not None:
                self._side_stream = torch.cuda.Stream(device=dev)
                self._prefetch_stream = None
            self._text_partition, self.text_cus = None, 0
            return
        total = device_cus(dev)
        text_cus = min(max(8, text_cus // 8 * 8), total - 8)       # whole CUs of every XCD on both sides
        torch.cuda.synchronize(dev)
        self._side_stream = self._text_partition = partition_stream(dev, 0, text_cus)
        self._prefetch_stream = partition_stream(dev, text_cus, total - text_cus)
        self.text_cus = text_cus

    def prefetch_image_features(self, image) -> bool:
        """Software pipelining across steps: with no visual prompts the image tower is a pure function of the image
        (frozen weights, trainers/mvlpt.py:855-858), so the features of the NEXT batch are computed on a third HIP stream
        beside the current step's text tower (forward, backward: small launches that cannot fill the chip), head and
        optimizer.  `forward(image)` picks the result up when it is called with the same tensor.  Successive prefetches
        queue up on that one stream (they share the tower workspace), so the trainer can issue the one for batch i+1 BEFORE
        step i's own forward: the image stream then never waits for a step's logits."""
        pl = self.prompt_learner
        if pl.vpt_embeddings is not None or self._side_stream is None:
            return False
        key = (image.data_ptr(), tuple(image.shape), image._version)
        if key in self._prefetched:
            return True
        main = torch.cuda.current_stream()
        if self._prefetch_stream is None:
            self._prefetch_stream = torch.cuda.Stream(device=self.clip_model.device, priority=1 if self._prio else 0)
        st = self._prefetch_stream
        # the image tensor was produced (uploaded) on the main stream; an image forward that ran on the main stream itself
        # (no prefetch: the first step, an evaluation) must be done with the tower workspace
        ready = torch.cuda.Event()
        ready.record(main)
        st.wait_event(ready)
        with torch.cuda.stream(st):
            feat = self.engine.image_fwd(image, None, None, save_for_bwd=False)
            ev = torch.cuda.Event()
            ev.record(st)
        image.record_stream(st)
        if len(self._prefetched) >= 4:            # never picked up (a loop that reads ahead without consuming): keep the newest
            self._prefetched.pop(next(iter(self._prefetched)))
        self._prefetched[key] = (feat, ev, image)
        return True

    def drop_prefetch(self) -> None:
        """Forget prefetched image forwards that nobody will pick up (the loop left the loader early): the main stream
        waits for the side stream so that the image-tower workspace is quiescent for whatever runs next."""
        pre, self._prefetched = self._prefetched, {}
        for entry in pre.values():
            torch.cuda.current_stream().wait_event(entry[1])

    def prompt_learner_visual_prompts(self):
        """(vpt, vpt_deep) as the image tower takes them: forward_mvlpt_proj, then vpt_proj (no dropout)."""
        _, vpt_emb, vpt_emb_deep = self.prompt_learner.forward_mvlpt_proj(self.dtype)
        if vpt_emb is not None:
            vpt_emb = self.prompt_learner.vpt_proj(vpt_emb)
            vpt_emb_deep = None if vpt_emb_deep is None else self.prompt_learner.vpt_proj(vpt_emb_deep)
        return vpt_emb, vpt_emb_deep

    def vpt_dropout_masks(self, B):
        """Outcome of `vpt_dropout` on the prompt rows of every prompted layer, drawn in the reference's order (the shallow prompts
        in forward_vpt, trainers/mvlpt.py:424, then layer 1, 2, ... in ImageEncoder.forward, :77), each on the rows already expanded
        over the batch: [n_layers, B, n_vpt, width] of 0 or 1 / (1 - p), or None when nothing is dropped (p = 0, eval mode)."""
        pl = self.prompt_learner
        p = float(pl.vpt_dropout.p)
        if pl.vpt_embeddings is None or p == 0.0 or not pl.training:
            return None
        if getattr(self, "_vpt_masks_override", None) is not None:      # tests: the masks the reference drew
            return self._vpt_masks_override
        n_layers = 1 + (pl.vpt_embeddings_deep.shape[0] if (pl.vpt_deep and pl.vpt_embeddings_deep is not None) else 0)
        dev = pl.vpt_embeddings.device
        ones = torch.ones(B, pl.vpt_n_ctx, self.clip_model.arch.vision_width, device=dev)
        return torch.stack([pl.vpt_dropout(ones) for _ in range(n_layers)])

    def forward(self, image, task=None):
        coop_emb, vpt_emb, vpt_emb_deep = self.prompt_learner.forward_mvlpt_proj(self.dtype)
        if vpt_emb is not None:
            # vpt_proj (VPT.PROJECT > -1: a trainable Linear, else Identity) in front of the shallow and of every deep prompt
            # (:424, :77): a few hundred FLOPs on [n, vpt_dim] rows, left to torch autograd like forward_mvlpt_proj
            proj = self.prompt_learner.vpt_proj
            vpt_emb = proj(vpt_emb)
            if vpt_emb_deep is not None:
                vpt_emb_deep = proj(vpt_emb_deep)
            self.engine.set_vpt_dropout(self.vpt_dropout_masks(image.shape[0]))
        lo = hi = None
        if self.multi_task_label_pertask:
            t = task.cpu().long() if torch.is_tensor(task) else torch.as_tensor(task).long()
            lo = self.class_index_pertask_start[t].to(torch.int32).to(image.device)
            hi = self.class_index_pertask_end[t].to(torch.int32).to(image.device)
        return _PromptedClipFn.apply(self, image, lo, hi, coop_emb, vpt_emb, vpt_emb_deep, torch.is_grad_enabled())

    def cross_entropy(self, logits, label):
        """HIP replacement for ``F.cross_entropy(output, label)`` (trainers/mvlpt.py:931)."""
        return _CrossEntropyFn.apply(self, logits, label)
